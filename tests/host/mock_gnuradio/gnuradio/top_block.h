#ifndef MOCK_GR_TOP_BLOCK_H
#define MOCK_GR_TOP_BLOCK_H
#include "gnuradio/block.h"
#include <memory>
namespace gr
{
class top_block
{
public:
    void connect(basic_block_sptr, int, basic_block_sptr, int) {}
    void disconnect(basic_block_sptr, int, basic_block_sptr, int) {}
};
typedef std::shared_ptr<top_block> top_block_sptr;
}  // namespace gr
#endif
