#ifndef MOCK_GR_BLOCKS_FLOAT_TO_COMPLEX_H
#define MOCK_GR_BLOCKS_FLOAT_TO_COMPLEX_H
// gr::blocks::float_to_complex: named by base_pcps_acquisition.h for the cbyte item type only; the tests feed gr_complex / cshort, so make() is never reached
#include "gnuradio/sync_block.h"
#include <memory>
namespace gr
{
namespace blocks
{
class float_to_complex : public sync_block
{
public:
    typedef std::shared_ptr<float_to_complex> sptr;
    static sptr make(size_t = 1) { return sptr(); }

protected:
    float_to_complex() : sync_block("float_to_complex", io_signature::make(1, 2, sizeof(float)), io_signature::make(1, 1, sizeof(gr_complex))) {}
};
}  // namespace blocks
}  // namespace gr
#endif
