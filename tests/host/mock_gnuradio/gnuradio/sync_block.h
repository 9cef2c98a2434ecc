#ifndef MOCK_GR_SYNC_BLOCK_H
#define MOCK_GR_SYNC_BLOCK_H
// gr::sync_block as far as gnss-sdr's headers need it (complex_byte_to_float_x2.h: the cbyte path of the reference's acquisition adapters, never run here)
#include "gnuradio/block.h"
namespace gr
{
class sync_block : public block
{
public:
    virtual int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) = 0;
    int general_work(int noutput_items, gr_vector_int&, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        const int n = work(noutput_items, input_items, output_items);
        if (n > 0) consume_each(n);
        return n;
    }

protected:
    sync_block(const std::string& name, io_signature::sptr in, io_signature::sptr out) : block(name, std::move(in), std::move(out)) {}
};
}  // namespace gr
#endif
