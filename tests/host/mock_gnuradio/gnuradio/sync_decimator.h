#ifndef MOCK_GR_SYNC_DECIMATOR_H
#define MOCK_GR_SYNC_DECIMATOR_H
// gr::sync_decimator as far as the reference's data_type_adapter blocks need it (oracle/ref_filt_api.cc drives them): work() over noutput_items outputs,
// decimation x noutput_items inputs consumed
#include "gnuradio/sync_block.h"
namespace gr
{
class sync_decimator : public sync_block
{
public:
    int general_work(int noutput_items, gr_vector_int&, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        const int n = work(noutput_items, input_items, output_items);
        if (n > 0) consume_each(n * static_cast<int>(d_decimation));
        return n;
    }
    unsigned decimation() const { return d_decimation; }

protected:
    sync_decimator(const std::string& name, io_signature::sptr in, io_signature::sptr out, unsigned decimation)
        : sync_block(name, std::move(in), std::move(out)), d_decimation(decimation) {}
    unsigned d_decimation;
};
}  // namespace gr
#endif
