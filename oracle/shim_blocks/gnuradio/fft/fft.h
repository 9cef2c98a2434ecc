/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build of the reference's acquisition blocks).
 *
 * Stand-in for GNU Radio's <gnuradio/fft/fft.h> (gr::fft::fft_complex_fwd / fft_complex_rev, used through
 * src/algorithms/libs/gnss_sdr_fft.h:26-45 with GNURADIO_FFT_USES_TEMPLATES=1).  GNU Radio and FFTW are not
 * installed here, so the transform itself is the one thing of the block that is not the reference's own code:
 *   default  -- an exact-definition mixed-radix DFT evaluated in double precision and rounded once to float
 *               (forward e^{-j}, reverse e^{+j}, both unnormalised, as FFTW's c2c plans are);
 *   hook     -- ref_fft_set_hook() lets a test substitute any other transform (e.g. scipy's single-precision
 *               pocketfft), so that the block's LOGIC can be compared value for value with a restatement that
 *               uses that same transform.
 * Everything around the transform (buffers, call order, element-wise operations, statistics, state machine) is the
 * reference's own translation unit.
 */
#ifndef ORACLE_SHIM_GR_FFT_H
#define ORACLE_SHIM_GR_FFT_H
#include <gnuradio/gr_complex.h>
#include <vector>

extern "C" {
typedef void (*ref_fft_hook_t)(const float* in_iq, float* out_iq, int n, int forward);
void ref_fft_set_hook(ref_fft_hook_t hook);
void ref_fft_execute(const float* in_iq, float* out_iq, int n, int forward);
}

namespace gr
{
namespace fft
{
template <bool forward>
class fft_complex_mock
{
public:
    explicit fft_complex_mock(int fft_size, int nthreads = 1) : d_in(fft_size), d_out(fft_size) { (void)nthreads; }
    gr_complex* get_inbuf() { return d_in.data(); }
    gr_complex* get_outbuf() { return d_out.data(); }
    int inbuf_length() const { return static_cast<int>(d_in.size()); }
    int outbuf_length() const { return static_cast<int>(d_out.size()); }
    void execute()
    {
        ref_fft_execute(reinterpret_cast<const float*>(d_in.data()), reinterpret_cast<float*>(d_out.data()), static_cast<int>(d_in.size()), forward ? 1 : 0);
    }

private:
    std::vector<gr_complex> d_in, d_out;
};
using fft_complex_fwd = fft_complex_mock<true>;
using fft_complex_rev = fft_complex_mock<false>;
}  // namespace fft
}  // namespace gr
#endif
