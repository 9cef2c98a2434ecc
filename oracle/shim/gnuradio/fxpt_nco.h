/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build).
 * Link-only stand-in for GNU Radio's <gnuradio/fxpt_nco.h> (GNU Radio is not
 * installed in this image).  gnss_signal_replica.cc needs the type to compile
 * complex_exp_gen()/complex_exp_gen_conj() (gnss_signal_replica.cc:26-39);
 * the oracle never calls those two functions, so this class only has to exist.
 * It is NOT GNU Radio's fixed-point NCO.
 */
#ifndef ORACLE_SHIM_FXPT_NCO_H
#define ORACLE_SHIM_FXPT_NCO_H
#include <cmath>
#include <complex>
namespace gr
{
class fxpt_nco
{
public:
    void set_freq(float angle_rate) { d_inc = angle_rate; }
    void sincos(std::complex<float>* out, int n, double ampl = 1.0)
    {
        double ph = 0.0;
        for (int i = 0; i < n; i++)
            {
                out[i] = std::complex<float>(static_cast<float>(ampl * std::cos(ph)), static_cast<float>(ampl * std::sin(ph)));
                ph += d_inc;
            }
    }

private:
    double d_inc{0.0};
};
}  // namespace gr
#endif
