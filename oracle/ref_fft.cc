/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- never linked into the product.
 *
 * The transform behind the gr::fft stand-in (shim_blocks/gnuradio/fft/fft.h): the DFT by its definition,
 *   X[k] = sum_n x[n] exp(-/+ j 2 pi n k / N)      (forward / reverse, unnormalised -- FFTW's c2c convention,
 *   which is what gr::fft::fft_complex_fwd / _rev execute),
 * factored by the prime factors of N (decimation in time, generic O(p^2) butterflies, recursion), evaluated in
 * double precision with exact-argument twiddles, inputs taken as float and the result rounded ONCE to float.
 * It is the arbiter transform: closer to the mathematical DFT than any float32 FFT, FFTW's included.
 */
#include <gnuradio/fft/fft.h>
#include <cmath>
#include <complex>
#include <map>
#include <mutex>
#include <vector>

namespace
{
typedef std::complex<double> cd;
ref_fft_hook_t g_hook = nullptr;

struct Plan
{
    int n;
    std::vector<int> factors;
    std::vector<cd> w;  // exp(-j 2 pi k / n), k in [0, n)
};

std::mutex g_mu;
std::map<int, Plan*> g_plans;

Plan* get_plan(int n)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(n);
    if (it != g_plans.end()) return it->second;
    Plan* p = new Plan;
    p->n = n;
    int m = n;
    for (int f = 2; f * f <= m;)
        {
            if (m % f == 0) { p->factors.push_back(f); m /= f; }
            else f++;
        }
    if (m > 1) p->factors.push_back(m);
    p->w.resize(n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n; k++)
        {
            const long double a = two_pi * static_cast<long double>(k) / static_cast<long double>(n);
            p->w[k] = cd(static_cast<double>(cosl(a)), static_cast<double>(-sinl(a)));
        }
    g_plans[n] = p;
    return p;
}

// out[0..m) = DFT_m of in[0], in[stride], ... ; m = product of factors[level..]; tw_step = N / m
void rec(const Plan& P, size_t level, int m, const cd* in, int stride, cd* out, cd* scratch, bool fwd)
{
    if (m == 1) { out[0] = in[0]; return; }
    const int p = P.factors[level];
    const int q = m / p;
    // p sub-transforms of length q over the decimated sequences in[r + p*t]
    for (int r = 0; r < p; r++) rec(P, level + 1, q, in + static_cast<size_t>(r) * stride, stride * p, scratch + static_cast<size_t>(r) * q, out, fwd);
    const int step = P.n / m;  // W_m^k = w[k * step]
    std::vector<cd> t(p);
    for (int k = 0; k < q; k++)
        {
            for (int r = 0; r < p; r++)
                {
                    cd tw = P.w[(static_cast<long long>(r) * k * step) % P.n];
                    if (!fwd) tw = std::conj(tw);
                    t[r] = scratch[static_cast<size_t>(r) * q + k] * tw;
                }
            for (int s = 0; s < p; s++)
                {
                    cd acc = t[0];
                    for (int r = 1; r < p; r++)
                        {
                            cd tw = P.w[(static_cast<long long>(r) * s % p) * (P.n / p)];
                            if (!fwd) tw = std::conj(tw);
                            acc += t[r] * tw;
                        }
                    out[k + static_cast<size_t>(s) * q] = acc;
                }
        }
}
}  // namespace

extern "C" void ref_fft_set_hook(ref_fft_hook_t hook) { g_hook = hook; }

extern "C" void ref_fft_execute(const float* in_iq, float* out_iq, int n, int forward)
{
    if (g_hook != nullptr)
        {
            g_hook(in_iq, out_iq, n, forward);
            return;
        }
    const Plan& P = *get_plan(n);
    std::vector<cd> a(n), b(n), s(n);
    for (int i = 0; i < n; i++) a[i] = cd(in_iq[2 * i], in_iq[2 * i + 1]);
    // the recursion ping-pongs between `out` and `scratch` per level: give every level its own pair by swapping roles
    rec(P, 0, n, a.data(), 1, b.data(), s.data(), forward != 0);
    for (int i = 0; i < n; i++)
        {
            out_iq[2 * i] = static_cast<float>(b[i].real());
            out_iq[2 * i + 1] = static_cast<float>(b[i].imag());
        }
}
