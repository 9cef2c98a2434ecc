// CPU emulation of the on-chip FFT (gnss-sdr_amd/csrc/fft_onchip.h): the SAME per-thread phase functions the
// HIP kernels call, executed thread by thread with a plain array standing in for LDS.  Test infrastructure:
// lets `pytest -m "not gpu"` check the index maps, the register DFTs, the twiddles and the LDS layouts against
// numpy.fft without a GPU.  Built by __graft_entry__.build() with the host side of clang (ext-vector types).
#include "fft_onchip.h"
#include <vector>

namespace
{
using gsh::oc::cf;

template <class P, bool USE64>
void run_plan(const float* in_iq, float* out_iq)
{
    struct A { cf v[P::R1]; };
    struct B { cf v[P::R2]; };
    struct C { cf v[P::R3]; };
    std::vector<A> a(P::T1);
    std::vector<B> b(P::T2);
    std::vector<C> c(P::T3);
    std::vector<float> lds32(P::LDS_FLOATS32, 0.0f);
    std::vector<cf> lds(P::LDS_CF, cf{0.0f, 0.0f});
    for (int t = 0; t < P::T1; t++)
        {
            for (int n1 = 0; n1 < P::R1; n1++)
                {
                    const int n = n1 * P::T1 + t;
                    a[t].v[n1] = cf{in_iq[2 * n], in_iq[2 * n + 1]};
                }
            P::stage1(a[t].v, t);
        }
    if constexpr (!USE64)
        {
            for (int t = 0; t < P::T1; t++) P::template ex1_write32<0>(a[t].v, t, lds32.data());
            for (int t = 0; t < P::T2; t++) P::template ex1_read32<0>(b[t].v, t, lds32.data());
            for (int t = 0; t < P::T1; t++) P::template ex1_write32<1>(a[t].v, t, lds32.data());
            for (int t = 0; t < P::T2; t++) P::template ex1_read32<1>(b[t].v, t, lds32.data());
            for (int t = 0; t < P::T2; t++) P::stage2(b[t].v, t);
            for (int t = 0; t < P::T2; t++) P::template ex2_write32<0>(b[t].v, t, lds32.data());
            for (int t = 0; t < P::T3; t++) P::template ex2_read32<0>(c[t].v, t, lds32.data());
            for (int t = 0; t < P::T2; t++) P::template ex2_write32<1>(b[t].v, t, lds32.data());
            for (int t = 0; t < P::T3; t++) P::template ex2_read32<1>(c[t].v, t, lds32.data());
        }
    else
        {
            // the kernels' step order (pcps_onchip.hip exchange1 / exchange2): "read phase p - 1, write phase p", barrier; the reads of the last phase
            // of exchange 1 and the first write of exchange 2 share a step, as on the device (no barrier between the two exchanges).  Readers are run
            // BEFORE the writers of the same step here, so that a write into a region that is still being read would corrupt the result.
            gsh::oc::static_for<P::NP1>([&](auto PH) {
                constexpr int p = decltype(PH)::value;
                if constexpr (p > 0)
                    for (int t = 0; t < P::T2; t++) P::template ex1_read<(p > 0 ? p - 1 : 0)>(b[t].v, t, lds.data());
                for (int t = 0; t < P::T1; t++) P::template ex1_write<p>(a[t].v, t, lds.data());
            });
            // stage 2 needs exchange 1's last phase; exchange 2's phase 0 is written in the same step as that read on the device: emulate the worst
            // order by writing phase 0 of a scratch copy first is not possible before stage 2 has run -- the device has the same dependency (a thread
            // writes its stage-2 results after ITS reads), so the order here is read, stage 2, write, with the region check below
            for (int t = 0; t < P::T2; t++) P::template ex1_read<P::NP1 - 1>(b[t].v, t, lds.data());
            static_assert(((P::NP1 - 1 + P::START1) % 2) != (P::START2 % 2), "exchange 2 must start in the region exchange 1 does not end in");
            for (int t = 0; t < P::T2; t++) P::stage2(b[t].v, t);
            gsh::oc::static_for<P::NP2>([&](auto PH) {
                constexpr int p = decltype(PH)::value;
                if constexpr (p > 0)
                    for (int t = 0; t < P::T3; t++) P::template ex2_read<(p > 0 ? p - 1 : 0)>(c[t].v, t, lds.data());
                for (int t = 0; t < P::T2; t++) P::template ex2_write<p>(b[t].v, t, lds.data());
            });
            for (int t = 0; t < P::T3; t++) P::template ex2_read<P::NP2 - 1>(c[t].v, t, lds.data());
        }
    for (int t = 0; t < P::T3; t++)
        {
            P::stage3(c[t].v);
            for (int k3 = 0; k3 < P::R3; k3++)
                {
                    const int k = t + P::T3 * k3;
                    out_iq[2 * k] = c[t].v[k3].x;
                    out_iq[2 * k + 1] = c[t].v[k3].y;
                }
        }
}
}  // namespace

extern "C"
{
    // forward DFT of n complex64 values with the on-chip plan for n; returns 0, or -1 when n has no plan.
    // form: 0 = the exchange form the plan uses on the device (Plan::EX64), 1 = component form, 2 = phased 64-bit form
    int oc_host_fft_form(int n, const float* in_iq, float* out_iq, int form)
    {
#define GSH_OC_CASE(r1, r2, r3)                                                          \
    if (n == (r1) * (r2) * (r3))                                                         \
        {                                                                                \
            using P = gsh::oc::Plan<r1, r2, r3>;                                         \
            const bool use64 = form == 0 ? P::EX64 : form == 2;                          \
            if (use64)                                                                   \
                run_plan<P, true>(in_iq, out_iq);                                        \
            else                                                                         \
                run_plan<P, false>(in_iq, out_iq);                                       \
            return 0;                                                                    \
        }
        GSH_OC_PLANS(GSH_OC_CASE)
#undef GSH_OC_CASE
        return -1;
    }
    int oc_host_fft(int n, const float* in_iq, float* out_iq) { return oc_host_fft_form(n, in_iq, out_iq, 0); }

    // plan geometry for the tests: {threads, lds_floats, S1, S2, P2, EX64, NP1, NP2}
    int oc_host_plan_info(int n, int* info)
    {
#define GSH_OC_CASE(r1, r2, r3)                                  \
    if (n == (r1) * (r2) * (r3))                                 \
        {                                                        \
            using P = gsh::oc::Plan<r1, r2, r3>;                 \
            info[0] = P::THREADS;                                \
            info[1] = P::LDS_FLOATS;                             \
            info[2] = P::S1;                                     \
            info[3] = P::S2;                                     \
            info[4] = P::P2;                                     \
            info[5] = P::EX64 ? 1 : 0;                           \
            info[6] = P::NP1;                                    \
            info[7] = P::NP2;                                    \
            return 0;                                            \
        }
        GSH_OC_PLANS(GSH_OC_CASE)
#undef GSH_OC_CASE
        return -1;
    }
}
