// PCPS acquisition, whole-transform-on-chip path (MI355X / gfx950).
//
// For transform lengths that have a plan in fft_onchip.h (N * 4 bytes of LDS, three register-resident stages) one
// work-group per compute unit carries out a complete length-N transform without touching HBM in between:
//
//   oc_forward_kernel   x (zero padded, acq.cc:230-247 / :657-664) * Doppler wipe-off (acq.cc:275-281, :531)
//                       -> FFT -> spectrum in NATURAL order               one work-group per Doppler bin (or code)
//   oc_cell_kernel      conj(X_bin[k]) * FFT(code_prn)[k] on load (acq.cc:538) -> FFT (= the reference's IFFT up to a
//                       conjugation that |.|^2 does not see, :541) -> |.|^2 (+ non-coherent accumulation, :545-553)
//                       -> per-row maximum, lowest arg-max, sum and second peak, all on chip (acq.cc:417-431, :485-513)
//                       one work-group per (PRN, bin) cell; the magnitude grid is written only when asked for
//                       the last cell of a PRN to finish also runs the bin scan + statistic of acq.cc:409-519 (agent-scope
//                       release / acquire hand-off through an arrival counter), so a dwell batch is two launches
//
// HBM traffic per cell is two N*8-byte reads that hit L2 / Infinity Cache (a bin's spectrum is shared by every PRN,
// a code spectrum by every bin; the cell -> XCD mapping keeps both inside one XCD's L2) and one 16-byte record.
//
// Lengths beyond what one compute unit holds (N * 4 bytes of LDS, 1024 threads): N = S * M with M a planned length and S = 2, 4 or 8.
// One decimation-in-frequency step of radix S in front of the planned transform makes the S output residues INDEPENDENT length-M
// transforms:   y[S m + r] = FFT_M( a_r )[m],   a_r[n] = ( sum_q Y[n + q M] W_S^{q r} ) W_N^{n r},   0 <= n < M,
// so a cell is S work-groups that never talk to each other (`sub-cells`: each reads the whole product spectrum, forms its own a_r in
// registers and owns the lags tau = S m + r); their row records are merged by the PRN's last arriver.  50 000 = 2 x 25 000 (50 Msps x 1 ms,
// and the bit-transition search at 25 Msps), 100 000 = 4 x 25 000, 128 000 = 8 x 16 000, 32 000 = 2 x 16 000 ...
// bit_transition_flag (acq.cc:110-112, :544): only the lags [offset, offset + effective) of the transform enter the search, as index
// tau - offset -- an epilogue predicate.
#include "pcps_fft.h"
#ifndef GSH_OC_EX2_WRITE_FIRST
#define GSH_OC_EX2_WRITE_FIRST 1
#endif
#ifndef GSH_OC_PASS_BARRIER
#define GSH_OC_PASS_BARRIER 0  // 1: a barrier between the passes of oc_cell_kernel whatever the plan (A/B)
#endif
#include "fft_onchip.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

#ifndef GSH_OC_DIT_MIN_S
#define GSH_OC_DIT_MIN_S 4
#endif
#ifndef GSH_OC_CELLS_PER_WG_DEFAULT
#define GSH_OC_CELLS_PER_WG_DEFAULT 6
#endif
#ifndef GSH_OC_STAGGER_GROUPS_DEFAULT
#define GSH_OC_STAGGER_GROUPS_DEFAULT 0
#define GSH_OC_STAGGER_TICKS_DEFAULT 0
#endif

namespace gsh
{
int onchip_split(int n);
namespace
{
using oc::cf;

constexpr int OC_MAX_WAVES = ONCHIP_MAX_WAVES;

// -DGSH_OC_PROFILE (profiles/ab/build_variant.py; never in the shipped library): every wave of every cell of the headline flavour leaves the shader clock
// (s_memtime) at eight points of its life -- entry, operands loaded + products, stage 1, exchange 1, stage 2, exchange 2, stage 3 + wave reduction, row published --
// and the 100 MHz wall clock at entry and exit; gsh_debug_oc_profile copies them out (profiles/ab/r05/oc_cell_phases.py -> profiles/oc_cell_annotated.txt)
#ifdef GSH_OC_PROFILE
constexpr int OC_PROF_WORDS = 10;
__device__ unsigned long long g_oc_prof[4096 * OC_MAX_WAVES * OC_PROF_WORDS];
#define OC_STAMP(k)                                                                                                                         \
    do                                                                                                                                      \
        {                                                                                                                                   \
            if ((threadIdx.x & 63) == 0 && cell < 4096) g_oc_prof[(static_cast<size_t>(cell) * OC_MAX_WAVES + (threadIdx.x >> 6)) * OC_PROF_WORDS + (k)] = __builtin_amdgcn_s_memtime(); \
        }                                                                                                                                   \
    while (0)
#define OC_STAMP_WALL(k)                                                                                                                    \
    do                                                                                                                                      \
        {                                                                                                                                   \
            if ((threadIdx.x & 63) == 0 && cell < 4096) g_oc_prof[(static_cast<size_t>(cell) * OC_MAX_WAVES + (threadIdx.x >> 6)) * OC_PROF_WORDS + (k)] = wall_clock64(); \
        }                                                                                                                                   \
    while (0)
#else
#define OC_STAMP(k) \
    do              \
        {           \
        }           \
    while (0)
#define OC_STAMP_WALL(k) \
    do                   \
        {                \
        }                \
    while (0)
#endif

struct OcFwdArgs
{
    const cf* src;
    size_t src_stride;
    int n_in;
    int place_off;
    const float* wipe_hz;
    double inv_fs;
    cf* dst;
    int fold;  // > 1: the wiped-off input is summed over `fold` segments of N samples (pcps_quicksync_acquisition_cc.cc:243-263)
               // (S == 1 only)
    int residue_major;  // split plans: 1 = sub-transform r writes X[S m + r] at dst[r M + m] (the decimation-in-time cells read one residue class each);
                        // 0 = natural order dst[S m + r]
};

struct OcCellArgs
{
    const cf* spectra;  // n_bins * N, natural order
    const cf* codes;    // n_prn * N, natural order, UNconjugated forward FFT of the placed code
    float* grid;        // n_prn * n_bins * effective (touched only when store_grid / accumulate)
    RowStat* rows;      // n_prn * n_bins
    RowStat* subrows;   // S > 1: n_prn * n_bins * S records of the sub-cells, merged into `rows` by the PRN's last arriver
    RowStat* waverows;  // flavours without the second peak: n_prn * n_bins * S * OC_MAX_WAVES per-WAVE partial records (maximum, lowest arg-max, sum) -- a cell's waves leave
                        // them and go; oc_rows_kernel, queued behind the cells, merges them in the order the in-kernel reduction used and forms the statistic
    int offset;         // first lag of the transform that enters the search (bit_transition_flag: effective; else 0)
    DevAcqResult* results;   // n_prn
    unsigned* arrivals;      // n_prn arrival counters (zero between launches): the last cell of a PRN forms its statistic
    int n_prn, n_bins;
    int xp, prn_per, bin_per;  // XCD tiling: xp * (8 / xp) XCDs, each owns prn_per PRNs x bin_per bins
    int effective, accumulate, store_grid;  // offset + effective <= N
    int samples_per_chip, want_second;
    int use_cfar;
    unsigned dwell_count;
    float weight;  // GRID path: every |.|^2 is scaled before it is added / stored (pcps_tong_acquisition_cc.cc:243-249); 1 otherwise
    cf* z;  // decimation-in-time split (oc_subcell_dit_kernel / oc_combine_dit_kernel): the sub-cells' length-M transforms, n_prn * n_bins * N values
    int dit_r_major;  // oc_subcell_dit_kernel: an XCD walks its sub-cells residue class by residue class (all its cells' sub-cell 0, then all its cells' sub-cell 1, ...)
    int cells_per_wg;   // oc_cell_kernel: a work-group carries out this many cells one after the other (slot, slot + gridDim / 8, ...): the staggered start of a
                        // work-group's sixteen waves -- 3.5 of a cell's 21 us, profiles/oc_cell_annotated.txt -- is paid once per work-group instead of once per cell
    int slots_per_xcd;  // prn_per * bin_per * S
    int prefetch_next;    // >= 1: the idle threads of stage 3 touch the next cell's bin spectrum (cells_per_wg > 1); >= 2: the idle WAVES also load their own operands of it
    int stagger_groups;   // > 1: the work-groups of the launch's FIRST round (blockIdx < stagger_first) start in this many groups, stagger_ticks of the 100 MHz clock apart:
    int stagger_ticks;    // all 256 compute units loading their operands in the same microseconds and transforming in the same microseconds leaves the L2 idle 60 % of
    int stagger_first;    // the time and overrun the rest (7.4 us for 400 KB per unit, against 1.9 us when they come apart); every cell takes the same time, so
                          // an offset given once stays for the whole launch (profiles/oc_cell_annotated.txt)
};

// x[k] for 0 <= k < n_in, else 0 -- as a load from a clamped index plus a select, not a branch around the load: the compiler turns the
// conditional form into one s_cbranch_execz + s_waitcnt vmcnt(0) per element, i.e. R1 dependent round trips to memory instead of R1 loads in flight
__device__ __forceinline__ cf load_or_zero(const cf* __restrict__ src, int k, int n_in)
{
    const bool ok = static_cast<unsigned>(k) < static_cast<unsigned>(n_in);
    const cf v = src[ok ? k : 0];
    return ok ? v : cf{0.0f, 0.0f};
}

// exp(-j 2 pi f n / fs) with the product reduced in double before the float sincos
__device__ __forceinline__ cf wipe_phasor(float f_hz, int n, double inv_fs)
{
    double rev = static_cast<double>(f_hz) * static_cast<double>(n) * inv_fs;
    rev -= rint(rev);
    float s, c;
    sincospif(static_cast<float>(2.0 * rev), &s, &c);
    return cf{c, -s};
}

// ---- the two LDS re-distributions.  Plan::EX64 picks the form (fft_onchip.h):
//   component form   one float component at a time (N * 8 bytes do not fit, N * 4 do): write x, read x, write y, read y
//   phased form      whole complex values, a few rows per phase, two regions in turn; step p = "read phase p - 1, write phase p", one barrier per
//                    step: the reads of a phase are in flight while the next phase is written, and no barrier separates the two exchanges
// every element of `v` "written" by an empty asm: ends the live range of whatever the registers held, without an instruction (oc_cell_kernel's pass loop)
template <int R>
__device__ __forceinline__ void fresh_values(cf (&v)[R])
{
    oc::static_for<R>([&](auto K) GSH_AI {
        constexpr int k = decltype(K)::value;
        float x, y;
        asm volatile("" : "=v"(x), "=v"(y));
        v[k] = cf{x, y};
    });
}

// FRESH (oc_cell_kernel's pass loop): the destination registers are declared written (fresh_values) right in front of the first read that fills them -- not earlier:
// the source registers die phase by phase, and both arrays whole do not fit the register file
template <class P, bool FRESH = false>
__device__ __forceinline__ void exchange1(const cf (&ra)[P::R1], cf (&rb)[P::R2], int t, unsigned char* lds_raw)
{
    if constexpr (P::EX64)
        {
            cf* lds = reinterpret_cast<cf*>(lds_raw);
            oc::static_for<P::NP1>([&](auto PH) GSH_AI {
                constexpr int p = decltype(PH)::value;
                if constexpr (p == 1 && FRESH) fresh_values<P::R2>(rb);
                if constexpr (p > 0)
                    if (t < P::T2) P::template ex1_read<(p > 0 ? p - 1 : 0)>(rb, t, lds);
                if (t < P::T1) P::template ex1_write<p>(ra, t, lds);
                __syncthreads();
            });
            if constexpr (P::NP1 == 1 && FRESH) fresh_values<P::R2>(rb);
            if (t < P::T2) P::template ex1_read<P::NP1 - 1>(rb, t, lds);
        }
    else
        {
            float* lds = reinterpret_cast<float*>(lds_raw);
            if (t < P::T1) P::template ex1_write32<0>(ra, t, lds);
            __syncthreads();
            if constexpr (FRESH) fresh_values<P::R2>(rb);
            if (t < P::T2) P::template ex1_read32<0>(rb, t, lds);
            __syncthreads();
            if (t < P::T1) P::template ex1_write32<1>(ra, t, lds);
            __syncthreads();
            if (t < P::T2) P::template ex1_read32<1>(rb, t, lds);
        }
}

template <class P, bool FRESH = false>
__device__ __forceinline__ void exchange2(const cf (&rb)[P::R2], cf (&rc)[P::R3], int t, unsigned char* lds_raw)
{
    if constexpr (P::EX64)
        {
            // (no barrier in front: phase 0 goes to the region exchange 1 did not end in, and whatever was read from it was read before
            // exchange 1's last barrier)
            cf* lds = reinterpret_cast<cf*>(lds_raw);
            oc::static_for<P::NP2>([&](auto PH) GSH_AI {
                constexpr int p = decltype(PH)::value;
#if GSH_OC_EX2_WRITE_FIRST  /* the phase's rows leave before the previous phase is read -- the two go to different regions: 6 - 18 registers fewer in every flavour, 0.4 % faster (profiles/oc_cell_annotated.txt; 0: the other order, A/B) */
                if (t < P::T2) P::template ex2_write<p>(rb, t, lds);
#endif
                if constexpr (p == 1 && FRESH) fresh_values<P::R3>(rc);
                if constexpr (p > 0)
                    if (t < P::T3) P::template ex2_read<(p > 0 ? p - 1 : 0)>(rc, t, lds);
#if !GSH_OC_EX2_WRITE_FIRST
                if (t < P::T2) P::template ex2_write<p>(rb, t, lds);
#endif
                __syncthreads();
            });
            if constexpr (P::NP2 == 1 && FRESH) fresh_values<P::R3>(rc);
            if (t < P::T3) P::template ex2_read<P::NP2 - 1>(rc, t, lds);
        }
    else
        {
            float* lds = reinterpret_cast<float*>(lds_raw);
            __syncthreads();  // every exchange-1 read has been issued and consumed
            if (t < P::T2) P::template ex2_write32<0>(rb, t, lds);
            __syncthreads();
            if constexpr (FRESH) fresh_values<P::R3>(rc);
            if (t < P::T3) P::template ex2_read32<0>(rc, t, lds);
            __syncthreads();
            if (t < P::T2) P::template ex2_write32<1>(rb, t, lds);
            __syncthreads();
            if (t < P::T3) P::template ex2_read32<1>(rc, t, lds);
        }
}
#define GSH_OC_LDS_DECL(P) __shared__ __align__(16) unsigned char lds[P::LDS_BYTES]

// ------------------------------------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(P::THREADS) void oc_forward_kernel(OcFwdArgs a)
{
    GSH_OC_LDS_DECL(P);
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    cf ra[P::R1], rb[P::R2], rc[P::R3];
    if (t < P::T1)
        {
            const cf* __restrict__ src = a.src + static_cast<size_t>(b) * a.src_stride;
            if (a.fold > 1)
                {
                    // sum_seg x[n + seg N] w[n + seg N] = w[n] * sum_seg x[n + seg N] W_seg with W_seg = w[seg N], uniform over the
                    // work-group: the segments are folded with their scalar phasor here, the per-sample wipe-off follows below
                    oc::static_for<P::R1>([&](auto N1) GSH_AI { ra[decltype(N1)::value] = cf{0.0f, 0.0f}; });
                    const float f = a.wipe_hz != nullptr ? a.wipe_hz[b] : 0.0f;
                    for (int seg = 0; seg < a.fold; seg++)
                        {
                            const cf ws = a.wipe_hz != nullptr ? wipe_phasor(f, seg * P::N, a.inv_fs) : cf{1.0f, 0.0f};
                            oc::static_for<P::R1>([&](auto N1) GSH_AI {
                                constexpr int n1 = decltype(N1)::value;
                                const int k = n1 * P::T1 + t + seg * P::N;
                                const cf u = oc::cmul(load_or_zero(src, k, a.n_in), ws);
                                ra[n1].x += u.x;
                                ra[n1].y += u.y;
                            });
                        }
                }
            else
                oc::static_for<P::R1>([&](auto N1) GSH_AI {
                    constexpr int n1 = decltype(N1)::value;
                    const int k = n1 * P::T1 + t - a.place_off;
                    ra[n1] = load_or_zero(src, k, a.n_in);
                });
            if (a.wipe_hz != nullptr)
                {
                    // w[n] = w[t] * (w[T1])^n1, both seeds exact, powers by the squaring tree
                    const float f = a.wipe_hz[b];
                    const cf w0 = wipe_phasor(f, t, a.inv_fs);
                    oc::mul_powers<P::R1>(ra, wipe_phasor(f, P::T1, a.inv_fs));
                    oc::static_for<P::R1>([&](auto N1) GSH_AI { ra[decltype(N1)::value] = oc::cmul(ra[decltype(N1)::value], w0); });
                }
            P::stage1(ra, t);
        }
    exchange1<P>(ra, rb, t, lds);
    if (t < P::T2) P::stage2(rb, t);
    exchange2<P>(rb, rc, t, lds);
    if (t < P::T3)
        {
            P::stage3(rc);
            cf* __restrict__ dst = a.dst + static_cast<size_t>(b) * P::N + t;
            oc::static_for<P::R3>([&](auto K3) GSH_AI { dst[decltype(K3)::value * P::T3] = rc[decltype(K3)::value]; });
        }
}

// ---- N = S * M: work-group (item, r) forms a_r on load and transforms it; X[S m + r] = FFT_M(a_r)[m]
// exp(-2 pi i q r / S): exact for the quarter turns, correctly rounded for the eighths
__device__ __forceinline__ cf radix_root(int qr, int s)
{
    return oc::unit_root(qr % s, s);
}

template <class P, int S>
__global__ __launch_bounds__(P::THREADS) void oc_forward_split_kernel(OcFwdArgs a)
{
    GSH_OC_LDS_DECL(P);
    constexpr int M = P::N, N = S * P::N;
    const int t = threadIdx.x;
    const int b = static_cast<int>(blockIdx.x) / S, r = static_cast<int>(blockIdx.x) - b * S;
    cf ra[P::R1], rb[P::R2], rc[P::R3];
    if (t < P::T1)
        {
            const cf* __restrict__ src = a.src + static_cast<size_t>(b) * a.src_stride;
            const bool wipe = a.wipe_hz != nullptr;
            const float f = wipe ? a.wipe_hz[b] : 0.0f;
            // sum_q x[n + q M] w[n + q M] W_S^{q r}: the wipe-off of sample n + q M is w[n] * w[q M], the second factor uniform over the work-group
            // (a rolled loop over q: unrolled, the scheduler hoists all S * R1 loads and spills)
            oc::static_for<P::R1>([&](auto N1) GSH_AI {
                constexpr int n1 = decltype(N1)::value;
                const int k = n1 * P::T1 + t - a.place_off;
                ra[n1] = load_or_zero(src, k, a.n_in);
            });
#pragma clang loop unroll(disable)
            for (int q = 1; q < S; q++)
                {
                    cf cq = radix_root(q * r, S);
                    if (wipe) cq = oc::cmul(cq, wipe_phasor(f, q * M, a.inv_fs));
                    // The R1 running sums already take up to half the register file, and the compiler would put all R1 loads of this pass in flight on
                    // top of them (read-only, no-alias memory: nothing orders them) and spill.  Eight at a time: the index of the next eight passes
                    // through an empty asm that also takes the sum the previous eight ended on, so they cannot be issued before that sum exists.
                    int kq = t + q * M - a.place_off;
                    oc::static_for<P::R1>([&](auto N1) GSH_AI {
                        constexpr int n1 = decltype(N1)::value;
                        const cf v = load_or_zero(src, n1 * P::T1 + kq, a.n_in);
                        const cf u = oc::cmul(v, cq);
                        ra[n1].x += u.x;
                        ra[n1].y += u.y;
                        if constexpr (n1 % 8 == 7) asm volatile("" : "+v"(kq) : "v"(ra[n1].x));
                    });
                }
            // per-sample factor w[n] W_N^{n r}, n = n1 T1 + t: seed (t) and step (T1) of both combined before the power tree
            cf seed = oc::unit_root((t * r) % N, N), step = oc::unit_root((P::T1 * r) % N, N);
            if (wipe)
                {
                    seed = oc::cmul(seed, wipe_phasor(f, t, a.inv_fs));
                    step = oc::cmul(step, wipe_phasor(f, P::T1, a.inv_fs));
                }
            if (wipe || r != 0)
                {
                    oc::mul_powers<P::R1>(ra, step);
                    oc::static_for<P::R1>([&](auto N1) GSH_AI { ra[decltype(N1)::value] = oc::cmul(ra[decltype(N1)::value], seed); });
                }
            P::stage1(ra, t);
        }
    exchange1<P>(ra, rb, t, lds);
    if (t < P::T2) P::stage2(rb, t);
    exchange2<P>(rb, rc, t, lds);
    if (t < P::T3)
        {
            P::stage3(rc);
            if (a.residue_major)  // uniform
                {
                    cf* __restrict__ dst = a.dst + static_cast<size_t>(b) * N + static_cast<size_t>(r) * M + t;
                    oc::static_for<P::R3>([&](auto K3) GSH_AI { dst[decltype(K3)::value * P::T3] = rc[decltype(K3)::value]; });
                }
            else
                {
                    cf* __restrict__ dst = a.dst + static_cast<size_t>(b) * N + static_cast<size_t>(t) * S + r;
                    oc::static_for<P::R3>([&](auto K3) GSH_AI { dst[static_cast<size_t>(decltype(K3)::value) * P::T3 * S] = rc[decltype(K3)::value]; });
                }
        }
}

// lowest index wins ties (K/volk_gnsssdr_32f_index_max_32u.h:457: strict '>' scanning upwards)
__device__ __forceinline__ void argmax_merge(float& v, unsigned& i, float ov, unsigned oi)
{
    if (ov > v || (ov == v && oi < i))
        {
            v = ov;
            i = oi;
        }
}

// ---- the bin scan of acq.cc:417-426 / :463-474 and the statistic of :428-445 / :516 for one PRN, by ONE wave (t < 64): gmax starts at 0 and only a strictly
// larger row maximum replaces it, so ties keep the lowest bin.  row_of(d): the row's record (maximum, lowest arg-max, sum, second peak).
template <int S, class RowOf>
__device__ __forceinline__ void prn_statistic(const OcCellArgs& a, int prn, int t, RowStat* rs, RowOf row_of)
{
    float gmax = 0.0f;
    unsigned gbin = 0xFFFFFFFFu, gtau = 0u;
    for (int d = t; d < a.n_bins; d += 64)
        {
            const RowStat rw = row_of(d);
            if constexpr (S > 1) rs[d] = rw;
            if (rw.maxv > gmax)
                {
                    gmax = rw.maxv;
                    gbin = static_cast<unsigned>(d);
                    gtau = rw.idx;
                }
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        {
            const float ov = __shfl_down(gmax, off, 64);
            const unsigned ob = __shfl_down(gbin, off, 64);
            const unsigned ot = __shfl_down(gtau, off, 64);
            if (ov > gmax || (ov == gmax && ob < gbin))
                {
                    gmax = ov;
                    gbin = ob;
                    gtau = ot;
                }
        }
    if (t == 0)
        {
            if (gbin == 0xFFFFFFFFu)  // every row maximum was 0: the reference leaves bin 0 / index 0
                {
                    gbin = 0u;
                    gtau = 0u;
                }
            DevAcqResult out;
            out.index_time = gtau;
            out.index_doppler = gbin;
            out.peak = gmax;
            out.input_power = 0.0f;
            out.second_peak = 0.0f;
            out.test_statistics = 0.0f;
            if (a.use_cfar)
                {
                    // acq.cc:429-431: power of the bin half a grid away, / effective / 2 / dwells
                    const unsigned opp = (gbin + static_cast<unsigned>(a.n_bins) / 2u) % static_cast<unsigned>(a.n_bins);
                    const float per_sample = row_of(static_cast<int>(opp)).sum / static_cast<float>(static_cast<unsigned>(a.effective));
                    const float power = static_cast<float>(static_cast<double>(per_sample) / 2.0 / static_cast<double>(a.dwell_count));
                    out.input_power = power;
                    out.test_statistics = (power < 1.1920928955078125e-07f) ? 0.0f : gmax / power;  // acq.cc:438-445
                }
            else
                {
                    // (S > 1: no sub-cell sees the whole row; oc_second_peak_kernel, queued right behind this launch, scans the stored winning
                    // row and fills in second_peak / test_statistics)
                    const float second_pk = (S == 1) ? row_of(static_cast<int>(gbin)).second : 0.0f;
                    out.second_peak = second_pk;
                    out.test_statistics = (S == 1) ? gmax / second_pk : 0.0f;  // acq.cc:516
                }
            a.results[prn] = out;
        }
}

// ---- publish a row record; the LAST cell of its PRN to arrive forms the PRN's statistic (acq.cc:409-519).  Called by every thread of the work-group;
// `sum` / `second` are thread 0's.  S > 1: the record is one of the row's S sub-cell records (merged by the last arriver); S == 1: the whole row's.
// Placement-independent hand-off: plain store -> agent-scope release -> relaxed ticket; the last arriver acquires.
template <int S>
__device__ __forceinline__ void publish_row(const OcCellArgs& a, int prn, int cell, int r, float peak, unsigned tau, float sum, float second, unsigned* s_i)
{
    const int t = threadIdx.x;
    if (t == 0)
        {
            RowStat rec;
            rec.maxv = peak;
            rec.idx = tau;
            rec.sum = sum;
            rec.second = second;
            if constexpr (S == 1)
                a.rows[cell] = rec;
            else
                a.subrows[static_cast<size_t>(cell) * S + r] = rec;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned ticket = __hip_atomic_fetch_add(&a.arrivals[prn], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_i[0] = (ticket == static_cast<unsigned>(a.n_bins * S) - 1u) ? 1u : 0u;
        }
    __syncthreads();
    if (s_i[0] == 0u || t >= 64) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    {
        RowStat* rs = a.rows + static_cast<size_t>(prn) * a.n_bins;
        // S > 1: a row is the union of its S sub-cells' lags -- maximum with the lowest index among equals, sums added; the merged record is
        // also what gsh_acq_read_row_peaks hands out
        auto row_of = [&](int d) GSH_AI -> RowStat {
            if constexpr (S == 1)
                return rs[d];
            else
                {
                    const RowStat* __restrict__ sub = a.subrows + (static_cast<size_t>(prn) * a.n_bins + d) * S;
                    RowStat m = sub[0];
                    for (int q = 1; q < S; q++)
                        {
                            const RowStat o = sub[q];
                            argmax_merge(m.maxv, m.idx, o.maxv, o.idx);
                            m.sum += o.sum;
                        }
                    return m;
                }
        };
        prn_statistic<S>(a, prn, t, rs, row_of);
        if (t == 0) __hip_atomic_store(&a.arrivals[prn], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
}

// ---- the cells WITHOUT a hand-off (round 5; every flavour that does not want the second peak).  Until round 4 every cell ended with: barrier, thread 0 merges the
// sixteen waves' partial records, stores the row, waits for the store, agent-scope release, a ticket from an atomic counter, barrier -- 4 of a cell's 21 us with the
// whole compute unit waiting (profiles/oc_cell_annotated.txt), 1312 times per batch, to save one small launch.  Now a wave leaves its partial record (maximum, lowest
// arg-max, sum) in `waverows` and is done; this kernel, one wave per PRN, queued behind the cells, merges the partials -- waves in ascending order, the order of the
// reduction it replaces, so every sum is the same float -- and runs the bin scan and the statistic.  n_waves: waves per cell work-group; s_rt: sub-cells per row.
__global__ __launch_bounds__(64) void oc_rows_kernel(OcCellArgs a, int n_waves, int s_rt)
{
    const int prn = blockIdx.x, t = threadIdx.x;
    RowStat* rs = a.rows + static_cast<size_t>(prn) * a.n_bins;
    // sub-cell q of row d: the waves' partials merged in ascending wave order (thread 0 of the cell used to take its own wave's and add waves 1, 2, ... to it)
    auto sub_of = [&](int d, int q) GSH_AI -> RowStat {
        const RowStat* __restrict__ w = a.waverows + ((static_cast<size_t>(prn) * a.n_bins + d) * s_rt + q) * OC_MAX_WAVES;
        RowStat m = w[0];
        for (int k = 1; k < n_waves; k++)
            {
                const RowStat o = w[k];
                argmax_merge(m.maxv, m.idx, o.maxv, o.idx);
                m.sum += o.sum;
            }
        m.second = 0.0f;
        return m;
    };
    // a row is the union of its sub-cells' lags: maximum with the lowest index among equals, sums added in ascending sub-cell order
    auto row_of = [&](int d) GSH_AI -> RowStat {
        RowStat row = sub_of(d, 0);
        for (int q = 1; q < s_rt; q++)
            {
                const RowStat m = sub_of(d, q);
                argmax_merge(row.maxv, row.idx, m.maxv, m.idx);
                row.sum += m.sum;
            }
        return row;
    };
    for (int d = t; d < a.n_bins; d += 64)  // what gsh_acq_read_row_peaks (and, on split plans, the sub-cell records' readers) hand out
        {
            if (s_rt > 1)
                for (int q = 0; q < s_rt; q++) a.subrows[(static_cast<size_t>(prn) * a.n_bins + d) * s_rt + q] = sub_of(d, q);
            rs[d] = row_of(d);
        }
    // (the statistic forms its rows from the partials again rather than reading back what other lanes have just stored)
    prn_statistic<1>(a, prn, t, rs, row_of);
}

// ------------------------------------------------------------------------------------------------------------
// GRID: the magnitude grid is read (accumulate) and / or written (store_grid); SECOND: the peak-ratio statistic's
// second peak is wanted.  Both are compile-time so that the headline configuration (CFAR statistic, single dwell,
// no dump) carries neither the grid addressing nor the second scan in its register budget.
// DIT (round 6): the sub-cells of a decimation-in-time split (see oc_combine_dit_kernel below) as a flavour of THIS kernel -- sub-cell r of cell (prn, bin) is a plain
// M-point cell over residue class r of both spectra that stores its transform Z_r instead of searching it -- so that they get the pass loop, the idle waves' operand
// prefetch and the touch of the next bin spectrum instead of one freshly dispatched work-group per sub-cell (128 000 points: 603 us of a batch's 926 were sub-cells).
template <class P, int S, bool GRID, bool SECOND, bool OFF, bool DIT = false>
__global__ __launch_bounds__(P::THREADS) void oc_cell_kernel(OcCellArgs a)
{
    static_assert(!DIT || (S > 1 && !GRID && !SECOND && !OFF), "a decimation-in-time sub-cell only transforms");
    constexpr bool PLAIN = (S == 1) || DIT;  // the operands are one M-point slice of each spectrum, multiplied element by element
    static_assert(S == 1 || !SECOND, "the second peak of a row needs the whole row in one work-group");
    static_assert(!SECOND || P::N * 4 <= P::LDS_BYTES, "the peak-ratio flavour parks the row's N magnitudes in the exchange buffer: a plan whose buffer is smaller needs another place for them");
    static_assert(GRID || !OFF, "the upper-half searches are instantiated on the GRID flavour only");
    constexpr int M = P::N, N = S * P::N;
    GSH_OC_LDS_DECL(P);
    __shared__ float s_v[OC_MAX_WAVES];
    __shared__ unsigned s_i[OC_MAX_WAVES];
    __shared__ float s_s[OC_MAX_WAVES];
    __shared__ float s_peak;
    __shared__ unsigned s_tau;

    // ---- which cell: block b runs on XCD b % 8; each XCD owns a (PRN range) x (bin range) tile and walks it
    // PRN-fastest, so the cells in flight on one XCD share a few code spectra and a few bin spectra in its L2
    // (S > 1: the S sub-cells of a cell follow each other on the same XCD -- they read the same two spectra)
    const int xcd = static_cast<int>(blockIdx.x & 7u);
    if (a.stagger_groups > 1 && static_cast<int>(blockIdx.x) < a.stagger_first)  // uniform over the work-group
        {
            const unsigned long long until = wall_clock64() + static_cast<unsigned long long>((static_cast<int>(blockIdx.x >> 3) % a.stagger_groups) * a.stagger_ticks);
            while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
        }
    // several cells per work-group: where a work-group has its compute unit to itself (the plans with the large exchange buffer or 1 024 threads -- the same ones
    // that use the phased exchanges); the small plans keep several work-groups resident per unit, whose starts and ends overlap anyway, and their registers for occupancy.
    // (The peak-ratio flavours end in a hand-off that not every thread returns from: one cell per work-group.)
    constexpr bool PERSIST = !SECOND && P::EX64;
    const int passes = PERSIST ? a.cells_per_wg : 1;
    // The waves that have no stage-3 butterfly at all (radix 40: threads 640 .. 1 023, six of sixteen) load their OWN operands of the work-group's next cell while the
    // others transform: they enter the next pass with their stage-1 inputs in registers, and the pass's operand phase -- every unit of the chip asking the L2 at
    // once, a third of the cell -- is the other ten waves' only.  The pass loop exists twice, and a wave takes one or the other for the whole launch: the idle waves'
    // copy carries the 2 R1 product registers across the passes and has no stage 3 (nor its 2 R3 registers); the other copy is the loop as it was.  (One loop
    // with the array carried for every wave did not fit: the registers of a value that is written on one side of a branch and read a pass later are not given to
    // the other side's stage 3.)  Both copies meet the same barriers, pass for pass.
    constexpr int OPF_FIRST = (P::T3 + 63) / 64 * 64;  // first thread of the first wholly idle wave
    constexpr bool OPF = PERSIST && !GRID && PLAIN && OPF_FIRST + 64 <= P::THREADS;
    auto pass_loop = [&](auto IDLE_TAG) GSH_AI {
    constexpr bool IDLE = decltype(IDLE_TAG)::value;
    [[maybe_unused]] cf pa[IDLE ? P::R1 : 1];
    [[maybe_unused]] bool have_operands = false;
    if constexpr (IDLE) fresh_values<P::R1>(pa);
#pragma clang loop unroll(disable)
    for (int pass = 0; pass < passes; pass++)
    {
    // (the thread index passes through an empty asm in every pass: whatever is derived from it -- twiddle seeds, LDS addresses -- is formed afresh, as in a new work-group,
    // instead of being kept in registers across the passes: the cell needs every one of its 128 registers)
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    if constexpr (OPF) __builtin_assume(IDLE ? t >= OPF_FIRST : t < OPF_FIRST);
    const int slot_r = static_cast<int>(blockIdx.x >> 3) + pass * static_cast<int>(gridDim.x >> 3);
    if (slot_r >= a.slots_per_xcd) break;  // uniform over the work-group
    const int slot = slot_r / S, r = slot_r - slot * S;
    const int xp_i = xcd % a.xp, xb_i = xcd / a.xp;
    const int bl = slot / a.prn_per, pl = slot - bl * a.prn_per;
    const int prn = xp_i * a.prn_per + pl, bin = xb_i * a.bin_per + bl;
    if (prn >= a.n_prn || bin >= a.n_bins) continue;  // uniform over the work-group
    const int cell = prn * a.n_bins + bin;
    // Between two passes: the previous cell's last reads of the exchange buffer must have been consumed before this cell's first writes.  With the phased exchanges that
    // needs no barrier of its own when exchange 2 ENDS in the region exchange 1 does not START in (25 x 25 x 40: three phases each, regions 0 1 0 / 1 0 1): the reads
    // still in flight are of region 1, the first write goes to region 0 -- which nobody has read since the last barrier of exchange 2 --, and the barrier behind that
    // write is behind every wave's last read.  Without it a wave that is done with its stage 3 (the waves of a SIMD take turns: 4 600, 7 100, 9 400 clocks) starts on
    // the next cell's operands at once instead of waiting for the slowest (profiles/oc_cell_annotated.txt).
    constexpr bool PASS_BARRIER = !P::EX64 || ((P::NP2 - 1 + P::START2) % 2 == P::START1) || GSH_OC_PASS_BARRIER;
    if (PASS_BARRIER && pass > 0) __syncthreads();
    OC_STAMP_WALL(8);
    OC_STAMP(0);

    cf ra[P::R1], rb[P::R2], rc[P::R3];
    // The three stages' registers are written under `t < T1 / T2 / T3`; inside the pass loop a lane that does not take part keeps "whatever the register held", which
    // the register allocator reads as: the previous pass's values stay live round the back edge -- 180 registers nobody needs, i.e. scratch.  An empty asm that
    // OUTPUTS every element, placed right in front of the stage that writes them, ends those live ranges without an instruction (zero-filling them at the top of the
    // pass did too, and starved the operand loads of registers: 7.4 us instead of 1.9 for the 50 loads of a thread).
    if constexpr (PERSIST) fresh_values<P::R1>(ra);
    if (t < P::T1)
        {
#ifdef GSH_OC_PROFILE_SAME_BIN  /* (timing experiment only: every cell reads the spectrum of bin 0 / bin & 7 -- the results are wrong, the loads hit L2) */
            const cf* __restrict__ X = a.spectra + static_cast<size_t>(bin & (GSH_OC_PROFILE_SAME_BIN)) * N + t;
#else
            const cf* __restrict__ X = a.spectra + static_cast<size_t>(bin) * N + (DIT ? static_cast<size_t>(r) * M : 0) + t;
#endif
            const cf* __restrict__ C = a.codes + static_cast<size_t>(prn) * N + (DIT ? static_cast<size_t>(r) * M : 0) + t;
            if constexpr (PLAIN)
                {
                    // all 2 * R1 operand loads are issued before the first product: the cell's registers are still free here, and one round of
                    // L2 / Infinity-Cache latency is cheaper than the two the scheduler otherwise settles for (12 loads, wait, 38 loads)
                    if (IDLE && have_operands)  // (uniform over the wave)
                        oc::static_for<P::R1>([&](auto N1) GSH_AI { ra[decltype(N1)::value] = pa[IDLE ? decltype(N1)::value : 0]; });
                    else
                    {
                    cf xv[P::R1], cv[P::R1];
                    oc::static_for<P::R1>([&](auto N1) GSH_AI {
                        constexpr int n1 = decltype(N1)::value;
#ifdef GSH_OC_PROFILE_NO_LOADS  /* (timing experiment only: the operands are made up -- what the cell costs without its 400 KB) */
                        xv[n1] = cf{static_cast<float>(t) * 1e-3f, static_cast<float>(n1 + bin)};
                        cv[n1] = cf{static_cast<float>(n1 + prn), static_cast<float>(t) * 1e-3f};
#else
                        xv[n1] = X[n1 * P::T1];
                        cv[n1] = C[n1 * P::T1];
#endif
                    });
                    __builtin_amdgcn_sched_group_barrier(0x20, 2 * P::R1, 0);  // VMEM reads first ...
                    oc::static_for<P::R1>([&](auto N1) GSH_AI {
                        constexpr int n1 = decltype(N1)::value;
                        ra[n1] = oc::cmul_conj(xv[n1], cv[n1]);
                    });
                    __builtin_amdgcn_sched_group_barrier(0x2, 8 * P::R1, 0);   // ... then the products
                    }
#ifdef GSH_OC_PROFILE
                    asm volatile("" ::"v"(ra[P::R1 - 1].x), "v"(ra[0].x));  // (the stamp below must not be scheduled above the last product)
#endif
                    OC_STAMP(1);
                }
            else
                {
                    // a_r[n] = ( sum_q Y[n + q M] W_S^{q r} ) W_N^{n r},  Y = conj(X) C
                    // (q = 1 .. S-1 as a rolled loop: unrolled, the scheduler hoists all 2 S R1 loads and spills)
                    oc::static_for<P::R1>([&](auto N1) GSH_AI {
                        constexpr int n1 = decltype(N1)::value;
                        ra[n1] = oc::cmul_conj(X[n1 * P::T1], C[n1 * P::T1]);
                    });
#pragma clang loop unroll(disable)
                    for (int q = 1; q < S; q++)
                        {
                            const cf cq = radix_root(q * r, S);
                            const cf* __restrict__ Xq = X + static_cast<size_t>(q) * M;
                            const cf* __restrict__ Cq = C + static_cast<size_t>(q) * M;
                            // (radix 32: the 32 running sums take half the register file; the operand loads of this pass are issued eight elements at a
                            // time -- the offset of the next eight passes through an empty asm together with the sum the previous eight ended on)
                            int kq = 0;
                            oc::static_for<P::R1>([&](auto N1) GSH_AI {
                                constexpr int n1 = decltype(N1)::value;
                                const cf u = oc::cmul(oc::cmul_conj(Xq[n1 * P::T1 + kq], Cq[n1 * P::T1 + kq]), cq);
                                ra[n1].x += u.x;
                                ra[n1].y += u.y;
                                if constexpr (P::R1 >= 32 && n1 % 8 == 7) asm volatile("" : "+v"(kq) : "v"(ra[n1].x));
                            });
                        }
                    if (r != 0)  // uniform over the work-group
                        {
                            oc::mul_powers<P::R1>(ra, oc::unit_root((P::T1 * r) % N, N));
                            const cf seed = oc::unit_root((t * r) % N, N);
                            oc::static_for<P::R1>([&](auto N1) GSH_AI { ra[decltype(N1)::value] = oc::cmul(ra[decltype(N1)::value], seed); });
                        }
                }
            P::stage1(ra, t);
#ifdef GSH_OC_PROFILE
            asm volatile("" ::"v"(ra[P::R1 - 1].x), "v"(ra[0].x));
#endif
        }
    OC_STAMP(2);
    exchange1<P, PERSIST>(ra, rb, t, lds);
#ifdef GSH_OC_PROFILE
    asm volatile("" ::"v"(rb[P::R2 - 1].x), "v"(rb[0].x));
#endif
    OC_STAMP(3);
    if (t < P::T2) P::stage2(rb, t);
#ifdef GSH_OC_PROFILE
    asm volatile("" ::"v"(rb[P::R2 - 1].x), "v"(rb[0].x));
#endif
    OC_STAMP(4);
    exchange2<P, PERSIST>(rb, rc, t, lds);
#ifdef GSH_OC_PROFILE
    asm volatile("" ::"v"(rc[P::R3 - 1].x), "v"(rc[0].x));
#endif
    OC_STAMP(5);

    // SECOND: the row's magnitudes are parked in the exchange buffer (N floats fit: it held N complex values a phase at a time) until the row's
    // peak is known; every exchange-2 read of every thread must have landed before the first magnitude overwrites it
    float* mag = reinterpret_cast<float*>(lds);
    if constexpr (SECOND) __syncthreads();
    // ---- |.|^2, optional accumulation / grid store, per-thread (max, lowest arg-max, sum)
    // element k3 of thread t is lag tau = S (t + T3 k3) + r of the length-N correlation; it enters the search as index tau - offset when
    // tau >= offset (offset + effective == N; offset != 0 only for bit_transition_flag, acq.cc:544)
    float best = -1.0f, sum = 0.0f;
    unsigned at = 0xFFFFFFFFu;
    // The threads that have no stage-3 butterfly (radix 40: 375 of them) touch the bin spectrum of the work-group's NEXT cell, one word per 64 bytes: an XCD's
    // 4 MB of L2 hold a round's 8 bin spectra and 4 code spectra, not the search's 41 -- every new bin comes over the fabric (2.4 of the operand phase's 7.4 us,
    // measured by letting every cell read bin 0: profiles/oc_cell_annotated.txt), and here it comes while the other threads transform.
#ifndef GSH_OC_DIT_TOUCH
#define GSH_OC_DIT_TOUCH 0  // (A/B) the touch on decimation-in-time sub-cells: the whole next bin spectrum is five times what the next sub-cell reads
#endif
    constexpr bool TOUCH = PERSIST && !GRID && PLAIN && (!DIT || GSH_OC_DIT_TOUCH) && P::T3 + 64 <= P::THREADS;
    const int nslot_r = slot_r + static_cast<int>(gridDim.x >> 3);
    const bool next_pass = (TOUCH || OPF) && pass + 1 < passes && nslot_r < a.slots_per_xcd;  // uniform over the work-group
    if constexpr (IDLE)
        {
            // this wave's own operands of the next cell (every lane loads and multiplies -- the lanes past T1, which have no stage-1 butterfly, the last butterfly's
            // operands over again)
            const int nslot = nslot_r / S, nr = nslot_r - nslot * S;
            const int nbl = nslot / a.prn_per, npl = nslot - nbl * a.prn_per;
            const int nprn = xp_i * a.prn_per + npl, nbin_own = xb_i * a.bin_per + nbl;
            have_operands = next_pass && a.prefetch_next >= 2 && nprn < a.n_prn && nbin_own < a.n_bins;
            if (have_operands)
                {
                    const int tt = t < P::T1 ? t : P::T1 - 1;
                    const cf* __restrict__ X = a.spectra + static_cast<size_t>(nbin_own) * N + (DIT ? static_cast<size_t>(nr) * M : 0) + tt;
                    const cf* __restrict__ C = a.codes + static_cast<size_t>(nprn) * N + (DIT ? static_cast<size_t>(nr) * M : 0) + tt;
                    cf xv[P::R1], cv[P::R1];
                    oc::static_for<P::R1>([&](auto N1) GSH_AI {
                        constexpr int n1 = decltype(N1)::value;
                        xv[n1] = X[n1 * P::T1];
                        cv[n1] = C[n1 * P::T1];
                    });
                    __builtin_amdgcn_sched_group_barrier(0x20, 2 * P::R1, 0);
                    oc::static_for<P::R1>([&](auto N1) GSH_AI {
                        constexpr int n1 = decltype(N1)::value;
                        pa[n1] = oc::cmul_conj(xv[n1], cv[n1]);
                    });
                    __builtin_amdgcn_sched_group_barrier(0x2, 8 * P::R1, 0);
                }
            else
                fresh_values<P::R1>(pa);  // (nothing is carried: says so to the register allocator)
        }
    if constexpr (TOUCH)
        if (t >= P::T3 && a.prefetch_next && next_pass)
            {
                const int nbin = xb_i * a.bin_per + (nslot_r / S) / a.prn_per;
                if (nbin < a.n_bins && nbin != bin)
                    {
                        // (the prn_per work-groups that go to that bin next take every prn_per-th line each)
                        constexpr int IDLE_THREADS = P::THREADS - P::T3, LINES = N * 8 / 64;
                        const int share = a.prn_per > 4 ? 4 : a.prn_per, mine = (nslot_r / S) % share;
                        const float* __restrict__ base = reinterpret_cast<const float*>(a.spectra + static_cast<size_t>(nbin) * N);
                        float acc = 0.0f;
                        constexpr int PER = (LINES + IDLE_THREADS - 1) / IDLE_THREADS;  // enough for share == 1; fewer lines each when the bin is shared
                        oc::static_for<PER>([&](auto K) GSH_AI {
                            const int line = ((t - P::T3) + decltype(K)::value * IDLE_THREADS) * share + mine;
                            if (decltype(K)::value * IDLE_THREADS * share < LINES) acc += base[static_cast<size_t>(line < LINES ? line : mine) * 16];
                        });
                        asm volatile("" ::"v"(acc));  // (the loads are real: their data is waited for, by threads that have nothing else to do)
                    }
            }
    if constexpr (DIT)
        {
            // the sub-cell's whole result: Z_r, for oc_combine_dit_kernel (queued behind this launch)
            if (!IDLE && t < P::T3)
                {
                    P::stage3(rc);
                    cf* __restrict__ zo = a.z + static_cast<size_t>(cell) * N + static_cast<size_t>(r) * M + t;
                    oc::static_for<P::R3>([&](auto K3) GSH_AI { zo[decltype(K3)::value * P::T3] = rc[decltype(K3)::value]; });
                }
            OC_STAMP(7);
            OC_STAMP_WALL(9);
            continue;
        }
    if (!IDLE && t < P::T3)
        {
            P::stage3(rc);
            float* __restrict__ g = a.grid + static_cast<size_t>(cell) * a.effective;
            {
                // OFF: offset == N / 2 == S T3 R3 / 2.  Element k3 is valid for k3 > K0, invalid for k3 < K0, and element K0 is valid for every
                // thread when R3 is even (K0 = R3 / 2) or for the threads with 2 (S t + r) >= S T3 when R3 is odd (K0 = (R3 - 1) / 2)
                constexpr int K0 = OFF ? P::R3 / 2 : 0;
                constexpr bool MIXED = OFF && (P::R3 % 2 == 1);
                const bool edge_ok = !MIXED || (2 * (S * t + r) >= S * P::T3);
                const int base = S * t + r - (OFF ? a.offset : 0);  // index of element k3 = base + S T3 k3
                if (GRID)
                    {
                        oc::static_for<P::R3>([&](auto K3) GSH_AI {
                            constexpr int k3 = decltype(K3)::value;
                            if constexpr (k3 >= K0) rc[k3].x = oc::norm2(rc[k3]) * a.weight;
                        });
                        if (a.accumulate)  // acq.cc:549-553
                            oc::static_for<P::R3>([&](auto K3) GSH_AI {
                                constexpr int k3 = decltype(K3)::value;
                                if constexpr (k3 > K0 || (k3 == K0 && !MIXED)) rc[k3].x += g[base + S * P::T3 * k3];
                                if constexpr (k3 == K0 && MIXED)
                                    if (edge_ok) rc[k3].x += g[base + S * P::T3 * k3];
                                // (the sub-cell and second-peak flavours sit at the 128-register limit: eight grid loads in flight at a time instead of all R3)
                                if constexpr ((S > 1 || SECOND) && k3 % 8 == 7) __builtin_amdgcn_sched_barrier(0);
                            });
                        if (a.store_grid)
                            oc::static_for<P::R3>([&](auto K3) GSH_AI {
                                constexpr int k3 = decltype(K3)::value;
                                if constexpr (k3 > K0 || (k3 == K0 && !MIXED)) g[base + S * P::T3 * k3] = rc[k3].x;
                                if constexpr (k3 == K0 && MIXED)
                                    if (edge_ok) g[base + S * P::T3 * k3] = rc[k3].x;
                            });

                    }
                int k_best = -1;
                oc::static_for<P::R3>([&](auto K3) GSH_AI {
                    constexpr int k3 = decltype(K3)::value;
                    if constexpr (k3 >= K0)
                        {
                            float m = GRID ? rc[k3].x : oc::norm2(rc[k3]);
                            if constexpr (k3 == K0 && MIXED) m = edge_ok ? m : -1.0f;  // never better than `best`
                            // branch-free bookkeeping (selects): k3 ascending = index ascending, so '>' keeps the lowest index
                            if constexpr (k3 == K0 && MIXED)
                                sum += edge_ok ? m : 0.0f;
                            else
                                sum += m;
                            const bool better = m > best;
                            best = better ? m : best;
                            k_best = better ? k3 : k_best;  // (the element, an inline constant; its index is formed once, below -- not one add per element)
                            if (SECOND) mag[t + P::T3 * k3] = m;  // kept for the second scan -- in the exchange buffer, idle from here on, not in 40 registers
                        }
                    else if (SECOND)
                        mag[t + P::T3 * k3] = -1.0f;  // lags below the offset: never the second peak either
                });
                at = k_best < 0 ? 0xFFFFFFFFu : static_cast<unsigned>(base + S * P::T3 * k_best);
            }
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        {
            const float ov = __shfl_down(best, off, 64);
            const unsigned oi = __shfl_down(at, off, 64);
            sum += __shfl_down(sum, off, 64);
            argmax_merge(best, at, ov, oi);
        }
    const int wave = t >> 6;
    constexpr int n_waves = P::THREADS / 64;
    static_assert(n_waves <= OC_MAX_WAVES, "work-group too large for the reduction scratch");
#ifdef GSH_OC_PROFILE
    asm volatile("" ::"v"(best), "v"(sum));
#endif
    OC_STAMP(6);
    if constexpr (!SECOND)
        {
            // the wave's partial record, and the wave is done: no barrier, no hand-off -- oc_rows_kernel, queued behind this launch, merges and decides
            if ((t & 63) == 0)
                {
                    RowStat rec;
                    rec.maxv = best;
                    rec.idx = at;
                    rec.sum = sum;
                    rec.second = 0.0f;
                    a.waverows[(static_cast<size_t>(cell) * S + r) * OC_MAX_WAVES + wave] = rec;
                }
            OC_STAMP(7);
            OC_STAMP_WALL(9);
            continue;
        }
    if ((t & 63) == 0)
        {
            s_v[wave] = best;
            s_i[wave] = at;
            s_s[wave] = sum;
        }
    __syncthreads();
    if (t == 0)
        {
            for (int w = 1; w < n_waves; w++)
                {
                    argmax_merge(best, at, s_v[w], s_i[w]);
                    sum += s_s[w];
                }
            s_peak = best;
            s_tau = at;
        }
    float second = 0.0f;  // blanked cells hold 0.0
    if (SECOND)
        {

    // ---- second peak of THIS row with +-samples_per_chip around its own peak blanked (acq.cc:485-513); the final
    // scan uses the record of the winning row, whose peak is the global one
            __syncthreads();
            const int tau_pk = static_cast<int>(s_tau);
            int e1 = tau_pk - a.samples_per_chip;
            int e2 = tau_pk + a.samples_per_chip;
            if (e1 < 0)
                e1 += a.effective;
            else if (e2 >= a.effective)
                e2 -= a.effective;
            const bool wraps = e1 > e2, blank_all = e1 == e2;  // the do-while of acq.cc:498-509 blanks everything when e1 == e2
            if (t < P::T3)
                {
                    oc::static_for<P::R3>([&](auto K3) GSH_AI {
                        constexpr int k3 = decltype(K3)::value;
                        const int tau = t + P::T3 * k3 - a.offset;   // S == 1 here; lags below the offset hold -1 and never win against 0.0
                        const bool ge1 = tau >= e1, lt2 = tau < e2;
                        const bool blank = blank_all | (wraps ? (ge1 | lt2) : (ge1 & lt2));
                        second = blank ? second : fmaxf(second, mag[t + P::T3 * k3]);
                    });
                }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) second = fmaxf(second, __shfl_down(second, off, 64));
            __syncthreads();  // s_v is reused
            if ((t & 63) == 0) s_v[wave] = second;
            __syncthreads();
            if (t == 0)
                for (int w = 1; w < n_waves; w++) second = fmaxf(second, s_v[w]);
        }

    publish_row<S>(a, prn, cell, r, s_peak, s_tau, sum, second, s_i);
    OC_STAMP(7);
    OC_STAMP_WALL(9);
    }  // pass
    };  // pass_loop
    if constexpr (OPF)
        {
            if (__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x)) >= OPF_FIRST)
                pass_loop(std::true_type{});
            else
                pass_loop(std::false_type{});
        }
    else
        pass_loop(std::false_type{});
}

// ---- N = S * M, decimation in time (round 3).  The sub-cells of oc_cell_kernel<P, S> (decimation in frequency) each read the WHOLE product spectrum --
// S times the loads, which is what bounds S >= 4.  Here the spectra are stored residue-major (OcFwdArgs::residue_major: class r of
// X at [r M, (r + 1) M)), and with Y = conj(X) C,  y[tau] = sum_k Y[k] W_N^{k tau},  k = S k' + r:
//     y[t' + M j] = sum_r ( Z_r[t'] W_N^{r t'} ) W_S^{r j},      Z_r = FFT_M( Y[S k' + r] over k' ).
// oc_subcell_dit_kernel: sub-cell r reads ONLY residue class r of both spectra (2 M values: what a plain M-point cell reads), runs the unchanged three-stage
// plan and leaves Z_r in `z`.  oc_combine_dit_kernel, the next launch on the stream: one work-group per cell -- per t' the S values, their twiddles, one
// S-point DFT in registers, |.|^2 of the S lags t' + M j, the row's statistics -- publishes the row like a plain cell.
// (Two launches, not a last-arriver hand-off inside one: the sub-cells of a cell run on different compute units, and making Z_r visible between them costs
// an agent-scope release per sub-cell and an acquire per cell -- L2 write-backs and invalidations that the whole XCD pays for: 3.7 ms instead of 1.44 ms at
// 128 000 points, profiles/ab/r03/acq_dit.txt.  The kernel boundary does the same for nothing.)
#ifndef GSH_OC_DIT_R_MAJOR
#define GSH_OC_DIT_R_MAJOR -1  // OcCellArgs::dit_r_major: -1 = chosen per split (launch_cells_dit), 0 / 1 = never / always
#endif
#ifndef GSH_OC_Z_NT_BELOW_S
#define GSH_OC_Z_NT_BELOW_S 1024  // the splits into fewer sub-cells than this get the hints: all of them (8 leaves S = 8 out -- A/B builds; onchip_dit_nontemporal below has the measurements)
#endif
#ifndef GSH_OC_Z_NT_STORE
#define GSH_OC_Z_NT_STORE 1  // the sub-cells' Z stores carry the non-temporal hint (0: A/B builds)
#endif
#ifndef GSH_OC_Z_NT_LOAD
#define GSH_OC_Z_NT_LOAD 1  // the combine launch's Z loads carry the non-temporal hint (0: A/B builds)
#endif
template <class P, int S>
__global__ __launch_bounds__(P::THREADS) void oc_subcell_dit_kernel(OcCellArgs a)
{
    constexpr int M = P::N, N = S * P::N;
    GSH_OC_LDS_DECL(P);
    const int xcd = static_cast<int>(blockIdx.x & 7u), slot_r = static_cast<int>(blockIdx.x >> 3);
    // the order in which an XCD's compute units meet its sub-cells decides what its 4 MB of L2 must hold: cell by cell (r fastest) the 32 sub-cells in flight read
    // ~10 bin and 20 code residue classes of 8 M bytes each (6 MB at M = 25 600); class by class (r slowest) 8 bin classes and the XCD's 4 code classes (2.5 MB)
    const int cells_per_xcd = a.prn_per * a.bin_per;
    const int slot = a.dit_r_major ? slot_r % cells_per_xcd : slot_r / S, r = a.dit_r_major ? slot_r / cells_per_xcd : slot_r - slot * S;
    const int xp_i = xcd % a.xp, xb_i = xcd / a.xp;
    const int bl = slot / a.prn_per, pl = slot - bl * a.prn_per;
    const int prn = xp_i * a.prn_per + pl, bin = xb_i * a.bin_per + bl;
    if (prn >= a.n_prn || bin >= a.n_bins) return;  // uniform over the work-group
    const int cell = prn * a.n_bins + bin;
    const int t = threadIdx.x;
    cf ra[P::R1], rb[P::R2], rc[P::R3];
    if (t < P::T1)
        {
            const cf* __restrict__ X = a.spectra + static_cast<size_t>(bin) * N + static_cast<size_t>(r) * M + t;
            const cf* __restrict__ C = a.codes + static_cast<size_t>(prn) * N + static_cast<size_t>(r) * M + t;
            cf xv[P::R1], cv[P::R1];
            oc::static_for<P::R1>([&](auto N1) GSH_AI {
                constexpr int n1 = decltype(N1)::value;
                xv[n1] = X[n1 * P::T1];
                cv[n1] = C[n1 * P::T1];
            });
            __builtin_amdgcn_sched_group_barrier(0x20, 2 * P::R1, 0);  // VMEM reads first ...
            oc::static_for<P::R1>([&](auto N1) GSH_AI {
                constexpr int n1 = decltype(N1)::value;
                ra[n1] = oc::cmul_conj(xv[n1], cv[n1]);
            });
            __builtin_amdgcn_sched_group_barrier(0x2, 8 * P::R1, 0);   // ... then the products
            P::stage1(ra, t);
        }
    exchange1<P>(ra, rb, t, lds);
    if (t < P::T2) P::stage2(rb, t);
    exchange2<P>(rb, rc, t, lds);
    if (t < P::T3)
        {
            P::stage3(rc);
            cf* __restrict__ zo = a.z + static_cast<size_t>(cell) * N + static_cast<size_t>(r) * M + t;
            // Z is written once here and read once by the combine launch: with the non-temporal hint it does not push the spectra every sub-cell re-reads out of
            // the L2 and the Infinity Cache (128 000 points: 0.924 -> 0.82 ms per batch, profiles/ab/r06/session43-45.txt)
            // (chosen at compile time: behind a run-time condition the two stores are merged into one and the hint is lost -- session 46)
            if constexpr (GSH_OC_Z_NT_STORE && S < GSH_OC_Z_NT_BELOW_S)
                oc::static_for<P::R3>([&](auto K3) GSH_AI { __builtin_nontemporal_store(rc[decltype(K3)::value], &zo[decltype(K3)::value * P::T3]); });
            else
                oc::static_for<P::R3>([&](auto K3) GSH_AI { zo[decltype(K3)::value * P::T3] = rc[decltype(K3)::value]; });
        }
}

constexpr int OC_COMBINE_THREADS = 1024;
#ifndef GSH_OC_COMBINE_HANDOFF
#define GSH_OC_COMBINE_HANDOFF 0  // 1: every work-group of the combine launch ends with the store - release - ticket hand-off, the last one forms the statistic (until round 6: A/B builds)
#endif
#ifndef GSH_OC_COMBINE_PARTS
#define GSH_OC_COMBINE_PARTS 1  // work-groups per cell of the combine launch (GSH_OC_COMBINE_PARTS in the environment overrides)
#endif
#ifndef GSH_OC_COMBINE_WG
#define GSH_OC_COMBINE_WG 1024  // threads per work-group of the combine launch (GSH_OC_COMBINE_THREADS in the environment overrides)
#endif
#ifndef GSH_OC_COMBINE_PAIRS
#define GSH_OC_COMBINE_PAIRS 1  // 0: one lag class per thread and trip, 8-byte loads, nothing in flight across trips (the round-3 form: A/B builds)
#endif
// Round 6: the combine step was bound by LATENCY, not by bandwidth (346 us of a 128 000-point batch's 926: 1.34 GB of Z at 3.9 TB/s, the device streams 6.1): a
// thread's trip was S 8-byte loads, a wait for all of them, and the arithmetic -- 25 dependent round trips to HBM per work-group.  Now a thread owns the PAIR of lags
// (2 i, 2 i + 1) of every residue class per trip (16-byte loads: a wave reads 1 KiB per instruction) and the next trip's S loads are issued BEFORE the current
// trip's arithmetic (a two-deep register queue), so a compute unit always has loads in flight.  The twiddles of the pair's second lag are the first's times the
// constant W_N^r; the trackers still meet the lags of one j in ascending order (lowest index wins ties).
template <int M, int S, bool GRID, bool OFF, int H>
__global__ __launch_bounds__(OC_COMBINE_THREADS) void oc_combine_dit_kernel(OcCellArgs a)
{
    static_assert(GRID || !OFF, "the upper-half searches are instantiated on the GRID flavour only");
    static_assert(H == 1 || GSH_OC_COMBINE_PAIRS, "parts of a cell are cut from its pairs of lag classes");
    constexpr int N = S * M;
#if GSH_OC_COMBINE_HANDOFF
    __shared__ float s_v[OC_MAX_WAVES];
    __shared__ unsigned s_i[OC_MAX_WAVES];
    __shared__ float s_s[OC_MAX_WAVES];
#endif
    const int cell = static_cast<int>(blockIdx.x) / H, part = static_cast<int>(blockIdx.x) - cell * H;
    const int prn = cell / a.n_bins;
    const int t = threadIdx.x;
    float best = -1.0f, sum = 0.0f;
    unsigned at = 0xFFFFFFFFu;
    {
        const cf* __restrict__ zc = a.z + static_cast<size_t>(cell) * N;
        float* __restrict__ g = a.grid + static_cast<size_t>(cell) * a.effective;
        const int offset = OFF ? a.offset : 0;
        // one (maximum, index) tracker per j: a thread meets the lags of one j in ascending order, so a strict '>' keeps the lowest index
        float bj[S];
        unsigned aj[S];
        oc::static_for<S>([&](auto J) GSH_AI {
            bj[decltype(J)::value] = -1.0f;
            aj[decltype(J)::value] = 0xFFFFFFFFu;
        });
        // the S-point DFT of one lag class m (u[r] = Z_r[m] already times W_N^{r m}): |.|^2 of the lags m + M j into the grid, the sum and the trackers
        auto lags_of = [&](cf (&u)[S], int m) GSH_AI {
            oc::Dft<S>::run(u);  // u[j] = y[m + M j]
            oc::static_for<S>([&](auto J) GSH_AI {
                constexpr int j = decltype(J)::value;
                const int idx = m + M * j - offset;  // the lag's index in the search (acq.cc:544)
                float v = oc::norm2(u[j]);
                const bool in = !OFF || idx >= 0;
                if constexpr (GRID)
                    {
                        v *= a.weight;
                        if (a.accumulate && in) v += g[idx];  // acq.cc:549-553
                        if (a.store_grid && in) g[idx] = v;
                    }
                if constexpr (OFF) v = in ? v : -1.0f;  // never better than a tracker's start value
                sum += (OFF && !in) ? 0.0f : v;
                const bool better = v > bj[j];
                bj[j] = better ? v : bj[j];
                aj[j] = better ? static_cast<unsigned>(idx) : aj[j];
            });
        };
#if GSH_OC_COMBINE_PAIRS
        static_assert(M % 2 == 0, "pairs of lag classes");
        // part `part` of the cell's H: the pairs [lo, hi) (H > 1: the combine launch's work-groups are 1 / H of a cell's work, so that its last, partly filled
        // round of work-groups on the device's compute units is 1 / H as long -- round 6, session 41)
        constexpr int PAIRS_ALL = M / 2, PER_PART = (PAIRS_ALL + H - 1) / H;
        const int STRIDE = static_cast<int>(blockDim.x);
        const int lo = part * PER_PART, PAIRS = (lo + PER_PART < PAIRS_ALL) ? lo + PER_PART : PAIRS_ALL;
        // twiddles W_N^{r m} at m = 2 (lo + t) (exact seeds), the pair's second lag is one W_N^r further, a trip 2 THREADS lags further
        cf tw[S], one[S], step[S];
        oc::static_for<S>([&](auto R) GSH_AI {
            constexpr int rr = decltype(R)::value;
            tw[rr] = oc::unit_root(static_cast<int>((static_cast<long long>(rr) * 2 * (lo + t)) % N), N);
            one[rr] = oc::unit_root(rr % N, N);
            step[rr] = oc::unit_root(static_cast<int>((static_cast<long long>(rr) * 2 * STRIDE) % N), N);
        });
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 q[S];
        constexpr bool nt = GSH_OC_Z_NT_LOAD && S < GSH_OC_Z_NT_BELOW_S;
        auto load = [&](int i) GSH_AI {
            oc::static_for<S>([&](auto R) GSH_AI {
                const f4* src = reinterpret_cast<const f4*>(zc + static_cast<size_t>(decltype(R)::value) * M + 2 * i);
                if constexpr (nt)
                    q[decltype(R)::value] = __builtin_nontemporal_load(src);
                else
                    q[decltype(R)::value] = *src;
            });
        };
        int i = lo + t;
        if (i < PAIRS) load(i);
#pragma clang loop unroll(disable)
        for (; i < PAIRS; i += STRIDE)
            {
                cf u0[S], u1[S];
                oc::static_for<S>([&](auto R) GSH_AI {
                    constexpr int rr = decltype(R)::value;
                    u0[rr] = cf{q[rr][0], q[rr][1]};
                    u1[rr] = cf{q[rr][2], q[rr][3]};
                });
                if (i + STRIDE < PAIRS) load(i + STRIDE);  // the next trip's loads fly during this trip's arithmetic
                oc::static_for<S>([&](auto R) GSH_AI {
                    constexpr int rr = decltype(R)::value;
                    if constexpr (rr > 0)
                        {
                            u0[rr] = oc::cmul(u0[rr], tw[rr]);                       // Z_r[m] W_N^{r m}
                            u1[rr] = oc::cmul(u1[rr], oc::cmul(tw[rr], one[rr]));     // Z_r[m + 1] W_N^{r (m + 1)}
                            tw[rr] = oc::cmul(tw[rr], step[rr]);
                        }
                });
                lags_of(u0, 2 * i);
                lags_of(u1, 2 * i + 1);
            }
#else
        // twiddles W_N^{r m}, r = 1 .. S-1, for m = t, t + THREADS, ...: exact seeds, then one complex product per step (at most M / THREADS steps: 25)
        cf tw[S], step[S];
        oc::static_for<S>([&](auto R) GSH_AI {
            constexpr int rr = decltype(R)::value;
            tw[rr] = oc::unit_root(static_cast<int>((static_cast<long long>(rr) * t) % N), N);
            step[rr] = oc::unit_root(static_cast<int>((static_cast<long long>(rr) * static_cast<int>(blockDim.x)) % N), N);
        });
#pragma clang loop unroll(disable)
        for (int m = t; m < M; m += static_cast<int>(blockDim.x))
            {
                cf u[S];
                oc::static_for<S>([&](auto R) GSH_AI { u[decltype(R)::value] = zc[static_cast<size_t>(decltype(R)::value) * M + m]; });
                oc::static_for<S>([&](auto R) GSH_AI {
                    constexpr int rr = decltype(R)::value;
                    if constexpr (rr > 0)
                        {
                            u[rr] = oc::cmul(u[rr], tw[rr]);  // u[r] *= W_N^{r m}
                            tw[rr] = oc::cmul(tw[rr], step[rr]);
                        }
                });
                lags_of(u, m);
            }
#endif
        oc::static_for<S>([&](auto J) GSH_AI { argmax_merge(best, at, bj[decltype(J)::value], aj[decltype(J)::value]); });
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        {
            const float ov = __shfl_down(best, off, 64);
            const unsigned oi = __shfl_down(at, off, 64);
            sum += __shfl_down(sum, off, 64);
            argmax_merge(best, at, ov, oi);
        }
    const int wave = t >> 6;
#if GSH_OC_COMBINE_HANDOFF
    const int n_waves = static_cast<int>(blockDim.x) / 64;
    if ((t & 63) == 0)
        {
            s_v[wave] = best;
            s_i[wave] = at;
            s_s[wave] = sum;
        }
    __syncthreads();
    if (t == 0)
        for (int w = 1; w < n_waves; w++)
            {
                argmax_merge(best, at, s_v[w], s_i[w]);
                sum += s_s[w];
            }
    __syncthreads();  // s_i is reused by publish_row
    publish_row<H>(a, prn, cell, part, best, at, sum, 0.0f, s_i);
#else
    // no hand-off at the end of a work-group (round 6, session 41 -- what round 5 did for the plain cells): a wave leaves its partial record and is done; oc_rows_kernel,
    // queued behind this launch, merges waves and parts in ascending order (the order of the reduction it replaces: every sum the same float) and forms the statistic
    (void)prn;
    if ((t & 63) == 0)
        {
            RowStat rec;
            rec.maxv = best;
            rec.idx = at;
            rec.sum = sum;
            rec.second = 0.0f;
            a.waverows[(static_cast<size_t>(cell) * H + part) * OC_MAX_WAVES + wave] = rec;
        }
#endif
}

// ---- first_vs_second_peak_statistic on a split plan (acq.cc:485-516).  The S sub-cells of a row each own every S-th lag, so none of them can blank
// +-samples_per_chip around the row's peak; the rows are kept in the magnitude grid instead and, once the cell launch has formed every PRN's peak, one
// work-group per PRN scans the winning row (a few hundred KB, 1 024 threads) and completes the record.
struct OcSecondArgs
{
    const float* grid;      // n_prn * n_bins * effective
    DevAcqResult* results;  // n_prn: index_time / index_doppler / peak set by the cell launch
    int n_bins, effective, samples_per_chip;
};

__global__ __launch_bounds__(1024) void oc_second_peak_kernel(OcSecondArgs a)
{
    __shared__ float s_m[16];
    const int prn = blockIdx.x, t = threadIdx.x;
    const DevAcqResult r = a.results[prn];
    const int tau_pk = static_cast<int>(r.index_time);
    int e1 = tau_pk - a.samples_per_chip;
    int e2 = tau_pk + a.samples_per_chip;
    if (e1 < 0)
        e1 += a.effective;
    else if (e2 >= a.effective)
        e2 -= a.effective;
    const bool wraps = e1 > e2, blank_all = e1 == e2;  // the do-while of acq.cc:498-509 blanks everything when e1 == e2
    const float* __restrict__ row = a.grid + (static_cast<size_t>(prn) * a.n_bins + r.index_doppler) * a.effective;
    float second = 0.0f;  // blanked cells hold 0.0
    for (int tau = t; tau < a.effective; tau += 1024)
        {
            const bool ge1 = tau >= e1, lt2 = tau < e2;
            const bool blank = blank_all | (wraps ? (ge1 | lt2) : (ge1 & lt2));
            second = blank ? second : fmaxf(second, row[tau]);
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) second = fmaxf(second, __shfl_down(second, off, 64));
    if ((t & 63) == 0) s_m[t >> 6] = second;
    __syncthreads();
    if (t == 0)
        {
            for (int w = 1; w < 16; w++) second = fmaxf(second, s_m[w]);
            a.results[prn].second_peak = second;
            a.results[prn].test_statistics = r.peak / second;  // acq.cc:516
        }
}

template <class P>
int launch_forward(const OcFwdArgs& a, int batch, hipStream_t s)
{
    hipLaunchKernelGGL((oc_forward_kernel<P>), dim3(batch), dim3(P::THREADS), 0, s, a);
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}

template <class P, int S>
int launch_forward_split(const OcFwdArgs& a, int batch, hipStream_t s)
{
    GSH_REQUIRE(a.fold <= 1, "folded searches have no split plan");
    hipLaunchKernelGGL((oc_forward_split_kernel<P, S>), dim3(batch * S), dim3(P::THREADS), 0, s, a);
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}

// GSH_OC_CELLS_PER_WG in the environment: cells one work-group of the cell kernel carries out in turn (flavours without the in-kernel hand-off)
int onchip_cells_per_wg()
{
    static const int v = [] {
        const char* e = std::getenv("GSH_OC_CELLS_PER_WG");
        return e != nullptr ? std::max(1, std::atoi(e)) : GSH_OC_CELLS_PER_WG_DEFAULT;
    }();
    return v;
}

template <class P, int S>
int launch_cells(const OcCellArgs& a_in, int n_blocks, hipStream_t s)
{
    OcCellArgs a = a_in;
    a.slots_per_xcd = a.prn_per * a.bin_per * S;
    if (!((S == 1) && a.want_second) && P::EX64)
        {
            // work-groups per XCD: an XCD's 32 compute units take one of these work-groups each.  Up to GSH_OC_WG_PER_XCD (28) of them are occupied for the whole launch --
            // four are left to whatever else is in flight: with two batches in flight the other batch's forward transforms (41 work-groups) run there instead of waiting
            // for a whole batch of cells (32 per XCD: 138 us alone but 112 us pipelined; 28: 139 / 99) -- and the cells are dealt to them in passes
            static const int wg_per_xcd = [] { const char* e = std::getenv("GSH_OC_WG_PER_XCD"); return e != nullptr ? std::max(1, std::atoi(e)) : 28; }();
            a.cells_per_wg = std::min(onchip_cells_per_wg(), std::max(1, a.slots_per_xcd));
            int per_xcd = (a.slots_per_xcd + a.cells_per_wg - 1) / a.cells_per_wg;
            per_xcd = std::max(per_xcd, std::min(wg_per_xcd, a.slots_per_xcd));
            // never more of these work-groups than the XCD has compute units: they would run in two rounds, the second partly filled (50 000 points, S = 2: 328 slots
            // were 55 work-groups x 6 sub-cells = 32 + 23; as 32 x 11: 0.363 -> 0.334 ms per batch alone, 0.293 -> 0.286 pipelined -- profiles/ab/r06/session56.txt)
            static const int wg_per_xcd_max = [] { const char* e = std::getenv("GSH_OC_WG_PER_XCD_MAX"); return e != nullptr ? std::max(1, std::atoi(e)) : 32; }();
            per_xcd = std::min(per_xcd, std::max(wg_per_xcd_max, std::min(wg_per_xcd, a.slots_per_xcd)));
            a.cells_per_wg = (a.slots_per_xcd + per_xcd - 1) / per_xcd;
            n_blocks = 8 * per_xcd;
        }
    const bool grid = a.accumulate || a.store_grid;
    const bool off = a.offset != 0;  // the upper-half (bit_transition_flag) searches run the GRID flavour whether or not a grid is kept
    const bool in_kernel_statistic = (S == 1) && a.want_second;  // the second peak of a row needs the row's peak inside the cell: those flavours keep the hand-off
    if (!in_kernel_statistic) GSH_REQUIRE(a.waverows != nullptr, "on-chip cells without the per-wave record buffer");
    if constexpr (S == 1)
        {
            if (off && a.want_second)
                hipLaunchKernelGGL((oc_cell_kernel<P, 1, true, true, true>), dim3(n_blocks), dim3(P::THREADS), 0, s, a);
            else if (off)
                hipLaunchKernelGGL((oc_cell_kernel<P, 1, true, false, true>), dim3(n_blocks), dim3(P::THREADS), 0, s, a);
            else if (grid && a.want_second)
                hipLaunchKernelGGL((oc_cell_kernel<P, 1, true, true, false>), dim3(n_blocks), dim3(P::THREADS), 0, s, a);
            else if (grid)
                hipLaunchKernelGGL((oc_cell_kernel<P, 1, true, false, false>), dim3(n_blocks), dim3(P::THREADS), 0, s, a);
            else if (a.want_second)
                hipLaunchKernelGGL((oc_cell_kernel<P, 1, false, true, false>), dim3(n_blocks), dim3(P::THREADS), 0, s, a);
            else
                hipLaunchKernelGGL((oc_cell_kernel<P, 1, false, false, false>), dim3(n_blocks), dim3(P::THREADS), 0, s, a);
            if (!in_kernel_statistic)
                {
                    GSH_HIP(hipGetLastError());
                    hipLaunchKernelGGL(oc_rows_kernel, dim3(a.n_prn), dim3(64), 0, s, a, P::THREADS / 64, 1);
                }
        }
    else
        {
            GSH_REQUIRE(!a.want_second || (a.store_grid && a.grid != nullptr), "the peak-ratio statistic on a split plan scans the stored winning row: it needs the grid");
            GSH_REQUIRE(a.subrows != nullptr, "split plan without sub-cell records");
            if (off)
                hipLaunchKernelGGL((oc_cell_kernel<P, S, true, false, true>), dim3(n_blocks), dim3(P::THREADS), 0, s, a);
            else if (grid)
                hipLaunchKernelGGL((oc_cell_kernel<P, S, true, false, false>), dim3(n_blocks), dim3(P::THREADS), 0, s, a);
            else
                hipLaunchKernelGGL((oc_cell_kernel<P, S, false, false, false>), dim3(n_blocks), dim3(P::THREADS), 0, s, a);
            GSH_HIP(hipGetLastError());
            hipLaunchKernelGGL(oc_rows_kernel, dim3(a.n_prn), dim3(64), 0, s, a, P::THREADS / 64, S);
            if (a.want_second)
                {
                    GSH_HIP(hipGetLastError());
                    OcSecondArgs sa;
                    sa.grid = a.grid;
                    sa.results = a.results;
                    sa.n_bins = a.n_bins;
                    sa.effective = a.effective;
                    sa.samples_per_chip = a.samples_per_chip;
                    hipLaunchKernelGGL(oc_second_peak_kernel, dim3(a.n_prn), dim3(1024), 0, s, sa);
                }
        }
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}

template <class P, int S>
int launch_cells_dit(const OcCellArgs& a_in, int n_blocks, hipStream_t s)
{
    OcCellArgs a = a_in;
    {
        // class by class where Z carries the non-temporal hint (S = 4, 5, 8: 0.670 -> 0.653 / 0.822 -> 0.721 / 1.61 -> 1.30 ms per batch; WITHOUT the hint, with free-running
        // lanes, cell by cell is the better order at S = 8: 1.50 against 1.565) -- profiles/ab/r06/session49.txt, session53.txt.  GSH_OC_DIT_R_MAJOR = 0 / 1 overrides.
        static const int r_major = [] { const char* e = std::getenv("GSH_OC_DIT_R_MAJOR"); return e != nullptr ? std::atoi(e) : GSH_OC_DIT_R_MAJOR; }();
        a.dit_r_major = r_major < 0 ? (onchip_dit_nontemporal(S) ? 1 : 0) : r_major;
    }
    const bool grid = a.accumulate || a.store_grid;
    const bool off = a.offset != 0;
    GSH_REQUIRE(!a.want_second || (a.store_grid && a.grid != nullptr), "the peak-ratio statistic on a split plan scans the stored winning row: it needs the grid");
    GSH_REQUIRE(a.z != nullptr, "decimation-in-time split without its scratch");
    {
        // the sub-cells: persistent work-groups walking their XCD's slots in passes where the plan has its compute unit to itself (launch_cells has the reasoning);
        // GSH_OC_DIT_PERSIST=0: one freshly dispatched work-group per sub-cell, as until round 5 (A/B)
        static const int persist = [] { const char* e = std::getenv("GSH_OC_DIT_PERSIST"); return e != nullptr ? std::atoi(e) : 0; }();
        if (P::EX64 && persist)
            {
                static const int wg_per_xcd = [] { const char* e = std::getenv("GSH_OC_WG_PER_XCD"); return e != nullptr ? std::max(1, std::atoi(e)) : 28; }();
                OcCellArgs b = a;
                b.slots_per_xcd = a.prn_per * a.bin_per * S;
                b.cells_per_wg = std::min(onchip_cells_per_wg(), std::max(1, b.slots_per_xcd));
                int per_xcd = (b.slots_per_xcd + b.cells_per_wg - 1) / b.cells_per_wg;
                per_xcd = std::max(per_xcd, std::min(wg_per_xcd, b.slots_per_xcd));
                b.cells_per_wg = (b.slots_per_xcd + per_xcd - 1) / per_xcd;
                hipLaunchKernelGGL((oc_cell_kernel<P, S, false, false, false, true>), dim3(8 * per_xcd), dim3(P::THREADS), 0, s, b);
            }
        else
            hipLaunchKernelGGL((oc_subcell_dit_kernel<P, S>), dim3(n_blocks), dim3(P::THREADS), 0, s, a);
    }
    GSH_HIP(hipGetLastError());
    // parts per cell (1, 2 or 4; never more than the S records a row has room for) and threads per work-group of the combine launch
    static const int parts_env = [] { const char* e = std::getenv("GSH_OC_COMBINE_PARTS"); return e != nullptr ? std::atoi(e) : GSH_OC_COMBINE_PARTS; }();
    static const int threads_env = [] { const char* e = std::getenv("GSH_OC_COMBINE_THREADS"); return e != nullptr ? std::atoi(e) : GSH_OC_COMBINE_WG; }();
    const int parts = (parts_env >= 4 && S >= 4) ? 4 : (parts_env >= 2 && S >= 2) ? 2 : 1;
    const dim3 threads(static_cast<unsigned>(std::min(OC_COMBINE_THREADS, std::max(64, threads_env / 64 * 64))));
    GSH_REQUIRE(parts == 1 || a.subrows != nullptr, "a combine launch over parts of cells without the parts' records");
    GSH_REQUIRE(GSH_OC_COMBINE_HANDOFF || a.waverows != nullptr, "a combine launch without the per-wave record buffer");
    const dim3 cells(static_cast<unsigned>(a.n_prn * a.n_bins * parts));
    auto combine = [&](auto Hc) {
        constexpr int H = decltype(Hc)::value;
        if (off)
            hipLaunchKernelGGL((oc_combine_dit_kernel<P::N, S, true, true, H>), cells, threads, 0, s, a);
        else if (grid)
            hipLaunchKernelGGL((oc_combine_dit_kernel<P::N, S, true, false, H>), cells, threads, 0, s, a);
        else
            hipLaunchKernelGGL((oc_combine_dit_kernel<P::N, S, false, false, H>), cells, threads, 0, s, a);
    };
    if constexpr (S >= 4)
        {
            if (parts == 4)
                combine(std::integral_constant<int, 4>{});
            else if (parts == 2)
                combine(std::integral_constant<int, 2>{});
            else
                combine(std::integral_constant<int, 1>{});
        }
    else if constexpr (S >= 2)
        {
            if (parts == 2)
                combine(std::integral_constant<int, 2>{});
            else
                combine(std::integral_constant<int, 1>{});
        }
    else
        combine(std::integral_constant<int, 1>{});
#if !GSH_OC_COMBINE_HANDOFF
    GSH_HIP(hipGetLastError());
    hipLaunchKernelGGL(oc_rows_kernel, dim3(a.n_prn), dim3(64), 0, s, a, static_cast<int>(threads.x) / 64, parts);
#endif
    if (a.want_second)
        {
            GSH_HIP(hipGetLastError());
            OcSecondArgs sa;
            sa.grid = a.grid;
            sa.results = a.results;
            sa.n_bins = a.n_bins;
            sa.effective = a.effective;
            sa.samples_per_chip = a.samples_per_chip;
            hipLaunchKernelGGL(oc_second_peak_kernel, dim3(a.n_prn), dim3(1024), 0, s, sa);
        }
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}
}  // namespace

// Z of a decimation-in-time split with the non-temporal hint, then its batches' cells one after the other (gsh_acq_time_dwells_pipelined) and its sub-cells walked class
// by class (launch_cells_dit): the three belong together, by measurement (profiles/ab/r06/session44.txt, 45, 49, 53) -- S = 4, 5 (100 000 / 128 000 points): 0.72 -> 0.65 /
// 0.877 -> 0.725 ms per batch; S = 8 (200 000 points, 2.1 GB of Z per batch): the hint and the order alone are SLOWER than plain accesses and free-running lanes (1.615
// against 1.495 ms), all three together 1.30.  A compile-time choice (GSH_OC_Z_NT_BELOW_S).
bool onchip_dit_nontemporal(int split) { return GSH_OC_Z_NT_STORE && GSH_OC_Z_NT_LOAD && split >= 2 && split < GSH_OC_Z_NT_BELOW_S; }

// split plans with at least this many sub-cells run decimation in time (GSH_OC_DIT_MIN_S in the environment overrides: 0 = never)
int onchip_dit_min_s()
{
    static const int v = [] {
        const char* e = std::getenv("GSH_OC_DIT_MIN_S");
        return e != nullptr ? std::atoi(e) : GSH_OC_DIT_MIN_S;
    }();
    return v;
}

bool onchip_dit(int n)
{
    const int sp = onchip_split(n);
    const int min_s = onchip_dit_min_s();
    return sp > 0 && min_s > 0 && sp >= min_s;
}

bool onchip_supported(int n)
{
#define GSH_OC_CASE(r1, r2, r3) \
    if (n == (r1) * (r2) * (r3)) return true;
    GSH_OC_PLANS(GSH_OC_CASE)
#undef GSH_OC_CASE
    return false;
}

int onchip_split(int n)
{
#define GSH_OC_CASE(sp, r1, r2, r3) \
    if (n == (sp) * (r1) * (r2) * (r3)) return sp;
    GSH_OC_SPLIT_PLANS(GSH_OC_CASE)
#undef GSH_OC_CASE
    return 0;
}

int onchip_forward(int n, const float2* src, size_t src_stride, int n_in, int place_off, const float* wipe_hz, double fs, float2* dst,
    int batch, hipStream_t s, int fold)
{
    if (batch <= 0) return GSH_OK;
    OcFwdArgs a;
    a.src = reinterpret_cast<const cf*>(src);
    a.src_stride = src_stride;
    a.n_in = n_in;
    a.place_off = place_off;
    a.wipe_hz = wipe_hz;
    a.inv_fs = 1.0 / fs;
    a.dst = reinterpret_cast<cf*>(dst);
    a.residue_major = onchip_dit(n) ? 1 : 0;
    a.fold = fold;
#define GSH_OC_CASE(r1, r2, r3) \
    if (n == (r1) * (r2) * (r3)) return launch_forward<oc::Plan<r1, r2, r3>>(a, batch, s);
    GSH_OC_PLANS(GSH_OC_CASE)
#undef GSH_OC_CASE
#define GSH_OC_CASE(sp, r1, r2, r3) \
    if (n == (sp) * (r1) * (r2) * (r3)) return launch_forward_split<oc::Plan<r1, r2, r3>, sp>(a, batch, s);
    GSH_OC_SPLIT_PLANS(GSH_OC_CASE)
#undef GSH_OC_CASE
    return set_error(GSH_ERR_UNSUPPORTED, "no on-chip plan for fft_size %d", n);
}

int onchip_correlate(int n, const float2* spectra, const float2* codes, float* grid, RowStat* rows, RowStat* subrows, DevAcqResult* results,
    unsigned* arrivals, int n_prn, int n_bins, int offset, int effective, int accumulate, int store_grid, int samples_per_chip, int use_cfar,
    unsigned dwell_count, float weight, hipStream_t s, float2* z, RowStat* waverows)
{
    if (n_prn <= 0 || n_bins <= 0) return GSH_OK;
    GSH_REQUIRE((offset == 0 && effective == n) || (2 * offset == n && effective == offset), "lags [%d, %d + %d) of a %d-point transform: neither all of it nor its upper half", offset, offset, effective, n);
    OcCellArgs a;
    a.spectra = reinterpret_cast<const cf*>(spectra);
    a.codes = reinterpret_cast<const cf*>(codes);
    a.grid = grid;
    a.rows = rows;
    a.subrows = subrows;
    a.waverows = waverows;
    a.z = reinterpret_cast<cf*>(z);
    a.offset = offset;
    a.results = results;
    a.arrivals = arrivals;
    a.n_prn = n_prn;
    a.n_bins = n_bins;
    // XCDs across the PRNs (the rest across the bins).  GSH_OC_XP in the environment caps it (A/B: profiles/ab/r06/session61.txt)
    static const int xp_cap = [] { const char* e = std::getenv("GSH_OC_XP"); return e != nullptr ? std::max(1, std::min(8, std::atoi(e))) : 8; }();
    int xp = 1;
    while (xp < xp_cap && 2 * xp <= n_prn) xp *= 2;
    const int xb = 8 / xp;
    a.xp = xp;
    a.prn_per = (n_prn + xp - 1) / xp;
    a.bin_per = (n_bins + xb - 1) / xb;
    a.effective = effective;
    a.accumulate = accumulate;
    a.store_grid = store_grid;
    a.samples_per_chip = samples_per_chip;
    a.want_second = use_cfar ? 0 : 1;
    a.use_cfar = use_cfar;
    a.dwell_count = dwell_count ? dwell_count : 1u;
    a.weight = weight;
    const int n_blocks = 8 * a.prn_per * a.bin_per;
    a.cells_per_wg = 1;
    a.slots_per_xcd = a.prn_per * a.bin_per;  // (x S in launch_cells)
    a.dit_r_major = 0;  // (launch_cells_dit decides)
    {
        // GSH_OC_STAGGER="groups,ticks" (A/B): see OcCellArgs::stagger_groups
        static const int groups = [] { const char* e = std::getenv("GSH_OC_STAGGER"); return e != nullptr ? std::atoi(e) : GSH_OC_STAGGER_GROUPS_DEFAULT; }();
        static const int ticks = [] {
            const char* e = std::getenv("GSH_OC_STAGGER");
            const char* c = e != nullptr ? std::strchr(e, ',') : nullptr;
            return c != nullptr ? std::atoi(c + 1) : GSH_OC_STAGGER_TICKS_DEFAULT;
        }();
        static const int prefetch = [] { const char* e = std::getenv("GSH_OC_PREFETCH"); return e != nullptr ? std::atoi(e) : 2; }();  // 0 none, 1 the next bin's lines, 2 + the idle waves' own operands
        a.prefetch_next = prefetch;
        a.stagger_groups = groups;
        a.stagger_ticks = ticks;
        a.stagger_first = 256;  // one work-group per compute unit in the first round
    }
#define GSH_OC_CASE(r1, r2, r3) \
    if (n == (r1) * (r2) * (r3)) return launch_cells<oc::Plan<r1, r2, r3>, 1>(a, n_blocks, s);
    GSH_OC_PLANS(GSH_OC_CASE)
#undef GSH_OC_CASE
    const bool dit = onchip_dit(n);
#define GSH_OC_CASE(sp, r1, r2, r3)                                                                              \
    if (n == (sp) * (r1) * (r2) * (r3))                                                                          \
        return dit ? launch_cells_dit<oc::Plan<r1, r2, r3>, sp>(a, n_blocks * (sp), s) : launch_cells<oc::Plan<r1, r2, r3>, sp>(a, n_blocks * (sp), s);
    GSH_OC_SPLIT_PLANS(GSH_OC_CASE)
#undef GSH_OC_CASE
    return set_error(GSH_ERR_UNSUPPORTED, "no on-chip plan for fft_size %d", n);
}

}  // namespace gsh

#ifdef GSH_OC_PROFILE
extern "C" int gsh_debug_oc_profile(unsigned long long* out, size_t n_words)
{
    const size_t all = sizeof(gsh::g_oc_prof) / sizeof(unsigned long long);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gsh::g_oc_prof), sizeof(unsigned long long) * (n_words < all ? n_words : all), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
