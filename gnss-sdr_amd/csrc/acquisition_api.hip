// C-ABI acquisition entry points (include/gnss_sdr_hip.h, gsh_acq_*): host bookkeeping around the
// kernels of pcps_fft.hip.  Mirrors the data flow of pcps_acquisition::acquisition_core
// (gnss-sdr, src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.cc:648-684) for a whole batch of
// local codes: the D wiped-off forward FFTs are computed ONCE and shared by every PRN (the reference recomputes
// them in every channel's block).
#include "pcps_fft.h"
#include "sample_convert.h"
#include "sample_stream.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <new>
#include <vector>

struct gsh_acq
{
    int device{0};
    gsh_acq_conf conf{};
    hipStream_t stream{nullptr};
    gsh::FftPlan plan;
    int n_bins{0};
    std::vector<float> h_bins_hz;   // the float the reference hands to update_local_carrier (acq.cc:289, :299)
    float* d_bins_hz{nullptr};
    int n_bins2{0};                 // num_doppler_bins_step2 (0: no fine-Doppler step)
    std::vector<float> h_bins2_hz;  // max_prn * n_bins2, rebuilt by every gsh_acq_dwell_step2 call
    float* d_bins2_hz{nullptr};
    int16_t* d_in16{nullptr};       // cshort input staging (gsh_acq_dwell_cshort), allocated on first use
    float2* d_in{nullptr};        // consumed_samples
    float2* d_spectra{nullptr};   // n_bins * n
    float2* d_codes{nullptr};     // max_prn * n   (forward FFT of the placed code, permuted layout, unconjugated)
    float2* d_codes_sel{nullptr}; // gsh_acq_dwell_slots: the spectra of the slots of one batch side by side (allocated on first use)
    float2* d_tmp{nullptr};       // chunk_prn * n_bins * n
    float* d_grid{nullptr};       // max_prn * n_bins * effective
    gsh::RowStat* d_rows{nullptr};
    gsh_acq_pair_peak* d_pair{nullptr};  // gsh_acq_noncoherent_pair_peaks: one record per bin
    gsh::RowStat* d_subrows{nullptr};  // split plans (N = S * M): the sub-cells' records, S per (PRN, bin)
    gsh::RowStat* d_waverows{nullptr}; // on-chip path: the cells' per-wave partial records, ONCHIP_MAX_WAVES per (PRN, bin, sub-cell) (pcps_fft.h)
    int split{0};
    float2* d_z{nullptr};              // decimation-in-time split plans (gsh::onchip_dit): the sub-cells' length-M transforms, max_prn * n_bins * n
    gsh::DevAcqResult* d_results{nullptr};
    unsigned* d_arrivals{nullptr};          // on-chip path: per-PRN arrival counters, zero between launches
    gsh::DevAcqResult* h_results{nullptr};  // pinned
    float2* h_stage{nullptr};               // pinned, max(consumed, code length)
    std::vector<char> code_set;
    int chunk_prn{1};
    bool onchip{false};  // whole-transform-on-chip path (pcps_onchip.hip); spectra then sit in natural order
    bool have_input{false};
    float grid_weight{1.0f};  // gsh_acq_set_grid_weight
    // lengths without any plan (a prime factor above 61 ...): the circular correlation of length N is computed as a linear one inside a
    // power-of-two transform of M >= 3 N points (signal at [0, N), the code twice at [0, 2 N), lags read from [M - N, M)); conf.fft_size is then M,
    // conf.effective_fft_size stays N and every magnitude is scaled by (N / M)^2 so that values match an N-point unnormalised inverse
    bool padded{false};
    uint32_t logical_n{0};
    float pad_scale{1.0f};
    double* d_power{nullptr};  // gsh_acq_input_power scratch
    float2* d_tc_code{nullptr};  // gsh_acq_time_correlate: code, delays and results
    size_t tc_code_len{0};
    uint32_t* d_tc_delays{nullptr};
    float2* d_tc_out{nullptr};
    hipEvent_t ev0{nullptr}, ev1{nullptr};
    // further issue lanes for gsh_acq_time_dwells_pipelined (on-chip path): a stream and per-batch buffers each, so that batch k+1's forward transforms
    // and first cells fill the compute units batch k's last cells leave idle (lane 0 is the handle's own stream and buffers)
    struct Lane
    {
        hipStream_t stream{nullptr};
        float2* d_spectra{nullptr};
        gsh::RowStat* d_rows{nullptr};
        gsh::RowStat* d_subrows{nullptr};
        gsh::RowStat* d_waverows{nullptr};
        float2* d_z{nullptr};
        gsh::DevAcqResult* d_results{nullptr};
        unsigned* d_arrivals{nullptr};
        hipEvent_t ev{nullptr};
        hipEvent_t ev_cells{nullptr};  // recorded behind the lane's last cell launches (decimation-in-time plans: the next batch's cells wait for it, gsh_acq_time_dwells_pipelined)
    };
    static constexpr int MAX_LANES = 4;
    Lane lane[MAX_LANES];
    int n_lanes{1};
};

namespace
{
using gsh::set_error;

// sum of |x|^2 over one input block (pcps_tong_acquisition_cc.cc:208-209, galileo_pcps_8ms_acquisition_cc.cc:190-191).
// Each term is the float the reference's volk_32fc_magnitude_squared_32f forms; the terms are added in double (the
// reference's volk_32f_accumulator_s32f adds them in float in an ISA-dependent lane order, so its last bits are not defined).
__global__ __launch_bounds__(1024) void input_power_kernel(const float2* __restrict__ x, int n, double* __restrict__ out)
{
    __shared__ double part[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024)
        {
            const float2 v = x[i];
            s += static_cast<double>(__fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y)));
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        {
            double t = 0.0;
            for (int w = 0; w < 16; w++) t += part[w];
            *out = t;
        }
}

// A second stream whose kernels really run next to those of `ref`.  The runtime spreads the streams of one priority over a few hardware queues and
// offers no way to ask which; two streams on one queue run their kernels one after the other.  Whether the second lane of the pipelined dwell loop
// overlapped with the first therefore used to depend on how many streams the process had created before (bench.py: 0.142 ms per 25 000-point batch, the same
// loop in a fresh process 0.113 ms; profiles/ab/r03/acq_hw_queues.txt).  So: try a few streams, time one 40-us spin on each of `ref` and the candidate
// started together, keep the first candidate for which the pair takes the time of one; fall back to a stream of another priority class (queues are per
// class: always concurrent, but the class takes precedence at dispatch, 0.121 ms).
__global__ void spin_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    for (int i = 0; i < 100000 && wall_clock64() - t0 < ticks; i++) __builtin_amdgcn_s_sleep(8);  // bounded whatever the clock does
}

int make_concurrent_stream(hipStream_t ref, const std::vector<hipStream_t>& others, hipStream_t* out)
{
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    GSH_HIP(hipEventCreate(&e0));
    GSH_HIP(hipEventCreate(&e1));
    GSH_HIP(hipEventCreate(&e2));
    int rate_khz = 100000;
    int dev = 0;
    GSH_HIP(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
    const long long ticks = static_cast<long long>(rate_khz) * 40 / 1000;  // 40 us
    std::vector<hipStream_t> tried;
    hipStream_t found = nullptr;
    for (int attempt = 0; attempt < 8 && found == nullptr; attempt++)
        {
            hipStream_t cand = nullptr;
            GSH_HIP(hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++)  // the first launch on a new stream also pays for its queue
                {
                    GSH_HIP(hipEventRecord(e0, ref));
                    GSH_HIP(hipStreamWaitEvent(cand, e0, 0));
                    for (hipStream_t o : others) GSH_HIP(hipStreamWaitEvent(o, e0, 0));
                    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, ref, ticks);
                    for (hipStream_t o : others) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, o, ticks);
                    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cand, ticks);
                    GSH_HIP(hipEventRecord(e2, cand));
                    GSH_HIP(hipStreamWaitEvent(ref, e2, 0));
                    for (hipStream_t o : others)  // (the lanes already found run side by side with `ref`: the candidate must with all of them)
                        {
                            GSH_HIP(hipEventRecord(e2, o));
                            GSH_HIP(hipStreamWaitEvent(ref, e2, 0));
                        }
                    GSH_HIP(hipEventRecord(e1, ref));
                    GSH_HIP(hipEventSynchronize(e1));
                    float ms = 0.0f;
                    GSH_HIP(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                }
            if (std::getenv("GSH_TRACE_STREAMS") != nullptr) std::fprintf(stderr, "gsh: stream probe %d: %zu 40-us spins in %.1f us\n", attempt, others.size() + 2, best * 1e3f);
            if (best < 0.080f)  // measured: 60-64 us side by side (one spin + the event round trip), 100 us one after the other
                found = cand;
            else
                tried.push_back(cand);
        }
    for (hipStream_t t : tried) (void)hipStreamDestroy(t);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(e2);
    if (found == nullptr)
        {
            int least = 0, greatest = 0;
            GSH_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            GSH_HIP(hipStreamCreateWithPriority(&found, hipStreamNonBlocking, greatest));
        }
    *out = found;
    return GSH_OK;
}

void fill_bins(gsh_acq* a)
{
    // acq.cc:284-291: doppler = -doppler_max + doppler_center + doppler_step * index (+ FDMA bias, :289)
    for (int d = 0; d < a->n_bins; d++)
        a->h_bins_hz[d] = static_cast<float>(a->conf.doppler_bias + (-a->conf.doppler_max + a->conf.doppler_center + a->conf.doppler_step * d));
}

int upload_bins(gsh_acq* a)
{
    GSH_HIP(hipMemcpyAsync(a->d_bins_hz, a->h_bins_hz.data(), sizeof(float) * a->n_bins, hipMemcpyHostToDevice, a->stream));
    GSH_HIP(hipStreamSynchronize(a->stream));
    return GSH_OK;
}

// queue one full dwell batch on the handle's stream (input already in d_in)
int enqueue_dwell(gsh_acq* a, uint32_t n_prn, int accumulate, uint32_t dwell_count)
{
    const gsh_acq_conf& c = a->conf;
    const int n = static_cast<int>(c.fft_size);
    const int eff = static_cast<int>(c.effective_fft_size);
    if (a->onchip)
        {
            // acq.cc:657-664 (zero padding) + :531-535 (wipe-off, forward FFT): one work-group per bin
            int rc = gsh::onchip_forward(n, a->d_in, 0, static_cast<int>(c.consumed_samples), 0, a->d_bins_hz, static_cast<double>(c.fs_in),
                a->d_spectra, a->n_bins, a->stream, static_cast<int>(std::max(1u, c.fold)));
            if (rc != GSH_OK) return rc;
            // acq.cc:538-553 + the per-row part of :409-519: one work-group per (PRN, bin) cell, nothing leaves the CU
            return gsh::onchip_correlate(n, a->d_spectra, a->d_codes, a->d_grid, a->d_rows, a->d_subrows, a->d_results, a->d_arrivals, static_cast<int>(n_prn),
                a->n_bins, c.bit_transition_flag ? eff : 0, eff, accumulate, (c.no_grid && !(a->split > 0 && !c.use_cfar)) ? 0 : 1, static_cast<int>(c.samples_per_chip), c.use_cfar, dwell_count ? dwell_count : 1u,
                a->grid_weight, a->stream, a->d_z, a->d_waverows);
        }
    // acq.cc:657-664 (zero padding) + :531-535 (wipe-off, forward FFT) for every bin
    int rc = gsh::fft_forward(a->plan, a->d_in, 0, static_cast<int>(c.consumed_samples), 0, a->d_bins_hz, static_cast<double>(c.fs_in),
        a->d_tmp, a->d_spectra, a->n_bins, a->stream, static_cast<int>(std::max(1u, c.fold)));
    if (rc != GSH_OK) return rc;
    const int grid_off = a->padded ? n - eff : (c.bit_transition_flag ? eff : 0);  // acq.cc:544; padded: the lags sit in the last N outputs
    for (uint32_t p0 = 0; p0 < n_prn; p0 += static_cast<uint32_t>(a->chunk_prn))
        {
            const int np = static_cast<int>(std::min<uint32_t>(a->chunk_prn, n_prn - p0));
            rc = gsh::correlate_grid(a->plan, a->d_spectra, a->d_codes + static_cast<size_t>(p0) * n, a->d_tmp,
                a->d_grid + static_cast<size_t>(p0) * a->n_bins * eff, np, a->n_bins, grid_off, eff, accumulate, a->grid_weight * a->pad_scale, a->stream);
            if (rc != GSH_OK) return rc;
        }
    return gsh::grid_statistics(a->d_grid, a->d_rows, a->d_results, static_cast<int>(n_prn), a->n_bins, eff,
        static_cast<int>(c.samples_per_chip), c.use_cfar, dwell_count ? dwell_count : 1u, a->stream);
}

// time-domain correlation of the resident block with an unfolded code at a few candidate delays, at one Doppler bin
// (pcps_quicksync_acquisition_cc.cc:300-323): out[c] = sum_j x[delay_c + j] w[delay_c + j] code[j].  One work-group per candidate;
// products in float as the reference forms them, the sum in double (the reference adds sequentially in float).
constexpr int TC_MAX_DELAYS = 100;  // complex_acumulator is std::array<gr_complex, 100> (quicksync.cc:295)
__global__ __launch_bounds__(1024) void time_correlate_kernel(const float2* __restrict__ x, const float2* __restrict__ code, int code_len,
    const uint32_t* __restrict__ delays, float f_hz, double inv_fs, float2* __restrict__ out)
{
    __shared__ double pr[16], pi[16];
    const int delay = static_cast<int>(delays[blockIdx.x]);
    double sr = 0.0, si = 0.0;
    for (int j = threadIdx.x; j < code_len; j += 1024)
        {
            const int n = delay + j;
            double rev = static_cast<double>(f_hz) * static_cast<double>(n) * inv_fs;
            rev -= rint(rev);
            float sn, cs;
            sincospif(static_cast<float>(2.0 * rev), &sn, &cs);
            const float2 v = x[n], c = code[j];
            // in_temp = in * wipe-off (quicksync.cc:251), then * code (:311): two float complex products
            const float wr = __fsub_rn(__fmul_rn(v.x, cs), __fmul_rn(v.y, -sn));
            const float wi = __fadd_rn(__fmul_rn(v.x, -sn), __fmul_rn(v.y, cs));
            sr += static_cast<double>(__fsub_rn(__fmul_rn(wr, c.x), __fmul_rn(wi, c.y)));
            si += static_cast<double>(__fadd_rn(__fmul_rn(wr, c.y), __fmul_rn(wi, c.x)));
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        {
            sr += __shfl_down(sr, off, 64);
            si += __shfl_down(si, off, 64);
        }
    if ((threadIdx.x & 63) == 0)
        {
            pr[threadIdx.x >> 6] = sr;
            pi[threadIdx.x >> 6] = si;
        }
    __syncthreads();
    if (threadIdx.x == 0)
        {
            double tr = 0.0, ti = 0.0;
            for (int w = 0; w < 16; w++)
                {
                    tr += pr[w];
                    ti += pi[w];
                }
            out[blockIdx.x] = make_float2(static_cast<float>(tr), static_cast<float>(ti));
        }
}

// Non-coherent combination of the magnitude rows of two code slots before the arg-max, per Doppler bin
// (galileo_e5a_noncoherent_iq_acquisition_caf_cc.cc:357-492).  The block correlates up to four local codes per bin -- data (I) and pilot (Q)
// component, each as (1,1,1) "A" and with the first code period inverted "B" --, keeps for each component the combination whose row maximum
// is larger (:403, :410, magt = max / N^4 compared with >=), ADDS the two kept magnitude rows cell by cell (:419-431) and takes the
// arg-max of the sum (:487-492).  One work-group per bin: thread 0 repeats the block's choice from the row records of the dwell that just
// ran, all threads add and scan the two rows.  As written in the reference, the Q-B candidate is ranked by the I-B row read at Q-B's
// arg-max (:393 `magt_QB = d_magnitudeIB[indext_QB] / ...`); that is reproduced.
struct PairPeakArgs
{
    const float* grid;           // [slot][bin][n]
    const gsh::RowStat* rows;    // [slot][bin]
    int n, n_bins;
    int ia, qa, ib, qb;          // slots; qa / ib / qb may be -1
    float divisor;               // (N^2)^2 as the block forms it in float (:327, :369)
    gsh_acq_pair_peak* out;      // [bin]
};

__global__ __launch_bounds__(1024) void pair_peaks_kernel(PairPeakArgs a)
{
    __shared__ int s_sel[2];
    __shared__ float s_v[16];
    __shared__ unsigned s_i[16];
    const int d = blockIdx.x, t = threadIdx.x;
    const size_t row = static_cast<size_t>(a.n);
    auto rec = [&](int slot) -> gsh::RowStat { return a.rows[static_cast<size_t>(slot) * a.n_bins + d]; };
    auto grow = [&](int slot) -> const float* { return a.grid + (static_cast<size_t>(slot) * a.n_bins + d) * row; };
    if (t == 0)
        {
            int sel_i = a.ia, sel_q = a.qa;
            const float m_ia = __fdiv_rn(rec(a.ia).maxv, a.divisor);
            if (a.ib >= 0)
                {
                    const float m_ib = __fdiv_rn(rec(a.ib).maxv, a.divisor);
                    if (!(m_ia >= m_ib)) sel_i = a.ib;
                    if (a.qa >= 0 && a.qb >= 0)
                        {
                            const float m_qa = __fdiv_rn(rec(a.qa).maxv, a.divisor);
                            const float m_qb = __fdiv_rn(grow(a.ib)[rec(a.qb).idx], a.divisor);  // :393, as written
                            if (!(m_qa >= m_qb)) sel_q = a.qb;
                        }
                }
            s_sel[0] = sel_i;
            s_sel[1] = sel_q;
        }
    __syncthreads();
    const int sel_i = s_sel[0], sel_q = s_sel[1];
    float best = -1.0f;
    unsigned at = 0xFFFFFFFFu;
    if (sel_q >= 0)
        {
            const float* __restrict__ gi = grow(sel_i);
            const float* __restrict__ gq = grow(sel_q);
            for (int i = t; i < a.n; i += 1024)
                {
                    const float m = __fadd_rn(gi[i], gq[i]);   // d_magnitudeI[i] += d_magnitudeQ[i]
                    if (m > best)                              // i ascending per thread: the lowest index among equals stays
                        {
                            best = m;
                            at = static_cast<unsigned>(i);
                        }
                }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                {
                    const float ov = __shfl_down(best, off, 64);
                    const unsigned oi = __shfl_down(at, off, 64);
                    if (ov > best || (ov == best && oi < at))
                        {
                            best = ov;
                            at = oi;
                        }
                }
            if ((t & 63) == 0)
                {
                    s_v[t >> 6] = best;
                    s_i[t >> 6] = at;
                }
            __syncthreads();
        }
    if (t == 0)
        {
            gsh_acq_pair_peak o;
            if (sel_q >= 0)
                {
                    for (int w = 1; w < 16; w++)
                        if (s_v[w] > best || (s_v[w] == best && s_i[w] < at))
                            {
                                best = s_v[w];
                                at = s_i[w];
                            }
                    o.peak = best;
                    o.index_time = at;
                    o.caf_q = rec(sel_q).maxv;   // d_magnitudeQ[A|B][indext_Q[A|B]] (:416, :427)
                }
            else
                {
                    const gsh::RowStat r = rec(sel_i);
                    o.peak = r.maxv;
                    o.index_time = r.idx;
                    o.caf_q = 0.0f;
                }
            o.caf_i = rec(sel_i).maxv;           // :407, :441
            o.i_slot = static_cast<uint32_t>(sel_i);
            o.q_slot = sel_q >= 0 ? static_cast<uint32_t>(sel_q) : 0xFFFFFFFFu;
            a.out[d] = o;
        }
}

int check_dwell_args(gsh_acq* a, uint32_t n_prn, const void* results)
{
    GSH_REQUIRE(a != nullptr && results != nullptr, "null argument");
    GSH_REQUIRE(n_prn >= 1 && n_prn <= a->conf.max_prn, "n_prn %u outside 1..%u", n_prn, a->conf.max_prn);
    for (uint32_t p = 0; p < n_prn; p++)
        if (!a->code_set[p]) return set_error(GSH_ERR_STATE, "local code of prn slot %u has not been set (set_local_code)", p);
    return GSH_OK;
}

int check_accumulate(gsh_acq* a, int accumulate)
{
    if (accumulate && a->conf.no_grid)
        return set_error(GSH_ERR_STATE, "accumulate != 0 needs the stored grid, but the handle was created with no_grid = 1");
    return GSH_OK;
}

int finish_results(gsh_acq* a, uint32_t n_prn, gsh_acq_result* results)
{
    GSH_HIP(hipMemcpyAsync(a->h_results, a->d_results, sizeof(gsh::DevAcqResult) * n_prn, hipMemcpyDeviceToHost, a->stream));
    GSH_HIP(hipStreamSynchronize(a->stream));
    for (uint32_t p = 0; p < n_prn; p++)
        {
            const gsh::DevAcqResult& r = a->h_results[p];
            gsh_acq_result& o = results[p];
            o.index_time = r.index_time;
            o.index_doppler = r.index_doppler;
            o.doppler_hz = -a->conf.doppler_max + a->conf.doppler_center + a->conf.doppler_step * static_cast<int32_t>(r.index_doppler);  // acq.cc:431/:478
            o.acq_delay_samples = std::fmod(static_cast<float>(r.index_time), a->conf.samples_per_code);                                    // acq.cc:582
            o.peak = r.peak;
            o.input_power = r.input_power;
            o.second_peak = r.second_peak;
            o.test_statistics = r.test_statistics;
        }
    return GSH_OK;
}

// ---- regularized incomplete gamma P(a, x), Q(a, x) = 1 - P and the inverse of P (for compute_threshold) ----
// Both tails are produced without cancellation: the series gives P (used where P is not close to 1), the
// continued fraction gives Q directly (the thresholds of interest sit at Q ~ 1e-9 ... 1e-13).
void gamma_pq(double a, double x, double* p_out, double* q_out)
{
    if (x <= 0.0)
        {
            *p_out = 0.0;
            *q_out = 1.0;
            return;
        }
    const double lg = std::lgamma(a);
    const double front = std::exp(-x + a * std::log(x) - lg);
    if (x < a + 1.0)
        {
            // P = e^{-x} x^a / Gamma(a) * sum_{k>=0} x^k / (a (a+1) ... (a+k))
            double term = 1.0 / a, sum = term;
            for (int k = 1; k < 100000; k++)
                {
                    term *= x / (a + k);
                    sum += term;
                    if (std::fabs(term) < std::fabs(sum) * 1e-17) break;
                }
            *p_out = sum * front;
            *q_out = 1.0 - *p_out;
            return;
        }
    // Q by the continued fraction (modified Lentz)
    const double tiny = 1e-300;
    double b = x + 1.0 - a;
    double cc = 1.0 / tiny;
    double d = 1.0 / b;
    double h = d;
    for (int i = 1; i < 100000; i++)
        {
            const double an = -static_cast<double>(i) * (static_cast<double>(i) - a);
            b += 2.0;
            d = an * d + b;
            if (std::fabs(d) < tiny) d = tiny;
            cc = b + an / cc;
            if (std::fabs(cc) < tiny) cc = tiny;
            d = 1.0 / d;
            const double del = d * cc;
            h *= del;
            if (std::fabs(del - 1.0) < 1e-16) break;
        }
    *q_out = front * h;
    *p_out = 1.0 - *q_out;
}

double gamma_p_inv_impl(double a, double p)
{
    if (p <= 0.0) return 0.0;
    if (p >= 1.0) return INFINITY;
    const double q = 1.0 - p;          // exact for p in [0.5, 1)
    const bool upper = p > 0.5;        // solve on the tail that is not close to 1
    const double lg = std::lgamma(a);
    auto resid = [&](double x) {
        double pp, qq;
        gamma_pq(a, x, &pp, &qq);
        return upper ? (q - qq) : (pp - p);  // increasing in x either way
    };
    double lo = 0.0, hi = std::max(1.0, a);
    while (resid(hi) < 0.0) hi *= 2.0;
    double x = 0.5 * (lo + hi);
    for (int it = 0; it < 300; it++)
        {
            const double f = resid(x);
            if (f > 0.0)
                hi = x;
            else
                lo = x;
            const double dfdx = std::exp((a - 1.0) * std::log(x) - x - lg);  // dP/dx = -dQ/dx
            double xn = (dfdx > 0.0) ? x - f / dfdx : 0.5 * (lo + hi);
            if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
            if (std::fabs(xn - x) <= 4e-16 * std::fabs(x)) return xn;
            x = xn;
        }
    return x;
}
}  // namespace

namespace gsh
{
double gamma_p_inv(double a, double p) { return gamma_p_inv_impl(a, p); }
}  // namespace gsh

extern "C"
{
    int gsh_acq_create(int device, const gsh_acq_conf* conf, gsh_acq_t** out)
    {
        GSH_REQUIRE(out != nullptr && conf != nullptr, "null argument");
        *out = nullptr;
        gsh_acq_conf c = *conf;
        GSH_REQUIRE(c.fs_in > 0, "fs_in must be positive");
        GSH_REQUIRE(c.fft_size >= 4 && c.fft_size <= (1u << 24), "fft_size %u outside 4..2^24", c.fft_size);
        if (c.fold > 1)
            {
                // pcps_quicksync_acquisition_cc.cc:243-263: the block is `fold` transform lengths long and is folded after the wipe-off
                GSH_REQUIRE(c.fold <= 10000, "fold %u outside 1..10000", c.fold);
                GSH_REQUIRE(static_cast<uint64_t>(c.fold) * c.fft_size == c.consumed_samples, "with fold = %u, consumed_samples must be fold * fft_size", c.fold);
                GSH_REQUIRE(!c.bit_transition_flag && c.num_doppler_bins_step2 == 0, "fold > 1 excludes bit_transition_flag and the two-step search");
            }
        else
            {
                GSH_REQUIRE(c.consumed_samples >= 1 && c.consumed_samples <= c.fft_size, "consumed_samples %u outside 1..fft_size", c.consumed_samples);
                GSH_REQUIRE(c.consumed_samples == c.fft_size || 2 * c.consumed_samples == c.fft_size,
                    "fft_size must be consumed_samples or twice it (acq.cc:111)");
            }
        if (c.bit_transition_flag)
            GSH_REQUIRE(c.effective_fft_size * 2 == c.fft_size, "bit_transition_flag needs effective_fft_size = fft_size/2 (acq.cc:112)");
        else
            GSH_REQUIRE(c.effective_fft_size == c.fft_size, "effective_fft_size must equal fft_size without bit_transition_flag (acq.cc:112)");
        GSH_REQUIRE(c.doppler_step > 0, "doppler_step must be positive");
        if (c.num_doppler_bins == 0)
            c.num_doppler_bins = static_cast<uint32_t>(std::ceil(static_cast<double>(2 * c.doppler_max) / static_cast<double>(c.doppler_step)));  // acq.cc:113
        GSH_REQUIRE(c.num_doppler_bins >= 1 && c.num_doppler_bins <= 4096, "num_doppler_bins %u outside 1..4096", c.num_doppler_bins);
        GSH_REQUIRE(c.max_prn >= 1 && c.max_prn <= 4096, "max_prn %u outside 1..4096", c.max_prn);
        GSH_REQUIRE(2 * c.samples_per_chip < c.effective_fft_size, "samples_per_chip %u too large for %u cells (acq.cc:485-509 would not terminate)", c.samples_per_chip, c.effective_fft_size);
        GSH_REQUIRE(c.transform_path == 0 || c.transform_path == 1, "transform_path %d outside 0..1", c.transform_path);
        GSH_REQUIRE(c.num_doppler_bins_step2 <= c.num_doppler_bins, "num_doppler_bins_step2 %u exceeds num_doppler_bins %u (the narrow grid reuses the wide grid's buffers, as d_magnitude_grid does, acq.cc:137)",
            c.num_doppler_bins_step2, c.num_doppler_bins);
        GSH_REQUIRE(c.num_doppler_bins_step2 == 0 || (std::isfinite(c.doppler_step2) && c.doppler_step2 > 0.0f), "doppler_step2 must be positive");
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh_acq* a = new (std::nothrow) gsh_acq();
        GSH_REQUIRE(a != nullptr, "out of host memory");
        a->device = device;
        a->conf = c;
        a->n_bins = static_cast<int>(c.num_doppler_bins);
        a->h_bins_hz.assign(a->n_bins, 0.0f);
        a->n_bins2 = static_cast<int>(c.num_doppler_bins_step2);
        a->h_bins2_hz.assign(static_cast<size_t>(a->n_bins2) * c.max_prn, 0.0f);
        a->code_set.assign(c.max_prn, 0);
        // one compute unit per transform when the length has a plan; S work-groups per transform for N = S * M (no folding there);
        // bit_transition_flag is an epilogue predicate of the same kernels.  The peak-ratio statistic on a split plan (round 3): the S sub-cells
        // of a row each own every S-th lag, so none of them can blank around the row's peak; the rows are kept in the magnitude grid instead
        // (stored whatever no_grid says) and a small kernel behind the cell launch scans every PRN's winning row for the second peak.
        a->onchip = (c.transform_path == 0) && gsh::onchip_supported(static_cast<int>(c.fft_size));
        if (!a->onchip && c.transform_path == 0 && c.fold <= 1 && gsh::onchip_split(static_cast<int>(c.fft_size)) > 0)
            {
                a->onchip = true;
                a->split = gsh::onchip_split(static_cast<int>(c.fft_size));
            }
        if (!a->onchip)
            {
                rc = gsh::plan_create(static_cast<int>(c.fft_size), &a->plan);
                if (rc != GSH_OK)
                    {
                        // no radix schedule for this length: fall back to the zero-padded power-of-two form (plain searches only)
                        const bool plain = !c.bit_transition_flag && c.consumed_samples == c.fft_size && c.fold <= 1 && c.num_doppler_bins_step2 == 0;
                        uint64_t m = 4;
                        while (m < 3ull * c.fft_size) m <<= 1;
                        if (!plain || m > (1ull << 21) || gsh::plan_create(static_cast<int>(m), &a->plan) != GSH_OK)
                            {
                                delete a;
                                return set_error(GSH_ERR_UNSUPPORTED,
                                    "fft_size %u has no radix schedule (prime factor above 61) and the zero-padded fallback covers plain searches up to 699050 points only",
                                    c.fft_size);
                            }
                        a->padded = true;
                        a->logical_n = c.fft_size;
                        const double ratio = static_cast<double>(c.fft_size) / static_cast<double>(m);
                        a->pad_scale = static_cast<float>(ratio * ratio);
                        c.fft_size = static_cast<uint32_t>(m);  // effective_fft_size and consumed_samples stay N
                        a->conf = c;
                    }
            }
        const size_t n = c.fft_size, eff = c.effective_fft_size, D = a->n_bins, P = c.max_prn;
        // scratch for the inverse's intermediate: bound it to 2 GiB and to the 65535-cell launch limit
        const size_t per_prn = D * n * sizeof(float2);
        size_t chunk = std::max<size_t>(1, (size_t(2) << 30) / per_prn);
        chunk = std::min(chunk, P);
        chunk = std::min<size_t>(chunk, std::max<size_t>(1, 65535 / D));
        a->chunk_prn = static_cast<int>(chunk);
        auto fail = [&](hipError_t e, const char* what) {
            gsh::hip_fail(e, what, __FILE__, __LINE__);
            gsh_acq_destroy(a);
            return GSH_ERR_HIP;
        };
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
        if ((e = hipMalloc(&a->d_bins_hz, sizeof(float) * D)) != hipSuccess) return fail(e, "hipMalloc(bins)");
        if (a->n_bins2 > 0 && (e = hipMalloc(&a->d_bins2_hz, sizeof(float) * a->n_bins2 * P)) != hipSuccess) return fail(e, "hipMalloc(bins2)");
        const size_t in_len = std::max<size_t>(n, c.consumed_samples);  // fold > 1: the block is longer than the transform
        if ((e = hipMalloc(&a->d_in, sizeof(float2) * in_len)) != hipSuccess) return fail(e, "hipMalloc(in)");
        if ((e = hipMalloc(&a->d_spectra, sizeof(float2) * D * n)) != hipSuccess) return fail(e, "hipMalloc(spectra)");
        if ((e = hipMalloc(&a->d_codes, sizeof(float2) * P * n)) != hipSuccess) return fail(e, "hipMalloc(codes)");
        // the four-step path keeps its inter-pass intermediate in HBM; the on-chip path only needs one row to stage a code
        const size_t tmp_rows = a->onchip ? size_t(1) : std::max(chunk * D, size_t(2));
        if ((e = hipMalloc(&a->d_tmp, sizeof(float2) * tmp_rows * n)) != hipSuccess) return fail(e, "hipMalloc(tmp)");
        const bool need_grid = !(a->onchip && c.no_grid) || (a->split > 0 && !c.use_cfar);
        if (need_grid)
            {
                if ((e = hipMalloc(&a->d_grid, sizeof(float) * P * D * eff)) != hipSuccess) return fail(e, "hipMalloc(grid)");
                if ((e = hipMemset(a->d_grid, 0, sizeof(float) * P * D * eff)) != hipSuccess) return fail(e, "hipMemset(grid)");
            }
        if ((e = hipMalloc(&a->d_rows, sizeof(gsh::RowStat) * P * D)) != hipSuccess) return fail(e, "hipMalloc(rows)");
        if (a->split > 0 && (e = hipMalloc(&a->d_subrows, sizeof(gsh::RowStat) * P * std::max<size_t>(D, a->n_bins2) * a->split)) != hipSuccess)
            return fail(e, "hipMalloc(subrows)");
        if (a->onchip && (e = hipMalloc(&a->d_waverows, sizeof(gsh::RowStat) * P * std::max<size_t>(D, a->n_bins2) * static_cast<size_t>(std::max(a->split, 1)) * gsh::ONCHIP_MAX_WAVES)) != hipSuccess)
            return fail(e, "hipMalloc(wave records)");
        if (a->split > 0 && gsh::onchip_dit(static_cast<int>(n)))
            {
                const size_t cells = P * std::max<size_t>(D, a->n_bins2);
                if ((e = hipMalloc(&a->d_z, sizeof(float2) * cells * n)) != hipSuccess) return fail(e, "hipMalloc(sub-cell transforms)");
            }
        if ((e = hipMalloc(&a->d_results, sizeof(gsh::DevAcqResult) * P)) != hipSuccess) return fail(e, "hipMalloc(results)");
        if ((e = hipMalloc(&a->d_arrivals, sizeof(unsigned) * P)) != hipSuccess) return fail(e, "hipMalloc(arrivals)");
        if ((e = hipMemset(a->d_arrivals, 0, sizeof(unsigned) * P)) != hipSuccess) return fail(e, "hipMemset(arrivals)");
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&a->h_results), sizeof(gsh::DevAcqResult) * P, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc");
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&a->h_stage), sizeof(float2) * in_len, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc");
        if ((e = hipEventCreate(&a->ev0)) != hipSuccess) return fail(e, "hipEventCreate");
        if ((e = hipEventCreate(&a->ev1)) != hipSuccess) return fail(e, "hipEventCreate");
        fill_bins(a);
        rc = upload_bins(a);
        if (rc != GSH_OK)
            {
                gsh_acq_destroy(a);
                return rc;
            }
        *out = a;
        return GSH_OK;
    }

    void gsh_acq_destroy(gsh_acq_t* a)
    {
        if (!a) return;
        (void)hipSetDevice(a->device);
        if (a->stream) (void)hipStreamSynchronize(a->stream);
        gsh::plan_destroy(&a->plan);
        if (a->d_bins_hz) (void)hipFree(a->d_bins_hz);
        if (a->d_bins2_hz) (void)hipFree(a->d_bins2_hz);
        if (a->d_in16) (void)hipFree(a->d_in16);
        if (a->d_power) (void)hipFree(a->d_power);
        if (a->d_tc_code) (void)hipFree(a->d_tc_code);
        if (a->d_tc_delays) (void)hipFree(a->d_tc_delays);
        if (a->d_tc_out) (void)hipFree(a->d_tc_out);
        if (a->d_in) (void)hipFree(a->d_in);
        if (a->d_spectra) (void)hipFree(a->d_spectra);
        if (a->d_codes) (void)hipFree(a->d_codes);
        if (a->d_codes_sel) (void)hipFree(a->d_codes_sel);
        if (a->d_tmp) (void)hipFree(a->d_tmp);
        if (a->d_grid) (void)hipFree(a->d_grid);
        if (a->d_rows) (void)hipFree(a->d_rows);
        if (a->d_subrows) (void)hipFree(a->d_subrows);
        if (a->d_waverows) (void)hipFree(a->d_waverows);
        if (a->d_z) (void)hipFree(a->d_z);
        if (a->d_pair) (void)hipFree(a->d_pair);
        if (a->d_results) (void)hipFree(a->d_results);
        if (a->d_arrivals) (void)hipFree(a->d_arrivals);
        if (a->h_results) (void)hipHostFree(a->h_results);
        if (a->h_stage) (void)hipHostFree(a->h_stage);
        if (a->ev0) (void)hipEventDestroy(a->ev0);
        if (a->ev1) (void)hipEventDestroy(a->ev1);
        for (int l = 0; l < gsh_acq::MAX_LANES; l++)
            if (a->lane[l].ev_cells) (void)hipEventDestroy(a->lane[l].ev_cells);
        for (int l = 1; l < gsh_acq::MAX_LANES; l++)
            {
                gsh_acq::Lane& ln = a->lane[l];
                if (ln.stream) (void)hipStreamSynchronize(ln.stream);
                if (ln.d_spectra) (void)hipFree(ln.d_spectra);
                if (ln.d_rows) (void)hipFree(ln.d_rows);
                if (ln.d_subrows) (void)hipFree(ln.d_subrows);
                if (ln.d_waverows) (void)hipFree(ln.d_waverows);
                if (ln.d_z) (void)hipFree(ln.d_z);
                if (ln.d_results) (void)hipFree(ln.d_results);
                if (ln.d_arrivals) (void)hipFree(ln.d_arrivals);
                if (ln.ev) (void)hipEventDestroy(ln.ev);
                if (ln.stream) (void)hipStreamDestroy(ln.stream);
            }
        if (a->stream) (void)hipStreamDestroy(a->stream);
        delete a;
    }

    int gsh_acq_set_local_code(gsh_acq_t* a, uint32_t prn_slot, const float* code_iq)
    {
        GSH_REQUIRE(a != nullptr && code_iq != nullptr, "null argument");
        GSH_REQUIRE(prn_slot < a->conf.max_prn, "prn_slot %u outside 0..%u", prn_slot, a->conf.max_prn - 1);
        GSH_HIP(hipSetDevice(a->device));
        const gsh_acq_conf& c = a->conf;
        // placement rules of acq.cc:230-247
        int n_in, place_off;
        if (a->padded)
            {
                // the code twice, back to back: lag -tau of the linear correlation with [c c] is lag tau of the circular one with c
                const size_t N = a->logical_n;
                std::memcpy(a->h_stage, code_iq, sizeof(float2) * N);
                std::memcpy(a->h_stage + N, code_iq, sizeof(float2) * N);
                float2* d_code_time = a->d_tmp + static_cast<size_t>(c.fft_size);
                if (a->n_bins < 2) d_code_time = a->d_spectra;
                GSH_HIP(hipMemcpyAsync(d_code_time, a->h_stage, sizeof(float2) * 2 * N, hipMemcpyHostToDevice, a->stream));
                int rcp = gsh::fft_forward(a->plan, d_code_time, 0, static_cast<int>(2 * N), 0, nullptr, 1.0, a->d_tmp, a->d_codes + static_cast<size_t>(prn_slot) * c.fft_size, 1,
                    a->stream);
                if (rcp != GSH_OK) return rcp;
                GSH_HIP(hipStreamSynchronize(a->stream));
                a->code_set[prn_slot] = 1;
                return GSH_OK;
            }
        if (c.bit_transition_flag)
            {
                n_in = static_cast<int>(c.fft_size / 2);
                place_off = static_cast<int>(c.fft_size / 2);
            }
        else if (c.consumed_samples == c.fft_size || c.fold > 1)
            {
                n_in = static_cast<int>(c.fft_size);  // fold > 1: the caller hands over the code already folded to fft_size (quicksync.cc:137-152)
                place_off = 0;
            }
        else
            {
                n_in = static_cast<int>(c.consumed_samples);
                place_off = static_cast<int>(c.fft_size - c.consumed_samples);
            }
        std::memcpy(a->h_stage, code_iq, sizeof(float2) * static_cast<size_t>(n_in));
        if (a->onchip)
            {
                GSH_HIP(hipMemcpyAsync(a->d_tmp, a->h_stage, sizeof(float2) * static_cast<size_t>(n_in), hipMemcpyHostToDevice, a->stream));
                int rc1 = gsh::onchip_forward(static_cast<int>(c.fft_size), a->d_tmp, 0, n_in, place_off, nullptr, 1.0,
                    a->d_codes + static_cast<size_t>(prn_slot) * c.fft_size, 1, a->stream);
                if (rc1 != GSH_OK) return rc1;
                GSH_HIP(hipStreamSynchronize(a->stream));
                a->code_set[prn_slot] = 1;
                return GSH_OK;
            }
        // stage the time-domain replica in d_tmp's tail-free area: d_in is reserved for the signal, so use d_spectra[0] row
        // as scratch input is unsafe while a dwell is queued; everything here is on one stream, so order is preserved.
        float2* d_code_time = a->d_tmp + static_cast<size_t>(c.fft_size);  // second row of tmp (tmp holds >= n_bins rows)
        float2* d_scratch = a->d_tmp;                                      // first row: column-pass output
        if (a->n_bins < 2)
            {
                // tmp has a single row: borrow the grid-independent spectra buffer instead
                d_code_time = a->d_spectra;
            }
        GSH_HIP(hipMemcpyAsync(d_code_time, a->h_stage, sizeof(float2) * static_cast<size_t>(n_in), hipMemcpyHostToDevice, a->stream));
        int rc = gsh::fft_forward(a->plan, d_code_time, 0, n_in, place_off, nullptr, 1.0, d_scratch,
            a->d_codes + static_cast<size_t>(prn_slot) * c.fft_size, 1, a->stream);
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipStreamSynchronize(a->stream));
        a->code_set[prn_slot] = 1;
        return GSH_OK;
    }

    int gsh_acq_set_doppler_center(gsh_acq_t* a, int32_t doppler_center)
    {
        GSH_REQUIRE(a != nullptr, "null handle");
        GSH_HIP(hipSetDevice(a->device));
        if (doppler_center != a->conf.doppler_center)  // acq.cc:741-745
            {
                a->conf.doppler_center = doppler_center;
                fill_bins(a);
                return upload_bins(a);
            }
        return GSH_OK;
    }

    int gsh_acq_set_doppler_bias(gsh_acq_t* a, int32_t doppler_bias)
    {
        GSH_REQUIRE(a != nullptr, "null handle");
        GSH_HIP(hipSetDevice(a->device));
        if (doppler_bias != a->conf.doppler_bias)  // acq.cc:252-272: is_fdma() re-derives d_doppler_bias for the PRN at hand
            {
                a->conf.doppler_bias = doppler_bias;
                fill_bins(a);
                return upload_bins(a);
            }
        return GSH_OK;
    }

    int gsh_acq_set_grid_weight(gsh_acq_t* a, float weight)
    {
        GSH_REQUIRE(a != nullptr, "null handle");
        GSH_REQUIRE(std::isfinite(weight), "weight must be finite");
        if (weight != 1.0f && a->conf.no_grid)
            return set_error(GSH_ERR_STATE, "a grid weight applies to the stored grid, but the handle was created with no_grid = 1");
        a->grid_weight = weight;
        return GSH_OK;
    }

    int gsh_acq_input_power(gsh_acq_t* a, float* mean_power)
    {
        GSH_REQUIRE(a != nullptr && mean_power != nullptr, "null argument");
        if (!a->have_input) return set_error(GSH_ERR_STATE, "no input block has been handed to this handle yet");
        GSH_HIP(hipSetDevice(a->device));
        if (a->d_power == nullptr) GSH_HIP(hipMalloc(&a->d_power, sizeof(double)));
        hipLaunchKernelGGL(input_power_kernel, dim3(1), dim3(1024), 0, a->stream, a->d_in, static_cast<int>(a->conf.consumed_samples), a->d_power);
        GSH_HIP(hipGetLastError());
        double sum = 0.0;
        GSH_HIP(hipMemcpyAsync(&sum, a->d_power, sizeof(double), hipMemcpyDeviceToHost, a->stream));
        GSH_HIP(hipStreamSynchronize(a->stream));
        // pcps_tong_acquisition_cc.cc:208-210: float sum of float |x|^2, then / (float) fft_size
        *mean_power = static_cast<float>(sum) / static_cast<float>(a->conf.consumed_samples);
        return GSH_OK;
    }

    int gsh_acq_dwell_device(gsh_acq_t* a, const void* device_in_iq, uint32_t n_prn, int accumulate, uint32_t dwell_count, gsh_acq_result* results)
    {
        int rc = check_dwell_args(a, n_prn, results);
        if (rc == GSH_OK) rc = check_accumulate(a, accumulate);
        if (rc != GSH_OK) return rc;
        GSH_REQUIRE(device_in_iq != nullptr, "null input");
        GSH_HIP(hipSetDevice(a->device));
        GSH_HIP(hipMemcpyAsync(a->d_in, device_in_iq, sizeof(float2) * a->conf.consumed_samples, hipMemcpyDeviceToDevice, a->stream));
        a->have_input = true;
        rc = enqueue_dwell(a, n_prn, accumulate, dwell_count);
        if (rc != GSH_OK) return rc;
        return finish_results(a, n_prn, results);
    }

    int gsh_acq_stage_input(gsh_acq_t* a, const float* in_iq)
    {
        GSH_REQUIRE(a != nullptr && in_iq != nullptr, "null argument");
        GSH_HIP(hipSetDevice(a->device));
        std::memcpy(a->h_stage, in_iq, sizeof(float2) * a->conf.consumed_samples);
        GSH_HIP(hipMemcpyAsync(a->d_in, a->h_stage, sizeof(float2) * a->conf.consumed_samples, hipMemcpyHostToDevice, a->stream));
        GSH_HIP(hipStreamSynchronize(a->stream));  // h_stage may be rewritten by the next call
        a->have_input = true;
        return GSH_OK;
    }

    int gsh_acq_stage_input_device(gsh_acq_t* a, const void* device_in_iq)
    {
        GSH_REQUIRE(a != nullptr && device_in_iq != nullptr, "null argument");
        GSH_HIP(hipSetDevice(a->device));
        GSH_HIP(hipMemcpyAsync(a->d_in, device_in_iq, sizeof(float2) * a->conf.consumed_samples, hipMemcpyDeviceToDevice, a->stream));
        a->have_input = true;
        return GSH_OK;
    }

    int gsh_acq_dwell_resident(gsh_acq_t* a, uint32_t n_prn, int accumulate, uint32_t dwell_count, gsh_acq_result* results)
    {
        int rc = check_dwell_args(a, n_prn, results);
        if (rc == GSH_OK) rc = check_accumulate(a, accumulate);
        if (rc != GSH_OK) return rc;
        if (!a->have_input) return set_error(GSH_ERR_STATE, "no input block resident: call gsh_acq_stage_input[_device] first");
        GSH_HIP(hipSetDevice(a->device));
        rc = enqueue_dwell(a, n_prn, accumulate, dwell_count);
        if (rc != GSH_OK) return rc;
        return finish_results(a, n_prn, results);
    }

    int gsh_acq_dwell_ring(gsh_acq_t* a, gsh_stream_t* ring, uint64_t first_sample, uint32_t n_prn, int accumulate, uint32_t dwell_count,
        gsh_acq_result* results)
    {
        GSH_REQUIRE(a != nullptr && ring != nullptr, "null argument");
        GSH_REQUIRE(ring->device == a->device, "the ring lives on device %d, the acquisition on device %d", ring->device, a->device);
        const float2* w = nullptr;
        int rc = gsh::stream_window(ring, first_sample, a->conf.consumed_samples, &w);
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipSetDevice(a->device));
        if (ring->pushed != nullptr) GSH_HIP(hipStreamWaitEvent(a->stream, ring->pushed, 0));  // conversions queued by gsh_stream_push_device
        rc = gsh_acq_dwell_device(a, w, n_prn, accumulate, dwell_count, results);
        if (rc != GSH_OK) return rc;
        return gsh::stream_mark_read(ring, first_sample, a->stream);  // (the dwell has already been waited for when results were asked for; harmless otherwise)
    }

    int gsh_acq_dwell(gsh_acq_t* a, const float* in_iq, uint32_t n_prn, int accumulate, uint32_t dwell_count, gsh_acq_result* results)
    {
        int rc = check_dwell_args(a, n_prn, results);
        if (rc == GSH_OK) rc = check_accumulate(a, accumulate);
        if (rc != GSH_OK) return rc;
        GSH_REQUIRE(in_iq != nullptr, "null input");
        GSH_HIP(hipSetDevice(a->device));
        std::memcpy(a->h_stage, in_iq, sizeof(float2) * a->conf.consumed_samples);
        GSH_HIP(hipMemcpyAsync(a->d_in, a->h_stage, sizeof(float2) * a->conf.consumed_samples, hipMemcpyHostToDevice, a->stream));
        a->have_input = true;
        rc = enqueue_dwell(a, n_prn, accumulate, dwell_count);
        if (rc != GSH_OK) return rc;
        return finish_results(a, n_prn, results);
    }

    int gsh_acq_dwell_slots(gsh_acq_t* a, const float* in_iq, uint32_t n, const uint32_t* prn_slots, gsh_acq_result* results)
    {
        GSH_REQUIRE(a != nullptr && in_iq != nullptr && prn_slots != nullptr && results != nullptr, "null argument");
        GSH_REQUIRE(n >= 1 && n <= a->conf.max_prn, "n %u outside 1..%u", n, a->conf.max_prn);
        bool prefix = true;
        for (uint32_t i = 0; i < n; i++)
            {
                GSH_REQUIRE(prn_slots[i] < a->conf.max_prn, "prn slot %u outside 0..%u", prn_slots[i], a->conf.max_prn - 1);
                if (!a->code_set[prn_slots[i]]) return set_error(GSH_ERR_STATE, "local code of prn slot %u has not been set (set_local_code)", prn_slots[i]);
                prefix = prefix && prn_slots[i] == i;
            }
        GSH_HIP(hipSetDevice(a->device));
        const size_t len = a->conf.fft_size;
        float2* const all_codes = a->d_codes;
        if (!prefix)
            {
                // the batch's code spectra side by side (a few hundred KB each, device to device): the kernels then see an ordinary batch of n codes
                if (a->d_codes_sel == nullptr) GSH_HIP(hipMalloc(&a->d_codes_sel, sizeof(float2) * len * a->conf.max_prn));
                for (uint32_t i = 0; i < n; i++)
                    GSH_HIP(hipMemcpyAsync(a->d_codes_sel + static_cast<size_t>(i) * len, all_codes + static_cast<size_t>(prn_slots[i]) * len, sizeof(float2) * len,
                        hipMemcpyDeviceToDevice, a->stream));
            }
        std::memcpy(a->h_stage, in_iq, sizeof(float2) * a->conf.consumed_samples);
        GSH_HIP(hipMemcpyAsync(a->d_in, a->h_stage, sizeof(float2) * a->conf.consumed_samples, hipMemcpyHostToDevice, a->stream));
        a->have_input = true;
        if (!prefix) a->d_codes = a->d_codes_sel;
        int rc = enqueue_dwell(a, n, 0, 1u);
        a->d_codes = all_codes;
        if (rc != GSH_OK) return rc;
        return finish_results(a, n, results);
    }

    int gsh_acq_dwell_cshort(gsh_acq_t* a, const int16_t* in_iq16, uint32_t n_prn, int accumulate, uint32_t dwell_count, gsh_acq_result* results)
    {
        int rc = check_dwell_args(a, n_prn, results);
        if (rc == GSH_OK) rc = check_accumulate(a, accumulate);
        if (rc != GSH_OK) return rc;
        GSH_REQUIRE(in_iq16 != nullptr, "null input");
        GSH_HIP(hipSetDevice(a->device));
        const size_t n = a->conf.consumed_samples;
        if (a->d_in16 == nullptr) GSH_HIP(hipMalloc(&a->d_in16, sizeof(int16_t) * 2 * n));
        // h_stage holds fft_size complex64 = room for 4 * fft_size int16
        std::memcpy(a->h_stage, in_iq16, sizeof(int16_t) * 2 * n);
        GSH_HIP(hipMemcpyAsync(a->d_in16, a->h_stage, sizeof(int16_t) * 2 * n, hipMemcpyHostToDevice, a->stream));
        rc = gsh::convert_to_complex(a->d_in16, GSH_ITEM_SHORT, 0, a->d_in, n, a->stream);  // acq.cc:653-656
        if (rc != GSH_OK) return rc;
        a->have_input = true;
        rc = enqueue_dwell(a, n_prn, accumulate, dwell_count);
        if (rc != GSH_OK) return rc;
        return finish_results(a, n_prn, results);
    }

    static int dwell_step2_common(gsh_acq_t* a, uint32_t n, const uint32_t* prn_slots, const float* centers, const float* powers, int accumulate,
        uint32_t dwell_count, gsh_acq_result* results)
    {
        const gsh_acq_conf& c = a->conf;
        const int nfft = static_cast<int>(c.fft_size);
        const int eff = static_cast<int>(c.effective_fft_size);
        const int D2 = a->n_bins2;
        const float half = static_cast<float>(std::floor(static_cast<double>(D2) / 2.0));  // acq.cc:298
        for (uint32_t i = 0; i < n; i++)
            for (int d = 0; d < D2; d++)
                {
                    const float doppler = (static_cast<float>(d) - half) * c.doppler_step2;    // acq.cc:298
                    a->h_bins2_hz[static_cast<size_t>(i) * D2 + d] = centers[i] + doppler;     // acq.cc:299
                }
        GSH_HIP(hipMemcpyAsync(a->d_bins2_hz, a->h_bins2_hz.data(), sizeof(float) * D2 * n, hipMemcpyHostToDevice, a->stream));
        const size_t grid_per_prn = static_cast<size_t>(a->n_bins) * eff;
        for (uint32_t i = 0; i < n; i++)
            {
                const uint32_t slot = prn_slots[i];
                const float* bins = a->d_bins2_hz + static_cast<size_t>(i) * D2;
                float* grid = a->d_grid ? a->d_grid + slot * grid_per_prn : nullptr;  // rows 0..D2-1 of the PRN's grid (acq.cc:526 reuses d_magnitude_grid)
                int rc;
                if (a->onchip)
                    {
                        rc = gsh::onchip_forward(nfft, a->d_in, 0, static_cast<int>(c.consumed_samples), 0, bins, static_cast<double>(c.fs_in), a->d_spectra, D2,
                            a->stream);
                        if (rc != GSH_OK) return rc;
                        rc = gsh::onchip_correlate(nfft, a->d_spectra, a->d_codes + static_cast<size_t>(slot) * nfft, grid, a->d_rows + static_cast<size_t>(i) * D2,
                            a->d_subrows ? a->d_subrows + static_cast<size_t>(i) * D2 * a->split : nullptr,
                            a->d_results + i, a->d_arrivals + i, 1, D2, c.bit_transition_flag ? eff : 0, eff, accumulate, (c.no_grid && !(a->split > 0 && !c.use_cfar)) ? 0 : 1, static_cast<int>(c.samples_per_chip), c.use_cfar,
                            dwell_count ? dwell_count : 1u, a->grid_weight, a->stream, a->d_z ? a->d_z + static_cast<size_t>(i) * D2 * nfft : nullptr,
                            a->d_waverows + static_cast<size_t>(i) * D2 * static_cast<size_t>(std::max(a->split, 1)) * gsh::ONCHIP_MAX_WAVES);
                    }
                else
                    {
                        rc = gsh::fft_forward(a->plan, a->d_in, 0, static_cast<int>(c.consumed_samples), 0, bins, static_cast<double>(c.fs_in), a->d_tmp,
                            a->d_spectra, D2, a->stream);
                        if (rc != GSH_OK) return rc;
                        rc = gsh::correlate_grid(a->plan, a->d_spectra, a->d_codes + static_cast<size_t>(slot) * nfft, a->d_tmp, grid, 1, D2,
                            c.bit_transition_flag ? eff : 0, eff, accumulate, a->grid_weight, a->stream);
                        if (rc != GSH_OK) return rc;
                        rc = gsh::grid_statistics(grid, a->d_rows + static_cast<size_t>(i) * D2, a->d_results + i, 1, D2, eff, static_cast<int>(c.samples_per_chip),
                            c.use_cfar, dwell_count ? dwell_count : 1u, a->stream);
                    }
                if (rc != GSH_OK) return rc;
            }
        GSH_HIP(hipMemcpyAsync(a->h_results, a->d_results, sizeof(gsh::DevAcqResult) * n, hipMemcpyDeviceToHost, a->stream));
        GSH_HIP(hipStreamSynchronize(a->stream));
        for (uint32_t i = 0; i < n; i++)
            {
                const gsh::DevAcqResult& r = a->h_results[i];
                gsh_acq_result& o = results[i];
                o.index_time = r.index_time;
                o.index_doppler = r.index_doppler;
                o.doppler_hz = static_cast<int32_t>(centers[i] + (static_cast<float>(r.index_doppler) - half) * c.doppler_step2);  // acq.cc:436 / :481
                o.acq_delay_samples = std::fmod(static_cast<float>(r.index_time), c.samples_per_code);
                o.peak = r.peak;
                o.second_peak = r.second_peak;
                if (c.use_cfar)
                    {
                        // acq.cc:428-445: d_input_power is NOT recomputed in step two; the statistic divides by step one's value
                        o.input_power = powers[i];
                        o.test_statistics = (powers[i] < std::numeric_limits<float>::epsilon()) ? 0.0f : r.peak / powers[i];
                    }
                else
                    {
                        o.input_power = r.input_power;
                        o.test_statistics = r.test_statistics;
                    }
            }
        return GSH_OK;
    }

    static int check_step2_args(gsh_acq_t* a, const void* in, uint32_t n, const uint32_t* prn_slots, const float* centers, const float* powers, int accumulate,
        gsh_acq_result* results)
    {
        GSH_REQUIRE(a != nullptr && in != nullptr && prn_slots != nullptr && centers != nullptr && results != nullptr, "null argument");
        if (a->n_bins2 <= 0) return set_error(GSH_ERR_STATE, "the handle was created with num_doppler_bins_step2 = 0 (make_two_steps off)");
        GSH_REQUIRE(n >= 1 && n <= a->conf.max_prn, "n %u outside 1..%u", n, a->conf.max_prn);
        GSH_REQUIRE(!a->conf.use_cfar || powers != nullptr, "the CFAR statistic of step two needs step one's input power (acq.cc:428-445)");
        for (uint32_t i = 0; i < n; i++)
            {
                GSH_REQUIRE(prn_slots[i] < a->conf.max_prn, "prn slot %u outside 0..%u", prn_slots[i], a->conf.max_prn - 1);
                if (!a->code_set[prn_slots[i]]) return set_error(GSH_ERR_STATE, "local code of prn slot %u has not been set (set_local_code)", prn_slots[i]);
                GSH_REQUIRE(std::isfinite(centers[i]), "doppler_center_step_two[%u] is not finite", i);
                for (uint32_t k = 0; k < i; k++) GSH_REQUIRE(prn_slots[k] != prn_slots[i], "prn slot %u listed twice", prn_slots[i]);
            }
        return check_accumulate(a, accumulate);
    }

    int gsh_acq_dwell_step2_device(gsh_acq_t* a, const void* device_in_iq, uint32_t n, const uint32_t* prn_slots, const float* doppler_center_step_two,
        const float* input_power_step_one, int accumulate, uint32_t dwell_count, gsh_acq_result* results)
    {
        int rc = check_step2_args(a, device_in_iq, n, prn_slots, doppler_center_step_two, input_power_step_one, accumulate, results);
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipSetDevice(a->device));
        GSH_HIP(hipMemcpyAsync(a->d_in, device_in_iq, sizeof(float2) * a->conf.consumed_samples, hipMemcpyDeviceToDevice, a->stream));
        a->have_input = true;
        return dwell_step2_common(a, n, prn_slots, doppler_center_step_two, input_power_step_one, accumulate, dwell_count, results);
    }

    int gsh_acq_dwell_step2(gsh_acq_t* a, const float* in_iq, uint32_t n, const uint32_t* prn_slots, const float* doppler_center_step_two,
        const float* input_power_step_one, int accumulate, uint32_t dwell_count, gsh_acq_result* results)
    {
        int rc = check_step2_args(a, in_iq, n, prn_slots, doppler_center_step_two, input_power_step_one, accumulate, results);
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipSetDevice(a->device));
        std::memcpy(a->h_stage, in_iq, sizeof(float2) * a->conf.consumed_samples);
        GSH_HIP(hipMemcpyAsync(a->d_in, a->h_stage, sizeof(float2) * a->conf.consumed_samples, hipMemcpyHostToDevice, a->stream));
        a->have_input = true;
        return dwell_step2_common(a, n, prn_slots, doppler_center_step_two, input_power_step_one, accumulate, dwell_count, results);
    }

    int gsh_acq_read_grid(gsh_acq_t* a, uint32_t prn_slot, float* grid)
    {
        GSH_REQUIRE(a != nullptr && grid != nullptr, "null argument");
        GSH_REQUIRE(prn_slot < a->conf.max_prn, "prn_slot %u outside 0..%u", prn_slot, a->conf.max_prn - 1);
        if (a->d_grid == nullptr) return set_error(GSH_ERR_STATE, "the handle was created with no_grid = 1: no grid is stored");
        GSH_HIP(hipSetDevice(a->device));
        const size_t row = static_cast<size_t>(a->n_bins) * a->conf.effective_fft_size;
        GSH_HIP(hipMemcpyAsync(grid, a->d_grid + prn_slot * row, sizeof(float) * row, hipMemcpyDeviceToHost, a->stream));
        GSH_HIP(hipStreamSynchronize(a->stream));
        return GSH_OK;
    }

    int gsh_acq_read_row_peaks(gsh_acq_t* a, uint32_t prn_slot, float* row_peak, uint32_t* row_index_time)
    {
        GSH_REQUIRE(a != nullptr && row_peak != nullptr && row_index_time != nullptr, "null argument");
        GSH_REQUIRE(prn_slot < a->conf.max_prn, "prn_slot %u outside 0..%u", prn_slot, a->conf.max_prn - 1);
        if (!a->have_input) return set_error(GSH_ERR_STATE, "no dwell has run on this handle yet");
        GSH_HIP(hipSetDevice(a->device));
        std::vector<gsh::RowStat> rows(static_cast<size_t>(a->n_bins));
        GSH_HIP(hipMemcpyAsync(rows.data(), a->d_rows + static_cast<size_t>(prn_slot) * a->n_bins, sizeof(gsh::RowStat) * a->n_bins, hipMemcpyDeviceToHost,
            a->stream));
        GSH_HIP(hipStreamSynchronize(a->stream));
        for (int d = 0; d < a->n_bins; d++)
            {
                row_peak[d] = rows[static_cast<size_t>(d)].maxv;
                row_index_time[d] = rows[static_cast<size_t>(d)].idx;
            }
        return GSH_OK;
    }

    int gsh_acq_noncoherent_pair_peaks(gsh_acq_t* a, int32_t slot_ia, int32_t slot_qa, int32_t slot_ib, int32_t slot_qb, gsh_acq_pair_peak* out)
    {
        GSH_REQUIRE(a != nullptr && out != nullptr, "null argument");
        const int32_t P = static_cast<int32_t>(a->conf.max_prn);
        GSH_REQUIRE(slot_ia >= 0 && slot_ia < P, "slot_ia %d outside 0..%d", slot_ia, P - 1);
        GSH_REQUIRE(slot_qa >= -1 && slot_qa < P && slot_ib >= -1 && slot_ib < P && slot_qb >= -1 && slot_qb < P, "slot outside -1..%d", P - 1);
        GSH_REQUIRE(slot_qb < 0 || (slot_qa >= 0 && slot_ib >= 0), "a Q-B slot needs the Q-A and I-B slots");
        if (!a->have_input) return set_error(GSH_ERR_STATE, "no dwell has run on this handle yet");
        if (a->d_grid == nullptr) return set_error(GSH_ERR_STATE, "the handle was created with no_grid = 1: the rows to be added are not stored");
        GSH_REQUIRE(!a->conf.bit_transition_flag, "the E5a block has no bit-transition search");
        GSH_HIP(hipSetDevice(a->device));
        if (a->d_pair == nullptr) GSH_HIP(hipMalloc(&a->d_pair, sizeof(gsh_acq_pair_peak) * a->n_bins));
        PairPeakArgs k;
        k.grid = a->d_grid;
        k.rows = a->d_rows;
        k.n = static_cast<int>(a->conf.effective_fft_size);
        k.n_bins = a->n_bins;
        k.ia = slot_ia;
        k.qa = slot_qa;
        k.ib = slot_ib;
        k.qb = slot_qb;
        const float fnf = static_cast<float>(a->padded ? a->logical_n : a->conf.fft_size) * static_cast<float>(a->padded ? a->logical_n : a->conf.fft_size);
        k.divisor = fnf * fnf;
        k.out = a->d_pair;
        hipLaunchKernelGGL(pair_peaks_kernel, dim3(a->n_bins), dim3(1024), 0, a->stream, k);
        GSH_HIP(hipGetLastError());
        GSH_HIP(hipMemcpyAsync(out, a->d_pair, sizeof(gsh_acq_pair_peak) * a->n_bins, hipMemcpyDeviceToHost, a->stream));
        GSH_HIP(hipStreamSynchronize(a->stream));
        return GSH_OK;
    }

    int gsh_acq_time_correlate(gsh_acq_t* a, const float* code_iq, uint32_t code_len, uint32_t doppler_index, const uint32_t* delays, uint32_t n_delays,
        float* out_iq)
    {
        GSH_REQUIRE(a != nullptr && code_iq != nullptr && delays != nullptr && out_iq != nullptr, "null argument");
        if (!a->have_input) return set_error(GSH_ERR_STATE, "no input block resident");
        GSH_REQUIRE(code_len >= 1 && code_len <= a->conf.consumed_samples, "code_len %u outside 1..consumed_samples", code_len);
        GSH_REQUIRE(n_delays >= 1 && n_delays <= static_cast<uint32_t>(TC_MAX_DELAYS), "n_delays %u outside 1..%d", n_delays, TC_MAX_DELAYS);
        GSH_REQUIRE(doppler_index < static_cast<uint32_t>(a->n_bins), "doppler_index %u outside 0..%d", doppler_index, a->n_bins - 1);
        for (uint32_t i = 0; i < n_delays; i++)
            GSH_REQUIRE(static_cast<uint64_t>(delays[i]) + code_len <= a->conf.consumed_samples, "delay %u + code_len %u runs past the %u resident samples", delays[i],
                code_len, a->conf.consumed_samples);
        GSH_HIP(hipSetDevice(a->device));
        if (a->tc_code_len < code_len)
            {
                if (a->d_tc_code) (void)hipFree(a->d_tc_code);
                a->d_tc_code = nullptr;
                a->tc_code_len = 0;
                GSH_HIP(hipMalloc(&a->d_tc_code, sizeof(float2) * code_len));
                a->tc_code_len = code_len;
            }
        if (a->d_tc_delays == nullptr) GSH_HIP(hipMalloc(&a->d_tc_delays, sizeof(uint32_t) * TC_MAX_DELAYS));
        if (a->d_tc_out == nullptr) GSH_HIP(hipMalloc(&a->d_tc_out, sizeof(float2) * TC_MAX_DELAYS));
        GSH_HIP(hipMemcpyAsync(a->d_tc_code, code_iq, sizeof(float2) * code_len, hipMemcpyHostToDevice, a->stream));
        GSH_HIP(hipMemcpyAsync(a->d_tc_delays, delays, sizeof(uint32_t) * n_delays, hipMemcpyHostToDevice, a->stream));
        hipLaunchKernelGGL(time_correlate_kernel, dim3(n_delays), dim3(1024), 0, a->stream, a->d_in, a->d_tc_code, static_cast<int>(code_len), a->d_tc_delays,
            a->h_bins_hz[doppler_index], 1.0 / static_cast<double>(a->conf.fs_in), a->d_tc_out);
        GSH_HIP(hipGetLastError());
        GSH_HIP(hipMemcpyAsync(out_iq, a->d_tc_out, sizeof(float2) * n_delays, hipMemcpyDeviceToHost, a->stream));
        GSH_HIP(hipStreamSynchronize(a->stream));
        return GSH_OK;
    }

    int gsh_acq_time_dwells(gsh_acq_t* a, uint32_t n_prn, int reps, float* avg_ms)
    {
        GSH_REQUIRE(a != nullptr && avg_ms != nullptr, "null argument");
        GSH_REQUIRE(reps >= 1, "reps %d", reps);
        GSH_REQUIRE(n_prn >= 1 && n_prn <= a->conf.max_prn, "n_prn %u outside 1..%u", n_prn, a->conf.max_prn);
        if (!a->have_input) return set_error(GSH_ERR_STATE, "no input block resident: call gsh_acq_dwell[_device] once first");
        GSH_HIP(hipSetDevice(a->device));
        int rc = enqueue_dwell(a, n_prn, 0, 1);  // warm-up
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipEventRecord(a->ev0, a->stream));
        for (int i = 0; i < reps; i++)
            {
                rc = enqueue_dwell(a, n_prn, 0, 1);
                if (rc != GSH_OK) return rc;
            }
        GSH_HIP(hipEventRecord(a->ev1, a->stream));
        GSH_HIP(hipEventSynchronize(a->ev1));
        float ms = 0.0f;
        GSH_HIP(hipEventElapsedTime(&ms, a->ev0, a->ev1));
        *avg_ms = ms / static_cast<float>(reps);
        return GSH_OK;
    }

    int gsh_acq_time_dwells_pipelined(gsh_acq_t* a, uint32_t n_prn, int reps, float* avg_ms)
    {
        GSH_REQUIRE(a != nullptr && avg_ms != nullptr, "null argument");
        GSH_REQUIRE(reps >= 2, "reps %d (need >= 2)", reps);
        GSH_REQUIRE(n_prn >= 1 && n_prn <= a->conf.max_prn, "n_prn %u outside 1..%u", n_prn, a->conf.max_prn);
        if (!a->have_input) return set_error(GSH_ERR_STATE, "no input block resident: call gsh_acq_dwell[_device] once first");
        if (!a->onchip) return gsh_acq_time_dwells(a, n_prn, reps, avg_ms);  // the four-step path shares one scratch buffer
        if (a->split > 0 && !a->conf.use_cfar) return gsh_acq_time_dwells(a, n_prn, reps, avg_ms);  // two batches in flight would share the stored rows
        GSH_HIP(hipSetDevice(a->device));
        const gsh_acq_conf& c = a->conf;
        const size_t n = c.fft_size, D = static_cast<size_t>(a->n_bins), P = c.max_prn;
        // lanes: two by default.  (GSH_ACQ_LANES: profiles/oc_cell_annotated.txt)
        static const int want_lanes = [] { const char* e = std::getenv("GSH_ACQ_LANES"); return e != nullptr ? std::min(std::max(std::atoi(e), 2), static_cast<int>(gsh_acq::MAX_LANES)) : 2; }();
        {
            gsh_acq::Lane& l0 = a->lane[0];
            l0.stream = a->stream;
            l0.d_spectra = a->d_spectra;
            l0.d_rows = a->d_rows;
            l0.d_subrows = a->d_subrows;
            l0.d_waverows = a->d_waverows;
            l0.d_z = a->d_z;
            l0.d_results = a->d_results;
            l0.d_arrivals = a->d_arrivals;
        }
        while (a->n_lanes < want_lanes)
            {
                // (every resource only where the slot does not hold one yet: a call that failed part-way -- an allocation refused -- left what it had got in the slot, and
                //  the next call must finish that lane, not overwrite its pointers; the handle's destructor frees whatever a slot holds.  Round-5 review.)
                gsh_acq::Lane& ln = a->lane[a->n_lanes];
                if (ln.stream == nullptr)
                    {
                        std::vector<hipStream_t> others;
                        for (int l = 1; l < a->n_lanes; l++) others.push_back(a->lane[l].stream);
                        const int rc2 = make_concurrent_stream(a->stream, others, &ln.stream);  // a stream on another hardware queue than the lanes so far
                        if (rc2 != GSH_OK) return rc2;
                    }
                if (ln.d_spectra == nullptr) GSH_HIP(hipMalloc(&ln.d_spectra, sizeof(float2) * D * n));
                if (ln.d_rows == nullptr) GSH_HIP(hipMalloc(&ln.d_rows, sizeof(gsh::RowStat) * P * D));
                if (a->split > 0 && ln.d_subrows == nullptr) GSH_HIP(hipMalloc(&ln.d_subrows, sizeof(gsh::RowStat) * P * D * a->split));
                if (ln.d_waverows == nullptr)
                    GSH_HIP(hipMalloc(&ln.d_waverows, sizeof(gsh::RowStat) * P * D * static_cast<size_t>(std::max(a->split, 1)) * gsh::ONCHIP_MAX_WAVES));
                if (a->d_z != nullptr && ln.d_z == nullptr)
                    {
                        GSH_HIP(hipMalloc(&ln.d_z, sizeof(float2) * P * D * n));
                    }
                if (ln.d_results == nullptr) GSH_HIP(hipMalloc(&ln.d_results, sizeof(gsh::DevAcqResult) * P));
                if (ln.d_arrivals == nullptr) GSH_HIP(hipMalloc(&ln.d_arrivals, sizeof(unsigned) * P));
                GSH_HIP(hipMemset(ln.d_arrivals, 0, sizeof(unsigned) * P));
                if (ln.ev == nullptr) GSH_HIP(hipEventCreate(&ln.ev));
                a->n_lanes++;
            }
        const int lanes = want_lanes;
        // Decimation-in-time plans (round 6, session 43): the sub-cells WRITE n_prn x n_bins x N values of Z and the combine launch READS them back -- with one batch's
        // sub-cells beside the other's combine launch the memory sees both directions at once and the pair runs slower than one after the other (non-temporal Z:
        // 0.905 ms per 128 000-point batch against 0.82 alone).  So on the plans with non-temporal Z (pcps_onchip.hip, onchip_dit_nontemporal) only the FORWARD transforms of the next batch (41 x S work-groups, which leave
        // most of the device idle) overlap the previous batch: its cells wait for the previous batch's.  GSH_ACQ_DIT_ORDERED=0: free-running lanes, as before.
        static const bool dit_ordered = [] { const char* e = std::getenv("GSH_ACQ_DIT_ORDERED"); return e == nullptr || std::atoi(e) != 0; }();
        const bool ordered = dit_ordered && a->d_z != nullptr && gsh::onchip_dit_nontemporal(a->split);
        if (ordered)
            for (int l = 0; l < lanes; l++)
                if (a->lane[l].ev_cells == nullptr) GSH_HIP(hipEventCreateWithFlags(&a->lane[l].ev_cells, hipEventDisableTiming));
        auto enqueue_cells = [&](const gsh_acq::Lane& ln) -> int {
            // no_grid handles only: the batches in flight must not share the magnitude grid
            return gsh::onchip_correlate(static_cast<int>(n), ln.d_spectra, a->d_codes, a->d_grid, ln.d_rows, ln.d_subrows, ln.d_results, ln.d_arrivals, static_cast<int>(n_prn), a->n_bins,
                c.bit_transition_flag ? static_cast<int>(c.effective_fft_size) : 0, static_cast<int>(c.effective_fft_size), 0, 0, static_cast<int>(c.samples_per_chip), c.use_cfar, 1u, 1.0f, ln.stream,
                ln.d_z, ln.d_waverows);
        };
        int last_lane = -1;
        auto enqueue = [&](int l) -> int {
            const gsh_acq::Lane& ln = a->lane[l];
            GSH_REQUIRE(c.fold <= 1, "the pipelined timing loop does not fold");
            int rc = gsh::onchip_forward(static_cast<int>(n), a->d_in, 0, static_cast<int>(c.consumed_samples), 0, a->d_bins_hz,
                static_cast<double>(c.fs_in), ln.d_spectra, a->n_bins, ln.stream);
            if (rc != GSH_OK) return rc;
            if (ordered && last_lane >= 0 && last_lane != l) GSH_HIP(hipStreamWaitEvent(ln.stream, a->lane[last_lane].ev_cells, 0));
            rc = enqueue_cells(ln);
            if (rc != GSH_OK) return rc;
            if (ordered) GSH_HIP(hipEventRecord(ln.ev_cells, ln.stream));
            last_lane = l;
            return GSH_OK;
        };
        int rc = GSH_OK;
        for (int l = 0; l < lanes && rc == GSH_OK; l++) rc = enqueue(l);  // warm-up on every lane
        if (rc != GSH_OK) return rc;
        for (int l = 0; l < lanes; l++) GSH_HIP(hipStreamSynchronize(a->lane[l].stream));
        GSH_HIP(hipEventRecord(a->ev0, a->stream));
        for (int l = 1; l < lanes; l++) GSH_HIP(hipStreamWaitEvent(a->lane[l].stream, a->ev0, 0));
        for (int i = 0; i < reps; i++)
            {
                rc = enqueue(i % lanes);
                if (rc != GSH_OK) return rc;
            }
        for (int l = 1; l < lanes; l++)
            {
                GSH_HIP(hipEventRecord(a->lane[l].ev, a->lane[l].stream));
                GSH_HIP(hipStreamWaitEvent(a->stream, a->lane[l].ev, 0));
            }
        GSH_HIP(hipEventRecord(a->ev1, a->stream));
        GSH_HIP(hipEventSynchronize(a->ev1));
        float ms = 0.0f;
        GSH_HIP(hipEventElapsedTime(&ms, a->ev0, a->ev1));
        *avg_ms = ms / static_cast<float>(reps);
        return GSH_OK;
    }

    float gsh_acq_compute_threshold(float pfa, uint32_t effective_fft_size, uint32_t num_doppler_bins, uint32_t max_dwells)
    {
        // acq.cc:52-56
        const int num_bins = static_cast<int>(effective_fft_size * num_doppler_bins);
        const double prob = std::pow(1.0 - static_cast<double>(pfa), 1.0 / static_cast<double>(static_cast<float>(num_bins)));
        return static_cast<float>(2.0 * gamma_p_inv_impl(2.0 * static_cast<double>(max_dwells), prob));
    }
}
