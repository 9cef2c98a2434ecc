// DLL/PLL tracking loop closed on the device (MI355X / gfx950): gsh_trk_* of include/gnss_sdr_hip.h.
//
// What it replaces, per code period, in gnss-sdr's dll_pll_veml_tracking (trk.cc =
// src/algorithms/tracking/gnuradio_blocks/dll_pll_veml_tracking.cc, T/ = src/algorithms/tracking/libs/):
//   do_correlation_step      trk.cc:1232-1257   -> mcdev::correlate_window_std (csrc/mcorr_device.h)
//   run_dll_pll              trk.cc:1260-1324   -> discriminators of T/tracking_discriminators.cc:27-149,
//                                                  Tracking_FLL_PLL_filter::get_carrier_error (T/tracking_FLL_PLL_filter.cc:72-99),
//                                                  Tracking_loop_filter::apply (T/tracking_loop_filter.cc:63-98)
//   update_tracking_vars     trk.cc:1409-1483
//   consume_each(d_current_prn_length_samples)  trk.cc:2287
// The host block calls the correlator once per channel per period and runs ~100 lines of scalar loop arithmetic in
// between; on a GPU that is one launch + one synchronisation per period.  Here ONE work-group per channel walks
// through all the periods of a device-resident stream: 256 threads correlate the window, thread 0 then runs the loop
// arithmetic in the reference's own float / double mix, publishes the next NCO settings through LDS, and the next
// window starts -- no host involvement until the requested number of periods is done.  The two local replicas
// (pilot + data for track_pilot signals) stay in LDS for the whole launch.
// 1024 threads per channel: a channel owns at most one compute unit and every period is a dependent step, so the
// only way to hide the window's load latency is more waves on that unit (measured: 16.6 us per 25 000-sample period
// with 256 threads)
#ifndef GSH_TRK_THREADS
#define GSH_TRK_THREADS 1024
#endif
#define GSH_MC_THREADS GSH_TRK_THREADS
#include "exact_division.h"
#include "mcorr_device.h"
#include "sample_stream.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <type_traits>
#include <new>
#include <vector>

namespace gsh
{
namespace
{
using namespace mcdev;

// the reference's pi (GNSS ICD value, src/core/system_parameters/MATH_CONSTANTS.h:47-49)
constexpr double GNSS_PI_D = 3.1415926535898;
constexpr double GNSS_HALF_PI_D = GNSS_PI_D / 2.0;
constexpr double GNSS_TWO_PI_D = 2.0 * GNSS_PI_D;

struct LoopFilterState  // Tracking_loop_filter (T/tracking_loop_filter.h): coefficients + the two 4-deep rings
{
    float in_c[4], out_c[4];
    float in_h[4], out_h[4];
    int n_in, n_out, idx;
};

struct FllPllState  // Tracking_FLL_PLL_filter (T/tracking_FLL_PLL_filter.h)
{
    float w, x, w0p, w0p2, w0p3, w0f, w0f2, a2, a3, b3;
    int order;
};

struct SmootherState  // Exponential_Smoother (T/exponential_smoother.h:40-69); the init buffer is only ever summed front to back
{
    float alpha, one_minus_alpha, old_value, min_value, offset, init_sum;
    int samples_for_initialization, init_counter, initializing;
};

struct LockState  // what cn0_and_tracking_lock_status keeps between periods (trk.cc:1167-1224)
{
    float prompt_buffer[2 * GSH_MAX_CN0_SAMPLES];
    SmootherState cn0_smoother, carrier_lock_test_smoother;
    int cn0_estimation_counter, carrier_lock_fail_counter, code_lock_fail_counter, pull_in_latched;
    float cn0_db_hz;
    double carrier_lock_test;
    // Round 4: the C/N0 half and the carrier-lock half of cn0_and_tracking_lock_status run on two different waves (trk_loop_kernel).  Neither reads a
    // word the other writes, so what both need to know about the prompt buffer exists twice:
    int cn0_slot;              // cn0_estimation_counter % cn0_samples once the buffer is full (kept instead of divided for)
    int carr_counter, carr_slot, pull_in_latched_carr;  // the carrier-lock wave's copies of cn0_estimation_counter (while the buffer fills), cn0_slot, pull_in_latched
    float first_prompt[2];     // its copy of d_Prompt_buffer[0] -- the one element carrier_lock_detector(buffer, 1) looks at (trk.cc:1184)
    // symbol synchronisation / narrow tracking (trk.cc:2026-2104, state 4 :2197-2252, save_correlation_results :1486-1596)
    int state;                 // d_state: 2 or 4
    int cloop;                 // d_cloop
    int ring_count, ring_head; // d_Prompt_circular_buffer (capacity d_secondary_code_length)
    int current_symbol, current_data_symbol, flag_pll_180, acc_phase_initialized;
    float p_data_accu[2];
    float ring[2 * GSH_MAX_SECONDARY];
    // extended integration (states 3 / 4, trk.cc:2114-2149, 2156-2195, 2241-2251)
    float2 accv[5];            // d_VE_accu .. d_VL_accu across the coherent integration
    int ext_count;             // d_extend_correlation_symbols_count
    int narrow;                // narrow loop filters / correlator spacing are in force
    float spc_now;             // d_trk_parameters.spc
    double corr_time;          // d_current_correlation_time_s
    float dll_narrow_in_c[4], dll_narrow_out_c[4];  // Tracking_loop_filter coefficients for (extend * period, dll_bw_narrow_hz), designed at start
    int dll_narrow_n_in, dll_narrow_n_out;
    FllPllState pll_narrow;    // Tracking_FLL_PLL_filter::set_params(fll_bw_hz, pll_bw_narrow_hz, order): coefficients only
    // HistogramBitSynchronizer (T/bit_synchronizer.{h,cc}) + the block's use of it (trk.cc:2046-2072)
    int bs_hist[GSH_MAX_BITSYNC_BINS];
    int bs_total_events, bs_locked, bs_edge_phase, bs_has_last_prompt, bs_has_last_sign, bs_last_sign, bs_has_last_best_bin, bs_last_best_bin, bs_stable_best_count;
    long long bs_epoch_count, bs_target_epoch;
    float bs_last_prompt[2];
    int use_hist, wait_for_bit_edge;
    // high dynamics: boost::circular_buffer<pair<step, samples>>(2 * smoother_length) for the carrier and the code NCO (trk.cc:698-699)
    double carr_hist[2 * GSH_MAX_SMOOTHER][2], code_hist[2 * GSH_MAX_SMOOTHER][2];
    int carr_hist_n, code_hist_n;
    long long carr_pushes, code_pushes;
    // experimental Doppler correction (trk.cc:1326-1346): d_dll_filt_history is only ever filled and cleared, and std::accumulate(begin, end, 0.0) adds its
    // floats to a double in push order -- a running double sum is the same arithmetic
    double dll_filt_sum;
    int dll_filt_count, corrected_doppler;
};

struct TrkChannel  // loop state of one channel, resident in device memory between launches
{
    double carrier_doppler_hz, carrier_phase_step_rad, code_freq_chips, code_phase_step_chips;
    double rem_code_phase_samples, rem_code_phase_chips, acc_carrier_phase_rad;
    double carrier_phase_rate_step_rad, code_phase_rate_step_chips;  // high_dyn (trk.cc:1425-1443, 1458-1480); 0 otherwise
    unsigned long long pos, acq_stamp;  // acq_stamp: bit 63 set = the pull-in transitory was over at the hand-over call (GSH_TRK_START_PULL_IN_OVER); the stamp is the rest
    float rem_carr_phase_rad, p_old_re, p_old_im;
    int active, code_len;
    LoopFilterState dll;
    FllPllState pll;
};

struct TrkTail  // what the host needs back from every channel after a launch, in one small contiguous copy
{
    unsigned long long pos;  // TrkChannel::pos: first sample of the next window
    int done;                // periods completed in this launch
    int active;              // TrkChannel::active
};

struct HotConstants  // the configuration's values thread 0 needs in every period, in LDS: they come in with its state in ONE pinned batch of reads (each was an s_load the lane waited for)
{
    double code_chip_rate, signal_carrier_freq, fs_in, cfo_frequency_hz;
    unsigned code_length_chips, vector_length;
    float code_samples_per_chip_f;
    int pad;
};
struct SerialMail  // results the code-loop lane and the lock-detector lane hand to thread 0 (trk_loop_kernel)
{
    double code_error_chips, code_error_filt_chips;
    int lost;          // the code lock fail counter is over its limit (C/N0 wave)
    int lost_carrier;  // the carrier lock fail counter is (carrier-lock wave)
    int coop_err;      // COOP kernels: a partner work-group did not answer in time (the loop ends at the top of the next period)
    // The lanes do not meet at a barrier: each says when it is done by writing the period's number (LDS operations of one wave are carried out in order, so a lane
    // that has seen the number sees what was written before it), and thread 0 waits only for what it needs -- the code loop's output always, the lock detectors'
    // verdict only in a period in which a fail counter CAN pass its limit (may_trip_*: left by the detectors' lanes for the next period).  Otherwise the two
    // detector lanes -- the longest of the four -- run beside thread 0's join / update_tracking_vars and are met at the barrier that ends the period.
    // (what thread 0 looks at when it has done its own part, side by side: ONE 32-byte read)
    alignas(16) int code_seq;
    int inputs_cn0, inputs_carr;        // the detector lanes have read the state thread 0 goes on to change (form_inputs)
    int pad0;
    int may_trip_code, may_trip_carr;
    int cn0_seq, carr_seq;
    // thread 0 -> the two seed-table waves: the next window's carrier phase and phase step are final (right behind update_tracking_vars)
    int step_seq;
    float seed_step, seed_rem;
#ifdef GSH_TRK_PROFILE
    long long t_lane[4];  // when each of the four lanes was done, in clocks since the correlation ended (-DGSH_TRK_PROFILE=3 puts them into the record)
#endif
};
// true when the code lane and the two `inputs` words say `seq`; may_trip = may_trip_code | may_trip_carr as read in the same go
__device__ __forceinline__ bool join_ready(SerialMail& m, int seq, int& may_trip)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef int i2 __attribute__((ext_vector_type(2)));
    const i4 lo = *reinterpret_cast<volatile i4*>(&m.code_seq);
    const i2 hi = *reinterpret_cast<volatile i2*>(&m.may_trip_code);
    may_trip = hi[0] | hi[1];
    return lo[0] == seq && lo[1] == seq && lo[2] == seq;
}
__device__ __forceinline__ void lane_says(int& word, int value)
{
    asm volatile("" ::: "memory");
    __hip_atomic_store(&word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lane_waits(int& word, int value)
{
    while (__hip_atomic_load(&word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != value) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
#ifndef GSH_TRK_SEED_TABLES
#define GSH_TRK_SEED_TABLES 1  // the lanes' seeds from tables two idle waves fill beside thread 0's end of the period (0: every lane evaluates its own; A/B)
#endif
#ifndef GSH_TRK_SERIAL_WAVES
#define GSH_TRK_SERIAL_WAVES 4
#endif
#ifndef GSH_TRK_PREFIX_ALL
#define GSH_TRK_PREFIX_ALL 1
#endif
constexpr int SERIAL_WAVES = GSH_TRK_SERIAL_WAVES;  // wave 0 carrier loop, wave 1 code loop, wave 2 C/N0 estimator, wave 3 carrier lock test + the record's early fields
constexpr int CN0_WAVE = 2, CARR_LOCK_WAVE = SERIAL_WAVES - 1;  // (three serial waves: both halves on wave 2, one after the other)
// the seed tables are filled by waves SERIAL_WAVES + 1 and SERIAL_WAVES + 2, one entry per lane (16 W entries: work-groups of up to 1 024 threads): a build with
// fewer waves would read tables nobody fills (round-5 review)
#if defined(GSH_TRK_SEED_TABLES) && !GSH_TRK_SEED_TABLES
#else
static_assert(GSH_MC_THREADS / 64 >= GSH_TRK_SERIAL_WAVES + 3 && GSH_MC_THREADS <= 1024, "the closed-loop kernel's seed tables need waves SERIAL_WAVES + 1 and + 2, and cover 16 waves");
#endif

// ---- live mode (gsh_trk_live_*): the kernel stays resident and follows the ring as it fills ------------------------------------------------------
// A launch per batch of periods costs the host ~270 us of queueing and waiting around ~180 us of kernel (round 3) -- at the reference's cadence
// of one period per general_work call that is all there is.  In live mode ONE launch ("residency") serves a channel for as long as samples keep coming:
//   * the ring says how far it is complete through two words in device memory that every push rewrites behind its copies (sample_stream.hip,
//     publish_live_kernel); an idle lane reads them during the loop arithmetic of every period, thread 0 polls them only when it has run out of samples;
//   * every period's record goes into a per-channel ring of records in page-locked host memory, and LiveTail::seq -- stored once the records below it
//     are known to have left the device (see the drain rule at the loop's top) -- tells the host how many there are: the host never waits for a stream;
//   * a residency ends by itself: nothing new for idle_ticks, residency_ticks used up (so that whatever waits for the device -- hipFree, a device-wide
//     synchronisation -- gets its turn), the host's quit word, the channel stopped, or the record ring full.  It can therefore never hang the device.
enum
{
    LIVE_EXIT_NONE = 0,
    LIVE_EXIT_IDLE = 1,      // no new samples within idle_ticks
    LIVE_EXIT_BUDGET = 2,    // residency_ticks used up
    LIVE_EXIT_QUIT = 3,      // the host asked (start / stop of a channel, destroy)
    LIVE_EXIT_INACTIVE = 4,  // the channel is not tracking (never started, stopped, lost lock)
    LIVE_EXIT_OVERRUN = 5    // the channel's next window has been overwritten in the ring
};
struct LiveArgs
{
    const unsigned long long* head;      // device memory: [0] one past the newest complete sample, [1] first resident index since the last seek
    LiveTail* tail;                      // host memory (mapped), n_channels; nullptr: not a live launch
    const unsigned long long* consumed;  // host memory (mapped), n_channels: records the host has taken (flow control of the record ring)
    const int* quit;                     // host memory (mapped)
    unsigned ring_len;                   // records per channel, a power of two
    unsigned long long idle_ticks, residency_ticks;  // of wall_clock64() (100 MHz)
};
struct LiveShared  // the live form's own words in LDS (the launched form has none of them: its LDS layout, and with it its register allocation, stay what they were)
{
    unsigned long long seq;       // periods completed by this channel = index of the record the current period writes
    unsigned long long consumed;  // the host's count as last read
    unsigned long long t_start;   // wall_clock64() when the residency began
    unsigned long long now;       // ... as the look-out lane last read it
    // what the look-out lane (lane 0 of the first wave without loop arithmetic) read from the ring's live words while the others worked
    unsigned long long head, origin, head_fenced;
    unsigned long long pub_pos;   // the channel's next window as it goes out with the record
    unsigned long long wpos;      // ... and where that window sits in the ring's memory (pub_pos mod capacity, kept up incrementally: a 64-bit remainder
                                  // evaluated by 1 024 threads at the top of every period is a microsecond of vector instructions)
    int quit;
    int exit_reason;
    int out_valid;  // the record in LDS is complete and has not been written to the host's ring yet
    int pad_;
};
struct NoLiveShared
{
};

struct TrkArgs
{
    const gsh_trk_conf* conf;  // device copy (a by-value struct with dynamically indexed arrays would be materialised in scratch by every thread)
    const float2* stream;
    unsigned long long n_stream;       // flat buffer: its length; ring: absolute index one past the newest resident sample
    unsigned long long ring_capacity;  // 0: flat buffer; else absolute sample i lives at stream[i % ring_capacity] (windows are contiguous: mirror)
    unsigned long long ring_oldest;    // ring: absolute index of the oldest resident sample
    const float* codes;       // n_channels * 2 * code_stride (pilot/primary code, then data code)
    int code_stride;
    TrkChannel* chan;
    LockState* lock;          // n_channels (used with conf.enable_lock_detectors)
    gsh_trk_epoch* records;   // n_channels * n_epochs or nullptr
    TrkTail* tail;            // n_channels
    int n_epochs;
    // loop invariants the host forms once per launch with the expressions the kernel used to evaluate in every period
    double code_period;                 // d_code_period = code_length_chips / code_chip_rate
    unsigned long long pull_in_limit;   // samples since acquisition below which the pull-in transitory lasts (trk.cc:1912-1915), see trk_launch
    unsigned long long bit_sync_limit;  // samples since acquisition from which a channel still in state 2 is declared lost (trk.cc:2000-2007); ~0: never
    // correctly rounded reciprocals of the two launch-constant divisors of the loop arithmetic, 0.0 when the configuration does not qualify (div_by_constant below)
    double inv_fs_in, inv_signal_carrier_freq;
    LiveArgs live;
    // cooperating work-groups (COOP kernels, gsh_trk_set_split): coop_g work-groups share every window of a channel.  coop_box: per channel COOP_STRIDE(g) 64-bit
    // words (value | tag << 32) -- the next window's seven words from the channel's main work-group, then each helper's 2 (NT + 1) partial sums; the word behind
    // the last channel's is the launch's error flag.  coop_seq0: the tag of this launch's first window (tags never repeat over a handle's life).
    unsigned long long* coop_box;
    int coop_g;
    unsigned coop_seq0;
    int coop_channels;
    int coop_helper_trips;   // > 0: trips of a helper's segment (A/B override, GSH_TRK_SPLIT_HELPER_TRIPS)
    __host__ __device__ int coop_n_channels() const { return coop_channels; }
};
constexpr int COOP_BOX_WORDS = 8;       // pos lo, pos hi, rem_carr, phase_step, rem_code, code_step, flags (go | narrow << 1), spare
constexpr int COOP_PART_WORDS = 16;     // a helper's sums: 2 (NT + 1) <= 12 words
__host__ __device__ constexpr int coop_stride(int g) { return COOP_BOX_WORDS + COOP_PART_WORDS * (g > 1 ? g - 1 : 0); }
constexpr unsigned long long COOP_TIMEOUT_TICKS = 20000000ull;  // 0.2 s of wall_clock64 (100 MHz): a partner that never answers must not hang the device

constexpr double INV_TWO_PI_D = 1.0 / GNSS_TWO_PI_D;  // correctly rounded by the compiler; 2 pi's significand is not all ones
// fmod(x, 2 pi) of the carrier phase remainder (a few turns): exact_division.h; arguments beyond a million turns take the library's path
template <bool FAST>
__device__ __forceinline__ double fmod_two_pi(double x)
{
    if constexpr (FAST && GSH_TRK_FAST_DIV != 0)
        {
            bool slow;
            const double r = fmod_by_constant(x, GNSS_TWO_PI_D, INV_TWO_PI_D, &slow);
            if (__builtin_expect(!slow, 1)) return r;
        }
    return fmod(x, GNSS_TWO_PI_D);
}
// div_by_constant with the "does it apply" decision taken by the caller at compile time (y is then known to be the reciprocal)
template <bool FAST>
__device__ __forceinline__ double div_by_constant_if(double a, double b, double y)
{
    if constexpr (FAST && GSH_TRK_FAST_DIV != 0) return div_by_constant_vetted(a, b, y);
    return a / b;
}

// ---- discriminators, T/tracking_discriminators.cc (float / double mix as written there) ---------------------
__device__ __forceinline__ double phase_unwrap_d(double p)  // :27-41
{
    if (p >= GNSS_HALF_PI_D) return p - GNSS_PI_D;
    if (p <= -GNSS_HALF_PI_D) return p + GNSS_PI_D;
    return p;
}
__device__ __forceinline__ double fll_diff_atan_d(float2 p1, float2 p2, double t1, double t2)  // :68-76, float arctangents
{
    double d = atanf(p2.y / p2.x) - atanf(p1.y / p1.x);
    if (isnan(d)) d = 0.0;
    return phase_unwrap_d(d) / (t2 - t1);
}
__device__ __forceinline__ double pll_cloop_two_quadrant_atan_d(float2 p)  // :99-106
{
    if (p.x != 0.0f) return static_cast<double>(atanf(p.y / p.x));
    return 0.0;
}
// :86-89; the reference calls gr::fast_atan2f (GNU Radio's table approximation, not vendored): exact atan2f here
__device__ __forceinline__ double pll_four_quadrant_atan_d(float2 p) { return static_cast<double>(atan2f(p.y, p.x)); }
__device__ __forceinline__ double dll_nc_e_minus_l_normalized_d(float2 e, float2 l, float spc, float slope, float y_intercept)  // :117-127
{
    const double pe = static_cast<double>(hypotf(e.x, e.y));
    const double pl = static_cast<double>(hypotf(l.x, l.y));
    const double s = pe + pl;
    if (s == 0.0) return 0.0;
    return ((y_intercept - slope * spc) / slope) * (pe - pl) / s;
}
__device__ __forceinline__ double dll_nc_vemlp_normalized_d(float2 ve, float2 e, float2 l, float2 vl)  // :139-149
{
    const double early = static_cast<double>(sqrtf(ve.x * ve.x + ve.y * ve.y + e.x * e.x + e.y * e.y));
    const double late = static_cast<double>(sqrtf(l.x * l.x + l.y * l.y + vl.x * vl.x + vl.y * vl.y));
    const double s = early + late;
    if (s == 0.0) return 0.0;
    return (early - late) / s;
}

// ---- loop filters ---------------------------------------------------------------------------------------------
// The two 4-deep histories are kept newest first (a shift register) instead of as the reference's ring with a moving index: in_h[i] / out_h[i] here are its
// d_inputs[(d_current_index + i) % 4] / d_outputs[...] -- the same products added in the same order, without a run-time index into LDS (each a dependent round trip
// on the code lane; idx stays 0).  The state comes out of LDS in one go (PIN: pinned up front, as in join_and_update).
template <bool PIN>
__device__ __forceinline__ float loop_filter_apply(LoopFilterState& fs, float x)  // T/tracking_loop_filter.cc:63-98
{
    LoopFilterState f = fs;
    if constexpr (PIN)
        asm volatile("" : "+v"(f.in_c[0]), "+v"(f.in_c[1]), "+v"(f.in_c[2]), "+v"(f.in_c[3]), "+v"(f.out_c[0]), "+v"(f.out_c[1]), "+v"(f.out_c[2]), "+v"(f.out_c[3]),
                     "+v"(f.in_h[0]), "+v"(f.in_h[1]), "+v"(f.in_h[2]), "+v"(f.out_h[0]), "+v"(f.out_h[1]), "+v"(f.out_h[2]), "+v"(f.n_in), "+v"(f.n_out));
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < f.n_out) r += f.out_c[i] * f.out_h[i];
    const float ih[4] = {x, f.in_h[0], f.in_h[1], f.in_h[2]};
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < f.n_in) r += f.in_c[i] * ih[i];
#pragma unroll
    for (int i = 0; i < 4; i++) fs.in_h[i] = ih[i];
    fs.out_h[0] = r;
#pragma unroll
    for (int i = 1; i < 4; i++) fs.out_h[i] = f.out_h[i - 1];
    return r;
}
__device__ __forceinline__ float fll_pll_carrier_error(FllPllState& f, float fll_disc, float pll_disc, float t)  // T/tracking_FLL_PLL_filter.cc:72-99
{
    float out;
    if (f.order == 3)
        {
            f.w = f.w + t * (f.w0p3 * pll_disc + f.w0f2 * fll_disc);
            f.x = f.x + t * (0.5f * f.w + f.a2 * f.w0f * fll_disc + f.a3 * f.w0p2 * pll_disc);
            out = 0.5f * f.x + f.b3 * f.w0p * pll_disc;
        }
    else
        {
            const float w_new = f.w + pll_disc * f.w0p2 * t + fll_disc * f.w0f * t;
            out = 0.5f * (w_new + f.w) + f.a2 * f.w0p * pll_disc;
            f.w = w_new;
        }
    return out;
}

// A zero the compiler cannot share: it kept ONE 64-bit zero in a VGPR pair from the kernel's first block to the state transitions that clear the
// accumulators -- across the whole epoch loop --, and with 128 VGPRs for 1 024 threads that pair went to scratch.
__device__ __forceinline__ float2 fresh_zero2()
{
    float2 z;
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(z.x), "=v"(z.y));
    return z;
}

// ---- lock detectors and C/N0 (T/lock_detectors.cc, T/exponential_smoother.cc), float32 and sequential as written there ----
// cn0_m2m4_estimator's three sums -- Psig = sum |re|, m_2 = sum (im^2 + re^2), m_4 = sum (im^2 + re^2)^2, each added up in buffer order starting from 0.0f
// (T/lock_detectors.cc:68-80) -- formed by the WHOLE wave: lane i holds the terms of buffered prompt i, and a step hands the running sums one lane up
// (v_add_f32_dpp row_shr:1: s[l] = s[l-1] + x[l]; the first lane of a row has no source and keeps what it holds).  Once lane l-1 holds its prefix
// ((x0 + x1) + ...) + x(l-1), the next step leaves lane l with that plus x(l) -- the additions of the one-thread loop, in its order (0.0f + x0 is x0 for
// these non-negative terms) -- and every further step recomputes the same value: fifteen steps settle a row of sixteen lanes whatever n is, no loop, no count.
// Rows go one after the other; the first lane of the next row gets the previous row's last prefix over v_readlane.  On return lane n-1 holds the totals.
// One thread walking the LDS buffer took ~1 700 clocks for 20 prompts (ten dependent LDS round trips); this is 2 x 45 instructions.  (Round 3 tried the lanes
// with one v_readlane per term: slower than the loop.)
#define GSH_M2M4_STEP                                                        \
    "v_add_f32_dpp %0, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"      \
    "v_add_f32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"      \
    "v_add_f32_dpp %2, %2, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void m2m4_sums_wave(float& psig, float& m2, float& m4, int n)
{
    const float xa = psig, xb = m2, xc = m4;
#pragma clang loop unroll(disable)
    for (int base = 0; base < n; base += 16)
        {
            if (base > 0)
                {
                    psig = __fadd_rn(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, psig), base - 1)), xa);
                    m2 = __fadd_rn(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m2), base - 1)), xb);
                    m4 = __fadd_rn(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m4), base - 1)), xc);
                }
            // (a DPP operand written by the instruction before wants two wait states -- the compiler does not look into this text: s_nop 1 up front; inside,
            // every sum is read three instructions after it was written)
            asm volatile("s_nop 1\n\t" GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP
                             GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP GSH_M2M4_STEP
                         : "+v"(psig), "+v"(m2), "+v"(m4)
                         : "v"(xa), "v"(xb), "v"(xc));
        }
}
#undef GSH_M2M4_STEP

// the rest of cn0_m2m4_estimator (T/lock_detectors.cc:81-110) on the three sums
__device__ __forceinline__ float cn0_from_sums_d(float Psig, float m_2, float m_4, int length, float coh_integration_time_s)
{
    float SNR_aux = 0.0f, aux;
    const float n = static_cast<float>(length);
    Psig = __fdiv_rn(Psig, n);
    Psig = __fmul_rn(Psig, Psig);
    m_2 = __fdiv_rn(m_2, n);
    m_4 = __fdiv_rn(m_4, n);
    // sqrtf, not __fsqrt_rn: hipcc renders the intrinsic as a bare v_sqrt_f32 (one ulp), sqrtf as the correctly rounded sequence (v_sqrt_f32 + two residual tests) -- what
    // std::sqrt in T/lock_detectors.cc:93 is.  (Found in round 4 with a buffer of ONE prompt, where the estimator's denominator is m_2 - sqrt(m_2^2): zero, or one ulp.)
    aux = sqrtf(__fsub_rn(__fmul_rn(__fmul_rn(2.0f, m_2), m_2), m_4));
    float denominator;
    if (isnan(aux))
        {
            denominator = __fsub_rn(m_2, Psig);
            if (denominator == 0.0f) return -100.0f;
            SNR_aux = __fdiv_rn(Psig, denominator);
        }
    else
        {
            denominator = __fsub_rn(m_2, aux);
            if (denominator == 0.0f) return -100.0f;
            SNR_aux = __fdiv_rn(aux, denominator);
        }
    if (SNR_aux == 0.0f) return -100.0f;
    return __fsub_rn(__fmul_rn(10.0f, log10f(SNR_aux)), __fmul_rn(10.0f, log10f(coh_integration_time_s)));
}

__device__ float carrier_lock_detector_d(const float* prompt_iq, int length)  // T/lock_detectors.cc:113-133
{
    float si = 0.0f, sq = 0.0f;
    for (int i = 0; i < length; i++)
        {
            si = __fadd_rn(si, prompt_iq[2 * i]);
            sq = __fadd_rn(sq, prompt_iq[2 * i + 1]);
        }
    const float nbp = __fadd_rn(__fmul_rn(si, si), __fmul_rn(sq, sq));
    const float nbd = __fsub_rn(__fmul_rn(si, si), __fmul_rn(sq, sq));
    if (nbp == 0.0f) return 0.0f;
    return __fdiv_rn(nbd, nbp);
}

// T/exponential_smoother.cc:83-112.  The state comes out of LDS in one go and only what changed goes back: field-by-field accesses were a chain of LDS round trips
// on the one lane the period waits for.
__device__ __forceinline__ float smoother_smooth_d(SmootherState& st, float raw)
{
    SmootherState s = st;
    float smoothed;
    if (s.initializing)
        {
            s.init_counter++;
            smoothed = raw;
            s.init_sum = __fadd_rn(s.init_sum, smoothed);
            if (s.init_counter == s.samples_for_initialization)
                {
                    s.old_value = __fdiv_rn(s.init_sum, static_cast<float>(s.init_counter));
                    st.old_value = s.old_value;
                    if (s.old_value < __fadd_rn(s.min_value, s.offset))
                        {
                            s.init_counter = 0;  // flush buffer and start again
                            s.init_sum = 0.0f;
                        }
                    else
                        {
                            st.initializing = 0;
                        }
                }
            st.init_counter = s.init_counter;
            st.init_sum = s.init_sum;
        }
    else
        {
            smoothed = __fadd_rn(__fmul_rn(s.alpha, raw), __fmul_rn(s.one_minus_alpha, s.old_value));
            st.old_value = smoothed;
        }
    return smoothed;
}

// cn0_and_tracking_lock_status, trk.cc:1167-1224, in two halves that share no word of state (each returns true when ITS fail counter is over its limit; the block's
// verdict is their OR, and it is formed -- and both counters are cleared, :1205-1206 -- by thread 0 after the join):
//   C/N0 half: the prompt goes into the buffer, M2M4 estimate from the sums the wave has formed (have_sums; otherwise the buffer is still filling), smoother, code lock counter
__device__ __forceinline__ bool cn0_half_d(LockState& st, const gsh_trk_conf& c, float2 P, int cnt, int slot, float Psig, float m_2, float m_4, double coh_integration_time_s,
    bool pull_in_transitory)
{
    const int ns = c.cn0_samples;
    st.cn0_estimation_counter = cnt + 1;
    if (cnt < ns)
        {
            st.prompt_buffer[2 * cnt] = P.x;
            st.prompt_buffer[2 * cnt + 1] = P.y;
            return false;
        }
    st.prompt_buffer[2 * slot] = P.x;
    st.prompt_buffer[2 * slot + 1] = P.y;
    st.cn0_slot = (slot + 1 == ns) ? 0 : slot + 1;
    const float coh = static_cast<float>(coh_integration_time_s);
    const float cn0_raw = (coh == 0.0f) ? -100.0f : cn0_from_sums_d(Psig, m_2, m_4, ns, coh);  // (length == 0 cannot happen: cn0_samples >= 1)
    const float cn0 = smoother_smooth_d(st.cn0_smoother, cn0_raw);
    st.cn0_db_hz = cn0;
    int fails = st.code_lock_fail_counter;
    if (!pull_in_transitory)
        {
            if (cn0 < static_cast<float>(c.cn0_min))
                fails++;
            else if (fails > 0)
                fails--;
            st.code_lock_fail_counter = fails;
        }
    return fails > c.max_code_lock_fail;
}

//   carrier-lock half: carrier_lock_detector(d_Prompt_buffer.data(), 1) -- length ONE, as the reference calls it (trk.cc:1184): the buffer's FIRST element, whichever
//   period put it there --, smoother, carrier lock counter.  The wave keeps its own copy of that element and of the buffer's fill / write position.
__device__ __forceinline__ bool carrier_lock_half_d(LockState& st, const gsh_trk_conf& c, float2 P, bool pull_in_transitory)
{
    const int ns = c.cn0_samples;
    const int cnt = st.carr_counter;
    if (cnt < ns)
        {
            if (cnt == 0)
                {
                    st.first_prompt[0] = P.x;
                    st.first_prompt[1] = P.y;
                }
            st.carr_counter = cnt + 1;
            return false;
        }
    const int slot = st.carr_slot;
    float first[2] = {st.first_prompt[0], st.first_prompt[1]};
    if (slot == 0)
        {
            first[0] = P.x;
            first[1] = P.y;
            st.first_prompt[0] = P.x;
            st.first_prompt[1] = P.y;
        }
    st.carr_slot = (slot + 1 == ns) ? 0 : slot + 1;
    const double test = static_cast<double>(smoother_smooth_d(st.carrier_lock_test_smoother, carrier_lock_detector_d(first, 1)));
    st.carrier_lock_test = test;
    int fails = st.carrier_lock_fail_counter;
    if (!pull_in_transitory)
        {
            if (test < c.carrier_lock_th)
                fails++;
            else if (fails > 0)
                fails--;
            st.carrier_lock_fail_counter = fails;
        }
    return fails > c.max_carrier_lock_fail;
}

// HistogramBitSynchronizer::update, T/bit_synchronizer.cc:41-124: true on the lock event
__device__ bool bit_sync_update_d(LockState& b, const gsh_trk_conf& c, float2 p, bool tracking_quality_ok)
{
    const int N = c.symbols_per_bit;  // bins() = bit_period_ms / epoch_ms (trk.cc:1397-1398)
    const int phase = (N > 0) ? static_cast<int>(b.bs_epoch_count % N) : 0;
    ++b.bs_epoch_count;
    if (!tracking_quality_ok || (hypotf(p.x, p.y) < c.bs_min_prompt_mag))
        {
            b.bs_last_prompt[0] = p.x;
            b.bs_last_prompt[1] = p.y;
            b.bs_has_last_prompt = 1;
            return false;
        }
    bool edge_event = false;
    if (c.bs_use_phase_dot_detector)
        {
            if (b.bs_has_last_prompt)
                {
                    const float dot = __fadd_rn(__fmul_rn(p.x, b.bs_last_prompt[0]), __fmul_rn(p.y, b.bs_last_prompt[1]));  // Re(Pk conj(Pk-1))
                    edge_event = static_cast<double>(dot) < 0.0;
                }
            b.bs_last_prompt[0] = p.x;
            b.bs_last_prompt[1] = p.y;
            b.bs_has_last_prompt = 1;
        }
    else
        {
            const int sgn = (p.x >= 0.0f) ? +1 : -1;
            if (b.bs_has_last_sign) edge_event = (sgn != b.bs_last_sign);
            b.bs_last_sign = sgn;
            b.bs_has_last_sign = 1;
        }
    if (edge_event && N > 0)
        {
            ++b.bs_hist[phase];
            ++b.bs_total_events;
        }
    if (!b.bs_locked && (b.bs_total_events >= c.bs_min_events_for_lock))
        {
            int best_bin = 0, best_count = N > 0 ? b.bs_hist[0] : 0;
            for (int i = 1; i < N; i++)
                if (b.bs_hist[i] > best_count)
                    {
                        best_count = b.bs_hist[i];
                        best_bin = i;
                    }
            const double ratio = (b.bs_total_events > 0) ? (static_cast<double>(best_count) / static_cast<double>(b.bs_total_events)) : 0.0;
            if (!b.bs_has_last_best_bin || (best_bin != b.bs_last_best_bin))
                {
                    b.bs_last_best_bin = best_bin;
                    b.bs_has_last_best_bin = 1;
                    b.bs_stable_best_count = 1;
                }
            else
                {
                    ++b.bs_stable_best_count;
                }
            if ((ratio >= c.bs_dominance_ratio) && (b.bs_stable_best_count >= c.bs_stable_best_required))
                {
                    b.bs_locked = 1;
                    b.bs_edge_phase = best_bin;
                    return true;
                }
        }
    return false;
}

struct NextWindow  // what thread 0 publishes for the next correlation (do_correlation_step's casts, trk.cc:1237-1243)
{
    unsigned long long pos;
    float rem_carr, phase_step, rem_code, code_step;
    float phase_rate, code_rate;  // high_dyn only
    int go;      // 1: correlate the window; 0: leave the loop; 2 (live mode only): drain the record stores, publish, and wait for samples (live_wait)
    int narrow;  // correlate with the narrow tap spacing (after extended integration has started)
    int seed_ok; // the seed tables in LDS were formed from exactly this rem_carr and phase_step (GSH_TRK_SEED_TABLES)
};

__device__ __forceinline__ void publish(NextWindow& w, const TrkChannel& s, const gsh_trk_conf& c, unsigned long long n_stream, int more,
    unsigned long long ring_oldest = 0ull)
{
    const float spcf = static_cast<float>(c.code_samples_per_chip);
    w.pos = s.pos;
    w.rem_carr = s.rem_carr_phase_rad;
    w.phase_step = static_cast<float>(s.carrier_phase_step_rad);
    w.rem_code = __fmul_rn(static_cast<float>(s.rem_code_phase_chips), spcf);
    w.code_step = __fmul_rn(static_cast<float>(s.code_phase_step_chips), spcf);
    w.phase_rate = static_cast<float>(s.carrier_phase_rate_step_rad);
    w.code_rate = __fmul_rn(static_cast<float>(s.code_phase_rate_step_chips), spcf);
    w.go = (more && s.active && s.pos + c.vector_length <= n_stream && s.pos >= ring_oldest) ? 1 : 0;
    w.narrow = 0;  // the caller overrides it from the channel's LockState
    w.seed_ok = 0;
}

// A field of a period's record: in the launched form a store into the launch's block of records (device or host memory; the end of the kernel makes it
// visible), in the live form a store into the record's copy in LDS (see the write-out at the loop's top).
template <bool LIVE, typename T, typename U>
__device__ __forceinline__ void rec_set(T& dst, U value)
{
    dst = static_cast<T>(value);
}

// ---- live mode, thread 0 only ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long live_oldest(unsigned long long head, unsigned long long origin, unsigned long long capacity)
{
    const unsigned long long by_capacity = head > capacity ? head - capacity : 0ull;
    return by_capacity > origin ? by_capacity : origin;
}

// End of a period (after publish() has formed the next window's NCO settings): the period's record is complete in LDS; whether the next window can be
// correlated straight away is decided from what the look-out lane read at the end of the correlation -- no memory access on this path.  Anything else (no
// samples yet, record ring full, budget used up, channel stopped, quit) is left to the drain round at the loop's top (win.go = 2).
__device__ __forceinline__ void live_advance(const TrkArgs& a, unsigned long long s_pos, int s_active, NextWindow& w, LiveShared& v, unsigned vlen)
{
    // (no store into host memory here: thread 0 would wait for its acknowledgement -- a PCIe round trip -- at the barrier that ends the period.  The
    // channel's position and the count of complete records go out with the record, from the wave that writes it at the top of the next period.)
    // The words come out of LDS in one go (pinned by the asm: read one by one in front of their uses they were half a dozen dependent round trips on thread 0's path).
    unsigned long long seq = v.seq, consumed = v.consumed, t_start = v.t_start, now = v.now, head = v.head, origin = v.origin, pub_pos = v.pub_pos, wpos = v.wpos;
    int quit = v.quit;
    asm volatile("" : "+v"(seq), "+v"(consumed), "+v"(t_start), "+v"(now), "+v"(head), "+v"(origin), "+v"(pub_pos), "+v"(wpos), "+v"(quit));
    {
        unsigned long long w2 = wpos + (s_pos - pub_pos);  // a period advances the window by far less than the ring holds
        if (w2 >= a.ring_capacity) w2 -= a.ring_capacity;
        v.wpos = w2;
    }
    v.pub_pos = s_pos;
    seq += 1ull;
    v.seq = seq;
    v.out_valid = 1;  // this period's record is complete in LDS (the lanes' stores lie before the barrier that ends the period, thread 0's are its own)
    const unsigned long long oldest = live_oldest(head, origin, a.ring_capacity);
    const bool resident = s_active && (s_pos + vlen <= head) && (s_pos >= oldest);
    const bool room = (seq - consumed) < static_cast<unsigned long long>(a.live.ring_len);
    const bool in_budget = (now - t_start) < a.live.residency_ticks;  // (the clock is the look-out lane's reading: s_memrealtime is a memory operation, ~a microsecond)
    w.go = (resident && room && in_budget && !quit) ? 1 : 2;
}

// The drain round: every record store of the channel has completed (the lanes that store have waited, a barrier lies in between), so the full count is
// published; then wait for the next window -- polling the ring's live words in device memory, now and then the host's quit word -- until it is resident,
// or one of the reasons to leave applies.
__device__ __forceinline__ void live_wait(const TrkArgs& a, int ch, TrkChannel& s, NextWindow& w, LiveShared& v, unsigned vlen)
{
    LiveTail* th = a.live.tail + ch;
    __hip_atomic_store(&th->pos, s.pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&th->active, s.active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&th->seq, v.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int reason = LIVE_EXIT_NONE;
    if (!s.active) reason = LIVE_EXIT_INACTIVE;
    const unsigned long long t_wait = wall_clock64();
    unsigned it = 0;
    while (reason == LIVE_EXIT_NONE)
        {
            if ((it++ & 7u) == 0u && __hip_atomic_load(a.live.quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)
                {
                    reason = LIVE_EXIT_QUIT;
                    break;
                }
            const unsigned long long head = __hip_atomic_load(a.live.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long now = wall_clock64();
            if ((now - v.t_start) >= a.live.residency_ticks)
                {
                    reason = LIVE_EXIT_BUDGET;
                    break;
                }
            if (s.pos + vlen <= head)
                {
                    // (the origin word is written before the head that goes with it: read after it, it is at least as new)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // ... and the window's samples are read afresh, not from this compute unit's L1
                    const unsigned long long origin = __hip_atomic_load(a.live.head + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (s.pos < live_oldest(head, origin, a.ring_capacity))
                        {
                            reason = LIVE_EXIT_OVERRUN;
                            break;
                        }
                    if ((v.seq - v.consumed) >= static_cast<unsigned long long>(a.live.ring_len))
                        v.consumed = __hip_atomic_load(a.live.consumed + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((v.seq - v.consumed) < static_cast<unsigned long long>(a.live.ring_len))
                        {
                            v.head = head;
                            v.origin = origin;
                            v.head_fenced = head;
                            break;  // go
                        }
                }
            if ((now - t_wait) >= a.live.idle_ticks)
                reason = LIVE_EXIT_IDLE;
            else
                __builtin_amdgcn_s_sleep(16);
        }
    v.exit_reason = reason;
    w.go = (reason == LIVE_EXIT_NONE) ? 1 : 0;
}

enum : unsigned
{
    CF_SYMBOL_SYNC = 1u << 0,
    CF_LOCK_DETECTORS = 1u << 1,
    CF_TRACK_PILOT = 1u << 2,
    CF_HAS_SECONDARY = 1u << 3,
    CF_FLL_PULL_IN = 1u << 4,
    CF_FLL_STEADY = 1u << 5,
    CF_DOPPLER_CORRECTION = 1u << 6,
    CF_CARRIER_AIDING = 1u << 7,
    CF_EXTEND_GT1 = 1u << 8,
    CF_SYMBOLS_GT1 = 1u << 9,
    CF_DATA_SECONDARY = 1u << 10,
    CF_CLOOP = 1u << 11,
};
__device__ __forceinline__ unsigned conf_switches(const gsh_trk_conf& c)
{
    return (c.enable_symbol_sync ? CF_SYMBOL_SYNC : 0u) | (c.enable_lock_detectors ? CF_LOCK_DETECTORS : 0u) | (c.track_pilot ? CF_TRACK_PILOT : 0u) |
           (c.has_secondary ? CF_HAS_SECONDARY : 0u) | (c.enable_fll_pull_in ? CF_FLL_PULL_IN : 0u) | (c.enable_fll_steady_state ? CF_FLL_STEADY : 0u) |
           (c.enable_doppler_correction ? CF_DOPPLER_CORRECTION : 0u) | (c.carrier_aiding ? CF_CARRIER_AIDING : 0u) |
           (c.extend_correlation_symbols > 1 ? CF_EXTEND_GT1 : 0u) | (c.symbols_per_bit > 1 ? CF_SYMBOLS_GT1 : 0u) |
           (c.data_secondary_code_length > 0 ? CF_DATA_SECONDARY : 0u) | (c.cloop != 0 ? CF_CLOOP : 0u);
}

// HD: Dll_Pll_Conf::high_dyn -- a compile-time switch so that the standard path does not carry the high-dynamics correlator's registers
// LIVE: the residency form of the loop (gsh_trk_live_*) -- a compile-time switch as well: the launched form keeps the code (and the registers) it had
// COOP (round 6, gsh_trk_set_split): a.coop_g work-groups on different compute units share every window of a channel -- see the block comment at `coop` below
template <int NT, bool HD, bool LIVE, bool COOP = false>
// conf: the device copy of the configuration as a parameter of its own, const and __restrict__: nothing the kernel writes aliases it, so its fields are
// fetched with scalar loads and may be hoisted -- through the pointer inside TrkArgs every c.field in thread 0's section was a vector memory load that could not
// move above the record stores before it.
__global__ __launch_bounds__(MC_THREADS) void trk_loop_kernel(TrkArgs a, const gsh_trk_conf* __restrict__ conf)
{
    extern __shared__ __align__(16) float lds[];
    __shared__ NextWindow win;
    // The channel's loop state and lock-detector / symbol-sync state live in LDS for the duration of the launch: thread 0 is the only one
    // that works on them, and its section between two barriers is a chain of dependent accesses -- LDS latency instead of device-memory
    // latency, and no thread holds a private copy (round 1: ~1 KB of scratch per thread, 1024 threads per channel).
    __shared__ __align__(16) TrkChannel s;
    __shared__ __align__(16) LockState lk;
    __shared__ SerialMail mail;  // between the lanes that share a period's loop arithmetic (below)
    constexpr bool SEEDS = GSH_TRK_SEED_TABLES && !HD;
    __shared__ __align__(16) float2 seed_tab[SEEDS ? mcdev::SEED_ENTRIES : 1];
    __shared__ __align__(16) HotConstants hc;
    __shared__ std::conditional_t<LIVE, LiveShared, NoLiveShared> lv;  // (not allocated in the launched form: nothing there touches it)
    // live form: the period's record is assembled HERE by the lanes that know its fields and written to the host's record ring by ONE wave
    // instruction (28 lanes x 8 bytes, system scope) at the top of the next period.  Field-by-field system-scope stores -- what visibility to a host that
    // reads while the kernel runs demands of stores into host memory -- are one PCIe write each: 55 per record, and 32 channels of them took 53 us per period.
    __shared__ __align__(16) std::conditional_t<LIVE, gsh_trk_epoch, NoLiveShared> lrec;
    static_assert(sizeof(gsh_trk_epoch) % 8 == 0 && sizeof(gsh_trk_epoch) / 8 <= 64, "the record is written out as 8-byte pieces by one wave");
    static_assert(sizeof(TrkChannel) % 4 == 0 && sizeof(LockState) % 4 == 0, "state is copied as 32-bit words");
    static_assert(!COOP || (!HD && !LIVE), "cooperating work-groups exist for the launched standard-mode loop");
    // COOP: block b sits on XCD b % 8; the coop_g work-groups of a channel take consecutive slots of ONE XCD (a hand-off between them is an L2 round trip: ~350 ns,
    // profiles/ubench/pingpong.hip), channel = (slot / coop_g) * 8 + xcd
    int ch = blockIdx.x, coop_rank = 0;
    if constexpr (COOP)
        {
            const int xcd = static_cast<int>(blockIdx.x & 7u), q = static_cast<int>(blockIdx.x >> 3);
            coop_rank = q % a.coop_g;
            ch = (q / a.coop_g) * 8 + xcd;
            if (ch >= static_cast<int>(a.coop_n_channels())) return;
        }
    const int tid = threadIdx.x;
    const gsh_trk_conf& c = *conf;
    // The configuration's switches in ONE word, formed once per launch: every `c.flag` in the loop arithmetic was a scalar load the lane then waited for
    // (s_load_dword - s_waitcnt - branch, a data-cache round trip each, two dozen of them on the path of every period); a bit test of a register is not.
    const unsigned cf = conf_switches(c);
    auto CF = [cf](unsigned bit) { return (cf & bit) != 0u; };
    {
        const unsigned* gs = reinterpret_cast<const unsigned*>(a.chan + ch);
        const unsigned* gl = reinterpret_cast<const unsigned*>(a.lock + ch);
        unsigned* ls = reinterpret_cast<unsigned*>(&s);
        unsigned* ll = reinterpret_cast<unsigned*>(&lk);
        for (int i = tid; i < static_cast<int>(sizeof(TrkChannel) / 4); i += MC_THREADS) ls[i] = gs[i];
        for (int i = tid; i < static_cast<int>(sizeof(LockState) / 4); i += MC_THREADS) ll[i] = gl[i];
    }
    __syncthreads();
    const int code_len = s.code_len;
    const unsigned long long acq_stamp = s.acq_stamp & 0x7fffffffffffffffull;
    // (trk.cc:1910-1917 looks at its latch in the pull-in call too, where a read pointer behind the acquisition's stamp wraps the unsigned difference: gsh_trk_pull_in_over)
    const unsigned long long pull_in_limit = (s.acq_stamp >> 63) ? 0ull : a.pull_in_limit;

    // ---- local replicas stay in LDS for the whole launch
    float* tab = lds;
    float* tab_data = lds + code_table_floats(code_len);
    float2* red = reinterpret_cast<float2*>(tab_data + (CF(CF_TRACK_PILOT) ? code_table_floats(code_len) : 0));
    if (s.active)
        {
            stage_code_table(tab, a.codes + static_cast<size_t>(ch) * 2 * a.code_stride, code_len);
            if (CF(CF_TRACK_PILOT)) stage_code_table(tab_data, a.codes + (static_cast<size_t>(ch) * 2 + 1) * a.code_stride, code_len);
        }

    // ---- tap offsets in code samples, trk.cc:632-648 / :829-840 (wide) and :2130-2148 (narrow): formed from the configuration at the top of
    // every period (a handful of scalar operations) rather than kept in registers across the whole launch
    const float sh_data[1] = {0.0f};
    constexpr int PROMPT = NT / 2;

    // ---- coop: cooperating work-groups.  A channel's period is issue-bound on its one compute unit (12 870 wave-instructions on four SIMDs) while most of the chip
    // idles at the headline channel count.  With a.coop_g > 1 the window [0, vector_length) is cut into coop_g segments: the channel's MAIN work-group (rank 0)
    // correlates segment 0, runs the loop arithmetic and, at the top of every period, puts the window's seven words into the channel's box (one relaxed agent-scope
    // 64-bit store per word, value | tag << 32: no fence, no flag -- a reader takes a word when its tag is the period's); the HELPERS (ranks 1 ..) wait for the
    // box, fill their own seed tables for rem_carr + seg_begin * phase_step, correlate their segment and leave 2 (NT + 1) tagged partial sums, which the main
    // work-group's serial waves add to their own in rank order.  Every wait is bounded (COOP_TIMEOUT_TICKS): a partner that never runs raises the launch's error
    // word, which ends the channel's loop instead of hanging the device.  The sums are those of a different order of summation than the one-work-group form's:
    // records agree with it to rounding, not bit for bit -- hence a switch (gsh_trk_set_split), not the default.
    unsigned long long* const coop_mine = COOP ? a.coop_box + static_cast<size_t>(ch) * coop_stride(a.coop_g) : nullptr;
    unsigned long long* const coop_err = COOP ? a.coop_box + static_cast<size_t>(a.coop_channels) * coop_stride(a.coop_g) : nullptr;
    int seg_begin = 0, seg_end = static_cast<int>(c.vector_length);
    if constexpr (COOP)
        {
            // Whole trips: a correlation costs ~1 us whatever its length plus ~0.6 us per trip of 2 x NCH x 1 024 samples per lane-pair row (mcorr_device.h), so
            // segments are cut at trip boundaries.  T trips over coop_g work-groups: every helper takes h = floor(T / g + 0.4) of them (at least one), the main
            // work-group the rest at the FRONT of the window -- it starts a hand-off before the helpers and their sums need another one to arrive, so the odd trip
            // is its (7 trips: 4 + 3; 1 + 2 + 2 + 2; with eight work-groups seven helpers take one each and the main one only joins).
            const int n = static_cast<int>(c.vector_length);
            const int trip = (NT <= 3 && !CF(CF_TRACK_PILOT)) ? 4 * MC_THREADS : 2 * MC_THREADS;
            const int T = (n + trip - 1) / trip;
            const int helpers = a.coop_g - 1;
            int h = helpers > 0 ? max(1, (10 * T + 4 * a.coop_g) / (10 * a.coop_g)) : 0;
            if (a.coop_helper_trips > 0) h = a.coop_helper_trips;
            const int main_trips = max(0, T - helpers * h);
            if (coop_rank == 0)
                {
                    seg_begin = 0;
                    seg_end = min(n, main_trips * trip);
                }
            else
                {
                    seg_begin = min(n, (main_trips + (coop_rank - 1) * h) * trip);
                    seg_end = (coop_rank == helpers) ? n : min(n, seg_begin + h * trip);
                    if (seg_end < seg_begin) seg_end = seg_begin;
                }
        }
    if constexpr (COOP)
        if (coop_rank > 0)
            {
                __syncthreads();  // code table(s) staged
                for (int e = 0; e < a.n_epochs; e++)
                    {
                        const unsigned tag = a.coop_seq0 + static_cast<unsigned>(e);
                        if ((tid >> 6) == 0)
                            {
                                // lanes 0 .. 6 take the window's words as soon as all of them carry this period's tag
                                const int lane = tid & 63;
                                unsigned long long v = 0ull;
                                const unsigned long long t0 = wall_clock64();
                                bool failed = false;
                                for (;;)
                                    {
                                        if (lane < 7) v = __hip_atomic_load(coop_mine + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        if (__all(lane >= 7 || static_cast<unsigned>(v >> 32) == tag)) break;
                                        if (wall_clock64() - t0 > COOP_TIMEOUT_TICKS)
                                            {
                                                failed = true;
                                                break;
                                            }
                                        __builtin_amdgcn_s_sleep(1);
                                    }
                                const unsigned lo = static_cast<unsigned>(v);
                                const unsigned w0 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(lo), 0));
                                const unsigned w1 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(lo), 1));
                                if (lane == 0)
                                    {
                                        if (failed) __hip_atomic_store(coop_err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        win.pos = static_cast<unsigned long long>(w0) | (static_cast<unsigned long long>(w1) << 32);
                                        win.phase_rate = 0.0f;
                                        win.code_rate = 0.0f;
                                        win.seed_ok = SEEDS ? 1 : 0;
                                    }
                                if (lane == 2) win.rem_carr = __uint_as_float(lo);
                                if (lane == 3) win.phase_step = __uint_as_float(lo);
                                if (lane == 4) win.rem_code = __uint_as_float(lo);
                                if (lane == 5) win.code_step = __uint_as_float(lo);
                                if (lane == 6)
                                    {
                                        win.go = failed ? 0 : static_cast<int>(lo & 1u);
                                        win.narrow = static_cast<int>((lo >> 1) & 1u);
                                    }
                            }
                        if constexpr (SEEDS)
                            {
                                // the helper's own seed table, for rem_carr + seg_begin * phase_step: its two table waves take the two words they need out of the box
                                // themselves, beside wave 0 (a fill behind wave 0's broadcast would sit on the helper's critical path; per-lane seeds cost it 0.75 us)
                                int tl = tid;
                                asm volatile("" : "+v"(tl));
                                const int wv = tl >> 6;
                                if (wv == SERIAL_WAVES + 1 || wv == SERIAL_WAVES + 2)
                                    {
                                        const int lane = tl & 63;
                                        unsigned long long v = 0ull;
                                        const unsigned long long t0 = wall_clock64();
                                        bool ok = true;
                                        for (;;)
                                            {
                                                if (lane < 7) v = __hip_atomic_load(coop_mine + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                                if (__all(lane >= 7 || static_cast<unsigned>(v >> 32) == tag)) break;
                                                if (wall_clock64() - t0 > COOP_TIMEOUT_TICKS)
                                                    {
                                                        ok = false;
                                                        break;
                                                    }
                                                __builtin_amdgcn_s_sleep(1);
                                            }
                                        const int lo = static_cast<int>(static_cast<unsigned>(v));
                                        const float b_rem = __int_as_float(__builtin_amdgcn_readlane(lo, 2)), b_step = __int_as_float(__builtin_amdgcn_readlane(lo, 3));
                                        const bool go = ok && (__builtin_amdgcn_readlane(lo, 6) & 1) != 0;
                                        if (go)
                                            mcdev::seed_table_fill(seed_tab, b_step, static_cast<double>(b_rem) + static_cast<double>(seg_begin) * static_cast<double>(b_step), lane,
                                                wv - (SERIAL_WAVES + 1));
                                    }
                            }
                        __syncthreads();
                        if (!win.go) break;  // uniform
#ifdef GSH_COOP_PROFILE
                        const unsigned long long tp_seen = wall_clock64();
#endif
                        const unsigned long long wpos = a.ring_capacity ? win.pos % a.ring_capacity : win.pos;
                        float sh[NT];
                        {
                            const float spcf = static_cast<float>(c.code_samples_per_chip);
                            const float el = (win.narrow ? c.early_late_space_narrow_chips : c.early_late_space_chips) * spcf;
                            const float vel = (win.narrow ? c.very_early_late_space_narrow_chips : c.very_early_late_space_chips) * spcf;
                            if (NT == 5)
                                {
                                    sh[0] = -vel;
                                    sh[1] = -el;
                                    sh[2] = 0.0f;
                                    sh[NT - 2] = el;
                                    sh[NT - 1] = vel;
                                }
                            else
                                {
                                    sh[0] = -el;
                                    sh[1] = 0.0f;
                                    sh[NT - 1] = el;
                                }
                        }
                        const float2* const seeds = SEEDS ? seed_tab : nullptr;
                        if (CF(CF_TRACK_PILOT))
                            correlate_window_std_aux<NT, false>(a.stream, wpos, static_cast<int>(c.vector_length), tab, tab_data, sh_data[0], code_len, sh, win.rem_carr, win.phase_step, win.rem_code,
                                win.code_step, red, mcdev::NoHook(), seeds, seg_begin, seg_end);
                        else
                            correlate_window_std<NT, false, false>(a.stream, wpos, static_cast<int>(c.vector_length), tab, code_len, sh, win.rem_carr, win.phase_step, win.rem_code, win.code_step, red,
                                mcdev::NoHook(), seeds, seg_begin, seg_end);
                        if ((tid >> 6) == 0)
                            {
                                float2 sums[NT + 1];
                                mcdev::sum_wave_partials<NT + 1>(red, sums);
                                const int lane = tid & 63;
                                float mine = 0.0f;
#pragma unroll
                                for (int t = 0; t < NT + 1; t++)
                                    {
                                        mine = (lane == 2 * t) ? sums[t].x : mine;
                                        mine = (lane == 2 * t + 1) ? sums[t].y : mine;
                                    }
                                if (lane < 2 * (NT + 1))
                                    __hip_atomic_store(coop_mine + COOP_BOX_WORDS + COOP_PART_WORDS * (coop_rank - 1) + lane,
                                        static_cast<unsigned long long>(__float_as_uint(mine)) | (static_cast<unsigned long long>(tag) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef GSH_COOP_PROFILE
                                if (ch == 0 && coop_rank == 1 && lane == 0 && e >= 8)
                                    {
                                        atomicAdd(coop_err + 1, tp_seen);                 // [1] sum of "seen" stamps
                                        atomicAdd(coop_err + 2, wall_clock64());          // [2] sum of "partial stored" stamps
                                        atomicAdd(coop_err + 3, 1ull);
                                    }
#endif
                            }
                        __syncthreads();  // the rows and win are rewritten by the next period
                    }
                return;
            }

    constexpr bool live = LIVE;
    if (tid == 0)
        {
            publish(win, s, c, a.n_stream, a.n_epochs > 0, a.ring_oldest);
            win.narrow = lk.narrow;
            hc.code_chip_rate = c.code_chip_rate;
            hc.signal_carrier_freq = c.signal_carrier_freq;
            hc.fs_in = c.fs_in;
            hc.cfo_frequency_hz = c.cfo_frequency_hz;
            hc.code_length_chips = c.code_length_chips;
            hc.vector_length = c.vector_length;
            hc.code_samples_per_chip_f = static_cast<float>(c.code_samples_per_chip);
            mail.code_seq = mail.cn0_seq = mail.carr_seq = mail.inputs_cn0 = mail.inputs_carr = 0;
            mail.step_seq = 0;
            mail.lost = mail.lost_carrier = 0;
            mail.coop_err = 0;
            mail.may_trip_code = (lk.code_lock_fail_counter + 1 > c.max_code_lock_fail) ? 1 : 0;
            mail.may_trip_carr = (lk.carrier_lock_fail_counter + 1 > c.max_carrier_lock_fail) ? 1 : 0;
            if constexpr (LIVE)
                {
                    // the channel's record count lives in the host's tail (this kernel is its only writer): residencies hand it on through there
                    lv.seq = __hip_atomic_load(&a.live.tail[ch].seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(&a.live.tail[ch].exit_reason, static_cast<int>(LIVE_EXIT_NONE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // resident (the host looks)
                    lv.exit_reason = LIVE_EXIT_NONE;
                    lv.t_start = wall_clock64();
                    lv.now = lv.t_start;
                    lv.consumed = __hip_atomic_load(a.live.consumed + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    win.go = 2;
                    lv.head = 0ull;
                    lv.origin = 0ull;
                    lv.head_fenced = 0ull;
                    lv.quit = 0;
                    lv.out_valid = 0;
                    lv.pub_pos = s.pos;
                    lv.wpos = s.pos % a.ring_capacity;
                }
            if constexpr (SEEDS) win.seed_ok = 1;  // (filled right below: EVERY period of every launch takes its seeds from the tables -- a launch's first period
                                                   // from a per-lane evaluation would make the records depend on where the launches' boundaries fall)
        }
    __syncthreads();
    if constexpr (SEEDS)
        {
            int tl = tid;
            asm volatile("" : "+v"(tl));  // (nothing of this may be kept for the loop: its registers are spoken for)
            const int wv = tl >> 6;
            if (wv == SERIAL_WAVES + 1 || wv == SERIAL_WAVES + 2) mcdev::seed_table_fill(seed_tab, win.phase_step, win.rem_carr, tl & 63, wv - (SERIAL_WAVES + 1));
            __syncthreads();
        }

    int done = 0;
    for (int e = 0; e < a.n_epochs; e++)
        {
            if constexpr (LIVE)
                {
                    // The record of the period that has just ended goes out: the first wave without loop arithmetic copies it from LDS to its slot in the host's
                    // ring, 8 bytes per lane, write-through (system scope: the store's acknowledgement means it has left the device -- a plain store is
                    // acknowledged by the L2, and the count published later could overtake it).  The drain rule: that wave waits for the acknowledgement at the end
                    // of the correlation that follows (live_hook: issued a whole correlation ago, the wait costs nothing), so at the end of period k thread 0 can
                    // vouch for the records below k (live_advance); when the channel is about to wait or to leave, the round below (the wait, then a barrier)
                    // lets it vouch for all of them.
                    if ((tid >> 6) == SERIAL_WAVES && lv.out_valid)
                        {
                            const int piece = tid & 63;
                            if (piece < static_cast<int>(sizeof(gsh_trk_epoch) / 8))
                                {
                                    const size_t slot = static_cast<size_t>(ch) * a.live.ring_len + (static_cast<unsigned>(lv.seq - 1ull) & (a.live.ring_len - 1u));
                                    const unsigned long long v = reinterpret_cast<const unsigned long long*>(&lrec)[piece];
                                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.records + slot) + piece, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                }
                            else if (piece == static_cast<int>(sizeof(gsh_trk_epoch) / 8))
                                {
                                    // ... and with it where the channel stands and how many records are COMPLETE: those below the one going out now (it was
                                    // preceded by a wait for its predecessor's stores, live_hook)
                                    LiveTail* th = a.live.tail + ch;
                                    __hip_atomic_store(&th->pos, lv.pub_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    __hip_atomic_store(&th->seq, lv.seq - 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                }
                            if (piece == 0) lv.out_valid = 0;
                        }
                    if (win.go == 2)  // uniform
                        {
                            if ((tid >> 6) == SERIAL_WAVES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __syncthreads();
                            if (tid == 0) live_wait(a, ch, s, win, lv, c.vector_length);  // (the NCO settings stand: publish() formed them; only residency was open)
                            __syncthreads();
                        }
                }
            if constexpr (COOP)
                {
                    // this period's window for the helpers (win is final behind the barrier that ended the last period): seven lanes of a wave without loop arithmetic
                    if ((tid >> 6) == SERIAL_WAVES + 3)
                        {
                            const int lane = tid & 63;
                            const unsigned long long pos = win.pos;
                            unsigned w = 0u;
                            w = lane == 0 ? static_cast<unsigned>(pos) : w;
                            w = lane == 1 ? static_cast<unsigned>(pos >> 32) : w;
                            w = lane == 2 ? __float_as_uint(win.rem_carr) : w;
                            w = lane == 3 ? __float_as_uint(win.phase_step) : w;
                            w = lane == 4 ? __float_as_uint(win.rem_code) : w;
                            w = lane == 5 ? __float_as_uint(win.code_step) : w;
                            w = lane == 6 ? ((win.go && !mail.coop_err) ? 1u : 0u) | (win.narrow ? 2u : 0u) : w;
                            if (lane < 7)
                                __hip_atomic_store(coop_mine + lane, static_cast<unsigned long long>(w) | (static_cast<unsigned long long>(a.coop_seq0 + static_cast<unsigned>(e)) << 32),
                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef GSH_COOP_PROFILE
                            if (ch == 0 && lane == 0 && e >= 8 && win.go) atomicAdd(coop_err + 4, wall_clock64());  // [4] sum of "published" stamps
#endif
                        }
                    if (mail.coop_err) break;  // uniform (written before the barrier that ended the last period)
                }
            if (!win.go) break;  // uniform: win is only rewritten between the barriers below
#ifdef GSH_TRK_PROFILE
            const long long t_begin = clock64();
#endif
            const unsigned long long pos = win.pos;
            unsigned long long wpos;  // where the window sits in memory
            if constexpr (LIVE)
                wpos = lv.wpos;
            else
                wpos = a.ring_capacity ? pos % a.ring_capacity : pos;
            // where this period's record is put together: slot e of the launch's block, or -- live -- the copy in LDS.  (The address is formed where it is
            // used, by the lanes that write it, not carried in registers through the correlation.)
            auto rec_ref = [&]() -> gsh_trk_epoch& {
                if constexpr (LIVE)
                    return lrec;
                else
                    return a.records[static_cast<size_t>(ch) * a.n_epochs + e];
            };
            const float rem_carr = win.rem_carr, phase_step = win.phase_step, rem_code = win.rem_code, code_step = win.code_step;
            const float2* const seeds = (SEEDS && GSH_TRK_SEED_TABLES != 2 && win.seed_ok) ? seed_tab : nullptr;  // uniform
            float sh[NT];
            {
                const float spcf = static_cast<float>(c.code_samples_per_chip);
                const float el = (win.narrow ? c.early_late_space_narrow_chips : c.early_late_space_chips) * spcf;
                const float vel = (win.narrow ? c.very_early_late_space_narrow_chips : c.very_early_late_space_chips) * spcf;
                if (NT == 5)
                    {
                        sh[0] = -vel;
                        sh[1] = -el;
                        sh[2] = 0.0f;
                        sh[NT - 2] = el;
                        sh[NT - 1] = vel;
                    }
                else
                    {
                        sh[0] = -el;
                        sh[1] = 0.0f;
                        sh[NT - 1] = el;
                    }
            }
            // live mode, between a wave's last trip and the barrier that ends the correlation (where the waves that finish first idle anyway):
            //   the look-out lane -- lane 0 of the first wave without loop arithmetic -- reads how far the ring is complete by now (and, every 64th period -- half a millisecond --, the host's
            //   quit word: host memory, a PCIe round trip);  its wave waits for the record it wrote out at the top of the period (the drain rule).
            auto live_hook = [&]() {
                if constexpr (LIVE)
                    {
                        if (tid == 64 * SERIAL_WAVES)
                            {
                                const unsigned long long head = __hip_atomic_load(a.live.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const unsigned long long origin = __hip_atomic_load(a.live.head + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                int quit = 0;
                                if ((static_cast<unsigned>(lv.seq) & 63u) == 63u) quit = __hip_atomic_load(a.live.quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                lv.now = wall_clock64();
                                lv.head = head;
                                lv.origin = origin;
                                lv.quit = quit;
                            }
                        if ((tid >> 6) == SERIAL_WAVES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record written out at the top of this period has left the device
                    }
            };
            const float phase_rate = win.phase_rate, code_rate = win.code_rate;
            // track_pilot in the standard mode: the data-component prompt (trk.cc:1246-1256) rides on the pilot's pass over the window
            const bool fused_data = !HD && CF(CF_TRACK_PILOT);
            float2 out[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) out[t] = make_float2(0.0f, 0.0f);
            float2 pdata = make_float2(0.0f, 0.0f);
            if constexpr (HD)  // set_high_dynamics_resampler(high_dyn), trk.cc:669-675: the high-dynamics resampler + rotator pair
                {
                    live_hook();  // (no idle stretch to hide it in: the high-dynamics correlator is an order of magnitude slower than the look-out)
                    correlate_window<NT, 1>(a.stream, wpos, static_cast<int>(c.vector_length), tab, code_len, sh, rem_carr, phase_step, phase_rate, rem_code, code_step, code_rate, red);
#pragma unroll
                    for (int t = 0; t < NT; t++) out[t] = red[t];
                    if (CF(CF_TRACK_PILOT))
                        {
                            __syncthreads();  // everyone has read red[0..NT) before it is reused
                            correlate_window<1, 1>(a.stream, wpos, static_cast<int>(c.vector_length), tab_data, code_len, sh_data, rem_carr, phase_step, phase_rate, rem_code, code_step, code_rate, red);
                            pdata = red[0];
                        }
                    __syncthreads();  // win and red have been read by everyone
                }
            else
                {
                    // Standard mode: the call returns after its first barrier with one row of partial sums per wave; only the wave that runs the loop arithmetic adds
                    // them up (sum_wave_partials: same order, same sums), the others go straight on to the barrier that ends the period.  win is rewritten and the
                    // rows are reused only after that barrier.  (Until round 3: sum by NT threads -> barrier -> every thread read the sums -> barrier.)
                    if (fused_data)
                        correlate_window_std_aux<NT, false>(a.stream, wpos, static_cast<int>(c.vector_length), tab, tab_data, sh_data[0], code_len, sh, rem_carr, phase_step, rem_code, code_step, red, live_hook, seeds,
                            seg_begin, seg_end);
#ifdef GSH_TRK_PAIRED_TAPS  // early tap read next to the late one: fewer instructions per trip, yet 0.5 us per period slower here (profiles/ab/r03/closed_loop_paired_taps.txt)
                    else if (NT == 3 && (static_cast<double>(sh[2]) - static_cast<double>(sh[0]) == 1.0) && code_step > 0.0f)  // mcorr_pair_eligible (uniform)
                        correlate_window_std<NT, true, false>(a.stream, wpos, static_cast<int>(c.vector_length), tab, code_len, sh, rem_carr, phase_step, rem_code, code_step, red);
#endif
                    else
                        correlate_window_std<NT, false, false>(a.stream, wpos, static_cast<int>(c.vector_length), tab, code_len, sh, rem_carr, phase_step, rem_code, code_step, red, live_hook, seeds,
                            seg_begin, seg_end);
                    if (tid < 64 * SERIAL_WAVES)  // the waves that hold a lane of the loop arithmetic below
                        {
                            float2 sums[NT + 1];
                            mcdev::sum_wave_partials<NT + 1>(red, sums);
#ifdef GSH_COOP_PROFILE
                            if (COOP && ch == 0 && tid == 0 && e >= 8) atomicAdd(coop_err + 5, wall_clock64());  // [5] own segment done
#endif
                            if constexpr (COOP)
                                {
                                    // the helpers' partial sums, added in rank order (every serial wave takes them itself: no barrier between the correlation and the lanes)
                                    const int lane = tid & 63;
                                    const int n_words = 2 * (NT + 1);
                                    const unsigned tag = a.coop_seq0 + static_cast<unsigned>(e);
                                    for (int r = 1; r < a.coop_g; r++)
                                        {
                                            unsigned long long v = 0ull;
                                            const unsigned long long* src = coop_mine + COOP_BOX_WORDS + COOP_PART_WORDS * (r - 1);
                                            const unsigned long long t0 = wall_clock64();
                                            bool failed = mail.coop_err != 0;
                                            while (!failed)
                                                {
                                                    if (lane < n_words) v = __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                                    if (__all(lane >= n_words || static_cast<unsigned>(v >> 32) == tag)) break;
                                                    if (wall_clock64() - t0 > COOP_TIMEOUT_TICKS) failed = true;
                                                }
                                            if (failed)
                                                {
                                                    if (lane == 0)
                                                        {
                                                            __hip_atomic_store(coop_err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                                            mail.coop_err = 1;
                                                        }
                                                    v = 0ull;
                                                }
                                            const int bits = static_cast<int>(static_cast<unsigned>(v));
#pragma unroll
                                            for (int t = 0; t < NT + 1; t++)
                                                {
                                                    sums[t].x += __int_as_float(__builtin_amdgcn_readlane(bits, 2 * t));
                                                    sums[t].y += __int_as_float(__builtin_amdgcn_readlane(bits, 2 * t + 1));
                                                }
                                        }
#ifdef GSH_COOP_PROFILE
                                    if (ch == 0 && tid == 0 && e >= 8)
                                        {
                                            atomicAdd(coop_err + 6, wall_clock64());  // [6] gathered
                                            atomicAdd(coop_err + 7, 1ull);
                                        }
#endif
                                }
#pragma unroll
                            for (int t = 0; t < NT; t++) out[t] = sums[t];
                            if (fused_data) pdata = sums[NT];
                        }
                }
#ifdef GSH_TRK_PROFILE
            const long long t_corr_done = clock64();
#endif
            // ---- the loop arithmetic between two correlations.  What does not depend on each other runs side by side on lane 0 of four different waves
            // (four SIMDs: one lane each, a chain of dependent operations) instead of one after the other on thread 0:
            //   wave 0  accumulators of states 3 / 4, carrier discriminator(s) + FLL/PLL filter              (s.pll, s.p_old_*, s.carrier_doppler_hz are its alone)
            //   wave 1  code discriminator + DLL filter                                                      (s.dll is its alone; results through `mail`)
            //   wave 2  cn0_and_tracking_lock_status, C/N0 half: M2M4 sums by the whole wave, estimate, smoother, code lock counter
            //   wave 3  ... carrier-lock half: carrier lock test, smoother, carrier lock counter; and the record's fields that are known before the join
            // (round 3 had the two halves on one lane: the period waited 2 900 clocks longer for it than for the carrier loop).
            // Each derives the few common inputs itself from state it only READS; whatever another lane reads is written by thread 0 after the barrier that
            // joins them.  The arithmetic of every value is what it was: records are bit-identical to the single-thread order (tests/test_tracking_loop_gpu.py).
            // On a loss of lock the period's loop-filter updates have happened although the reference skips them (trk.cc:2009-2014) -- nothing reads them again:
            // the channel stops, and gsh_trk_start builds its state afresh.
            const double code_period = a.code_period;  // d_code_period
            const double inv_two_pi = (a.inv_fs_in != 0.0) ? INV_TWO_PI_D : 0.0;  // (the FLL term's dividend has the correlation time in it: under the host's verdict on the configuration)
            float2 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = out[t];
            int run_state = 0;
            bool pull_in = false;
            double carr_phase_error_hz = 0.0, carr_freq_error_hz = 0.0, carr_error_filt_hz = 0.0;
            int next_symbol = 0;
            // the (wave-uniform) inputs of the lanes below, from state that is only read here
            auto form_inputs = [&]() {
                // trk.cc:1912-1915: pull-in ends once more than pull_in_time_s whole seconds have passed since acquisition
                pull_in = (pos - acq_stamp) < pull_in_limit;
                // the accumulators the loop works on (d_VE_accu .. d_VL_accu): the period's outputs in state 2 (trk.cc:1984-1991); in state 4
                // save_correlation_results adds them, times the secondary code chip, to accumulators zeroed at the end of the previous period
                run_state = CF(CF_SYMBOL_SYNC) ? lk.state : 0;
                if (run_state == 3 || run_state == 4)
                    {
                        float sgn = 1.0f;
                        next_symbol = lk.current_symbol;
                        if (CF(CF_HAS_SECONDARY))
                            {
                                sgn = c.secondary_code[next_symbol] == '0' ? 1.0f : -1.0f;
                                next_symbol = (next_symbol + 1) % c.secondary_code_length;
                            }
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            {
                                acc[t].x = __fadd_rn(lk.accv[t].x, __fmul_rn(sgn, out[t].x));  // the float += / -= of trk.cc:1493-1512
                                acc[t].y = __fadd_rn(lk.accv[t].y, __fmul_rn(sgn, out[t].y));  // (lk.accv itself is updated after the join)
                            }
                    }
            };
            // ---- C/N0 wave, all 64 lanes: the M2M4 sums over the prompt buffer as it will be once this period's prompt is in (m2m4_sums_wave)
            int cn0_cnt = 0, cn0_slot = 0;
            float cn0_psig = 0.0f, cn0_m2 = 0.0f, cn0_m4 = 0.0f;
            auto cn0_sums = [&]() {
                if (CF(CF_LOCK_DETECTORS) && run_state != 3)
                    {
                        const int ns = c.cn0_samples;
                        cn0_cnt = lk.cn0_estimation_counter;
                        cn0_slot = lk.cn0_slot;
                        if (cn0_cnt >= ns)  // uniform
                            {
                                const int lane = tid & 63;
                                float2 bp = make_float2(0.0f, 0.0f);
                                if (lane < ns) bp = *reinterpret_cast<const float2*>(&lk.prompt_buffer[2 * lane]);
                                if (lane == cn0_slot) bp = acc[PROMPT];
                                float xa = fabsf(bp.x);
                                float xb = __fadd_rn(__fmul_rn(bp.y, bp.y), __fmul_rn(bp.x, bp.x));
                                float xc = __fmul_rn(xb, xb);
                                m2m4_sums_wave(xa, xb, xc, ns);
                                cn0_psig = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xa), ns - 1));
                                cn0_m2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xb), ns - 1));
                                cn0_m4 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xc), ns - 1));
                            }
                    }
            };
            const int seq = e + 1;  // the period's number inside the launch: what the lanes say when they are done
            const bool flag_join = !HD && CF(CF_LOCK_DETECTORS);  // how the lanes meet (below); uniform over the work-group
            auto say = [&](int& word) {
                if constexpr (!HD) lane_says(word, seq);  // (the high-dynamics flavours join at a barrier, below)
            };
#if GSH_TRK_PREFIX_ALL
            if (tid < 64 * SERIAL_WAVES)
                {
                    form_inputs();
                    if (tid == 64 * CN0_WAVE) say(mail.inputs_cn0);
                    if (tid == 64 * CARR_LOCK_WAVE) say(mail.inputs_carr);
                    if ((tid >> 6) == CN0_WAVE) cn0_sums();
                }
#else
            if ((tid >> 6) == CN0_WAVE && CF(CF_LOCK_DETECTORS))
                {
                    form_inputs();
                    cn0_sums();
                }
            else if ((tid & 63) == 0 && tid < 64 * SERIAL_WAVES)
                form_inputs();
            if (tid == 64 * CN0_WAVE) say(mail.inputs_cn0);
            if (tid == 64 * CARR_LOCK_WAVE) say(mail.inputs_carr);
#endif
            if ((tid & 63) == 0 && tid < 64 * SERIAL_WAVES)
                {
                    const int extend = (CF(CF_SYMBOL_SYNC) && CF(CF_EXTEND_GT1)) ? c.extend_correlation_symbols : 1;
                    const float2 P = acc[PROMPT], E = acc[PROMPT - 1], L = acc[PROMPT + 1];
                    if (tid == 0)
                        {
                            // the carrier filter's state and the previous prompt come out of LDS in one go, up front (pinned by the asm: see the note at join_and_update)
                            FllPllState pl = s.pll;
                            float p_old_re = s.p_old_re, p_old_im = s.p_old_im;
                            if constexpr (!HD)  // (the high-dynamics flavours sit at the register limit: thirteen more live values there are scratch)
                                asm volatile("" : "+v"(pl.w), "+v"(pl.x), "+v"(pl.w0p), "+v"(pl.w0p2), "+v"(pl.w0p3), "+v"(pl.w0f), "+v"(pl.w0f2), "+v"(pl.a2), "+v"(pl.a3), "+v"(pl.b3),
                                             "+v"(pl.order), "+v"(p_old_re), "+v"(p_old_im));
                            if (run_state == 3 || run_state == 4)
                                {
                                    const float2 pd = CF(CF_TRACK_PILOT) ? pdata : out[PROMPT];
                                    if (CF(CF_SYMBOLS_GT1))
                                        {
                                            float ds = 1.0f;
                                            if (CF(CF_DATA_SECONDARY))
                                                {
                                                    ds = c.data_secondary_code[lk.current_data_symbol] == '0' ? 1.0f : -1.0f;
                                                    lk.current_data_symbol = (lk.current_data_symbol + 1) % c.data_secondary_code_length;
                                                }
                                            else
                                                {
                                                    lk.current_data_symbol = (lk.current_data_symbol + 1) % c.symbols_per_bit;
                                                }
                                            lk.p_data_accu[0] = __fadd_rn(lk.p_data_accu[0], __fmul_rn(ds, pd.x));
                                            lk.p_data_accu[1] = __fadd_rn(lk.p_data_accu[1], __fmul_rn(ds, pd.y));
                                        }
                                    else
                                        {
                                            lk.p_data_accu[0] = pd.x;
                                            lk.p_data_accu[1] = pd.y;
                                        }
                                    lk.cloop = CF(CF_TRACK_PILOT) ? 0 : 1;  // trk.cc:1587-1595
                                }
                            // ---- run_dll_pll, carrier half, trk.cc:1260-1303 (skipped during coherent integration, state 3: trk.cc:2156-2161)
                            if (run_state != 3)
                                {
                                    const bool cloop_now = CF(CF_SYMBOL_SYNC) ? (lk.cloop != 0) : CF(CF_CLOOP);
                                    const double corr_time = (CF(CF_SYMBOL_SYNC) && lk.corr_time > 0.0) ? lk.corr_time : code_period;  // d_current_correlation_time_s
                                    carr_phase_error_hz = div_by_constant_if<GSH_TRK_FAST_DIV != 0>(cloop_now ? pll_cloop_two_quadrant_atan_d(P) : pll_four_quadrant_atan_d(P), GNSS_TWO_PI_D, INV_TWO_PI_D);  // (a float arctangent: zero, or no smaller than 1e-45)
                                    float carr_error_filt;
                                    if ((pull_in && CF(CF_FLL_PULL_IN)) || CF(CF_FLL_STEADY))
                                        {
                                            carr_freq_error_hz = div_by_constant(fll_diff_atan_d(make_float2(p_old_re, p_old_im), P, 0.0, corr_time), GNSS_TWO_PI_D, inv_two_pi);
                                            s.p_old_re = P.x;
                                            s.p_old_im = P.y;
                                            if (pull_in && CF(CF_FLL_PULL_IN))
                                                carr_error_filt = fll_pll_carrier_error(pl, static_cast<float>(carr_freq_error_hz), 0.0f, static_cast<float>(corr_time));
                                            else
                                                carr_error_filt = fll_pll_carrier_error(pl, static_cast<float>(carr_freq_error_hz), static_cast<float>(carr_phase_error_hz), static_cast<float>(corr_time));
                                        }
                                    else
                                        {
                                            carr_error_filt = fll_pll_carrier_error(pl, 0.0f, static_cast<float>(carr_phase_error_hz), static_cast<float>(corr_time));
                                        }
#ifdef GSH_TRK_PROFILE
                                    mail.t_lane[0] = clock64() - t_corr_done;
#endif
                                    s.pll.w = pl.w;  // (the filter's memories; its coefficients are not written)
                                    s.pll.x = pl.x;
                                    carr_error_filt_hz = carr_error_filt;
                                    s.carrier_doppler_hz = carr_error_filt_hz;
                                }
                        }
                    else if (tid == 64)
                        {
                            // ---- run_dll_pll, code half, trk.cc:1305-1316
                            double code_error_chips = 0.0, code_error_filt_chips = 0.0;
                            if (run_state != 3)
                                {
                                    const float spc_now = (CF(CF_SYMBOL_SYNC) && lk.narrow) ? lk.spc_now : c.spc;
                                    if (NT == 5)
                                        code_error_chips = dll_nc_vemlp_normalized_d(acc[0], acc[1], acc[NT - 2], acc[NT - 1]);
                                    else
                                        code_error_chips = dll_nc_e_minus_l_normalized_d(E, L, spc_now, c.slope, c.y_intercept);
                                    code_error_filt_chips = loop_filter_apply<!HD>(s.dll, static_cast<float>(code_error_chips));
                                }
                            mail.code_error_chips = code_error_chips;
                            mail.code_error_filt_chips = code_error_filt_chips;
#ifdef GSH_TRK_PROFILE
                            mail.t_lane[1] = clock64() - t_corr_done;
#endif
                            say(mail.code_seq);
                            if (a.records != nullptr)
                                {
                                    gsh_trk_epoch& r = rec_ref();
                                    rec_set<LIVE>(r.code_error_chips, code_error_chips);
                                    rec_set<LIVE>(r.code_error_filt_chips, code_error_filt_chips);
                                }
                        }
                    else
                      {
                        if (tid == 64 * CN0_WAVE)
                        {
                            // ---- cn0_and_tracking_lock_status, C/N0 half (cn0_half_d)
                            bool lost_now = false;
                            if (CF(CF_LOCK_DETECTORS))
                                {
                                    if (lk.pull_in_latched && !pull_in)  // trk.cc:1912-1916
                                        {
                                            lk.pull_in_latched = 0;
                                            lk.code_lock_fail_counter = 0;
                                        }
                                    if (run_state != 3)  // coherent integration runs no lock test (trk.cc:2156-2161)
                                        lost_now = cn0_half_d(lk, c, P, cn0_cnt, cn0_slot, cn0_psig, cn0_m2, cn0_m4,
                                            run_state == 4 ? code_period * static_cast<double>(extend) : code_period, pull_in);  // trk.cc:2008, :2203
                                }
                            mail.lost = lost_now ? 1 : 0;
                            mail.may_trip_code = (CF(CF_LOCK_DETECTORS) && lk.code_lock_fail_counter + 1 > c.max_code_lock_fail) ? 1 : 0;  // (next period: the counter moves by one)
                            if (a.records != nullptr) rec_set<LIVE>(rec_ref().cn0_db_hz, CF(CF_LOCK_DETECTORS) ? lk.cn0_db_hz : 0.0f);
#ifdef GSH_TRK_PROFILE
                            mail.t_lane[2] = clock64() - t_corr_done;
#endif
                            say(mail.cn0_seq);
                        }
                        if (tid == 64 * CARR_LOCK_WAVE)
                        {
                            // ---- carrier-lock half (carrier_lock_half_d).
                            // This lane also writes the part of the period's record that is known before the join (thread 0 adds the loop's outputs after it, or, on a
                            // loss of lock, clears what does not belong into that record): the accumulators the loop works on -- what log_data dumps as
                            // |d_VE_accu| .. |d_VL_accu| (trk.cc:1624-1636) --, the correlator outputs, the window's position, state and flags
                            gsh_trk_epoch* const rp = a.records != nullptr ? &rec_ref() : nullptr;
                            if (rp != nullptr)
                                {
#pragma unroll
                                    for (int t = 0; t < 5; t++)  // (all ten slots: the device buffer is not cleared between launches)
                                        {
                                            rec_set<LIVE>(rp->accu[2 * t], (t < NT) ? acc[t < NT ? t : 0].x : 0.0f);
                                            rec_set<LIVE>(rp->accu[2 * t + 1], (t < NT) ? acc[t < NT ? t : 0].y : 0.0f);
                                            rec_set<LIVE>(rp->corr[2 * t], (t < NT) ? out[t < NT ? t : 0].x : 0.0f);
                                            rec_set<LIVE>(rp->corr[2 * t + 1], (t < NT) ? out[t < NT ? t : 0].y : 0.0f);
                                        }
                                    rec_set<LIVE>(rp->prompt_data[0], pdata.x);
                                    rec_set<LIVE>(rp->prompt_data[1], pdata.y);
                                    rec_set<LIVE>(rp->sample_counter, pos);
                                    rec_set<LIVE>(rp->flags, pull_in ? 1 : 0);
                                    rec_set<LIVE>(rp->state, run_state);
                                }
                            bool lost_now = false;
                            if (CF(CF_LOCK_DETECTORS))
                                {
                                    if (lk.pull_in_latched_carr && !pull_in)  // trk.cc:1912-1916
                                        {
                                            lk.pull_in_latched_carr = 0;
                                            lk.carrier_lock_fail_counter = 0;
                                        }
                                    // trk.cc:2000-2007, state 2 only: no secondary-code / bit synchronisation within the time limit forces the loss-of-lock condition
                                    if (run_state == 2 && (pos - acq_stamp) >= a.bit_sync_limit) lk.carrier_lock_fail_counter = 300000;
                                    if (run_state != 3) lost_now = carrier_lock_half_d(lk, c, P, pull_in);
                                }
                            mail.lost_carrier = lost_now ? 1 : 0;
                            mail.may_trip_carr = (CF(CF_LOCK_DETECTORS) && lk.carrier_lock_fail_counter + 1 > c.max_carrier_lock_fail) ? 1 : 0;
                            if (rp != nullptr) rec_set<LIVE>(rp->carrier_lock_test, CF(CF_LOCK_DETECTORS) ? lk.carrier_lock_test : 0.0);
                            // (thread 0 rewrites some of this lane's record fields when the channel loses lock: the stores above have to have landed by then)
                            // (only where the lanes meet through the words: in front of a barrier the wait would hold everybody up for the stores' round trip)
                            if constexpr (!LIVE)
                                if (flag_join) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GSH_TRK_PROFILE
                            mail.t_lane[3] = clock64() - t_corr_done;
#endif
                            say(mail.carr_seq);
                        }
                      }
                }
            // No barrier here when the lock detectors run: see SerialMail.  Without them there is nothing to overlap and the barrier is the cheaper meeting (thread 0's
            // look at the words costs an LDS round trip and a few comparisons, ~200 clocks: 8.05 against 7.93 us per period); the high-dynamics flavours, at the
            // register limit, keep the barrier as well.
            if (!flag_join) __syncthreads();
            if constexpr (LIVE)
              if (tid == 64 * SERIAL_WAVES && lv.head != lv.head_fenced)
                {
                    // samples pushed since this compute unit last looked: whatever its L1 holds of those ring positions is a lap old.  One lane's agent-scope
                    // acquire invalidates the unit's L1; it runs beside thread 0's join and is over before the barrier that ends the period.
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    lv.head_fenced = lv.head;
                }
            if (tid == 0)
                {
#ifdef GSH_TRK_EXP_EARLY_SAY  /* (timing experiment only: the table waves start on last period's values -- wrong seeds, no waiting) */
                    if constexpr (SEEDS) lane_says(mail.step_seq, seq);
#endif
                    bool lost = false;
                    if (!flag_join)
                        lost = (mail.lost | mail.lost_carrier) != 0;
                    else
                        {
                            asm volatile("" ::: "memory");
                            int may_trip = 0;
                            while (!join_ready(mail, seq, may_trip))  // (the `inputs` words long since: they are the first thing their lanes do)
                                ;
                            asm volatile("" ::: "memory");
                            // can a fail counter pass its limit in this period?  Only from within one of its limit (the lanes said so last period -- or have said so for
                            // the next one already, having finished: then their verdict is in, and a limit passed now leaves the counter within one of it), or when the
                            // bit synchronisation time limit sets the carrier counter to 300000 (trk.cc:2000-2007) -- then the verdict is waited for
                            if (CF(CF_LOCK_DETECTORS) && (may_trip != 0 || (run_state == 2 && (pos - acq_stamp) >= a.bit_sync_limit)))
                                {
                                    lane_waits(mail.cn0_seq, seq);
                                    lane_waits(mail.carr_seq, seq);
                                    lost = (mail.lost | mail.lost_carrier) != 0;
                                }
                        }
#ifdef GSH_TRK_PROFILE
                    const long long t_join = clock64();
#endif
                    const int extend = (CF(CF_SYMBOL_SYNC) && CF(CF_EXTEND_GT1)) ? c.extend_correlation_symbols : 1;
                    if (run_state == 3 || run_state == 4)  // what the other lanes read above
                        {
                            lk.current_symbol = next_symbol;
#pragma unroll
                            for (int t = 0; t < NT; t++) lk.accv[t] = acc[t];
                        }
                    if (lost)  // trk.cc:2009-2014: clear_tracking_vars, d_state = 0 -- the channel stops here
                        {
                            float rec_cn0 = 0.0f;
                            double rec_lock_test = 0.0;
                            if (CF(CF_LOCK_DETECTORS))
                                {
                                    rec_cn0 = lk.cn0_db_hz;
                                    rec_lock_test = lk.carrier_lock_test;
                                }
                            lk.carrier_lock_fail_counter = 0;  // trk.cc:1205-1206
                            lk.code_lock_fail_counter = 0;
                            if (a.records != nullptr)
                                {
                                    gsh_trk_epoch& r = rec_ref();  // written in place, field by field
                                    {
                                        unsigned* rw = reinterpret_cast<unsigned*>(&r);
                                        for (int i = 0; i < static_cast<int>(offsetof(gsh_trk_epoch, accu) / 4); i++) rec_set<LIVE>(rw[i], 0u);  // (accu, the record's tail, is already in place)
                                    }
                                    rec_set<LIVE>(r.sample_counter, pos);
                                    rec_set<LIVE>(r.flags, (pull_in ? 1 : 0) | 2);
#pragma unroll
                                    for (int t = 0; t < 5; t++)
                                        {
                                            rec_set<LIVE>(r.corr[2 * t], (t < NT) ? out[t < NT ? t : 0].x : 0.0f);
                                            rec_set<LIVE>(r.corr[2 * t + 1], (t < NT) ? out[t < NT ? t : 0].y : 0.0f);
                                        }
                                    rec_set<LIVE>(r.prompt_data[0], pdata.x);
                                    rec_set<LIVE>(r.prompt_data[1], pdata.y);
                                    rec_set<LIVE>(r.cn0_db_hz, rec_cn0);
                                    rec_set<LIVE>(r.carrier_lock_test, rec_lock_test);
                                    rec_set<LIVE>(r.state, run_state);
                                }
                            s.active = 0;
                            publish(win, s, c, a.n_stream, 0);
                            if constexpr (SEEDS) lane_says(mail.step_seq, seq);  // (the table waves must not wait for ever; win.seed_ok is 0)
                            if constexpr (LIVE) live_advance(a, s.pos, 0, win, lv, c.vector_length);  // (the channel is stopped: the drain round publishes this record and leaves)
                        }
                    else
                        {
                    // The join and update_tracking_vars hold the loop's divisions by launch constants.  Whether the exact short form applies (div_by_constant,
                    // exact_division.h) is the host's verdict on the configuration -- ONE wave-uniform branch around the whole stretch, compile-time inside: a test at
                    // every division would cut the lane's instruction stream into pieces the scheduler cannot interleave (8.06 instead of 7.96 us per period).
                    int prn_len = 0;
#ifdef GSH_TRK_PROFILE
                    long long t_c = 0;
#endif
                    // Thread 0 works on REGISTER copies of the channel state from here to the end of the period: what it needs comes out of LDS in one go at the top
                    // (the asm below pins the reads there -- the compiler, scheduling for few live values, had put every read in front of its first use: a dozen
                    // LDS round trips on the one lane the work-group waits for), what it changes goes back in one go, and the record and the next window are formed
                    // from the registers (they used to re-read what had just been written).
                    double st_code_freq = s.code_freq_chips, st_rem_code_samples = s.rem_code_phase_samples, st_acc_phase = s.acc_carrier_phase_rad;
                    double st_doppler = s.carrier_doppler_hz, st_phase_rate = s.carrier_phase_rate_step_rad, st_code_rate = s.code_phase_rate_step_chips;
                    float st_rem_carr = s.rem_carr_phase_rad;
                    double code_error_filt_chips = mail.code_error_filt_chips;
                    // (high-dynamics flavours: straight from the configuration, where they are used -- values the compiler may fetch again cost no register)
                    double k_chip_rate = HD ? c.code_chip_rate : hc.code_chip_rate, k_carrier_freq = HD ? c.signal_carrier_freq : hc.signal_carrier_freq;
                    double k_fs_in = HD ? c.fs_in : hc.fs_in, k_cfo = HD ? c.cfo_frequency_hz : hc.cfo_frequency_hz;
                    unsigned k_code_length = HD ? c.code_length_chips : hc.code_length_chips, k_vlen = HD ? c.vector_length : hc.vector_length;
                    float k_spcf = HD ? static_cast<float>(c.code_samples_per_chip) : hc.code_samples_per_chip_f;
                    int st_narrow = HD ? 0 : lk.narrow;
                    if constexpr (!HD)  // (the high-dynamics flavours sit at the register limit: pinned there, the values cost scratch)
                        asm volatile("" : "+v"(st_code_freq), "+v"(st_rem_code_samples), "+v"(st_acc_phase), "+v"(st_doppler), "+v"(st_phase_rate), "+v"(st_code_rate), "+v"(st_rem_carr),
                                     "+v"(code_error_filt_chips), "+v"(k_chip_rate), "+v"(k_carrier_freq), "+v"(k_fs_in), "+v"(k_cfo), "+v"(k_code_length), "+v"(k_vlen), "+v"(k_spcf),
                                     "+v"(st_narrow));
                    double st_phase_step = 0.0, st_code_step = 0.0, st_rem_code_chips = 0.0;
                    auto join_and_update = [&](auto fast_tag) {
                    constexpr bool FAST = decltype(fast_tag)::value;
                    // ---- run_dll_pll, the join: trk.cc:1317-1324
                    if (run_state != 3)
                        {
                            st_code_freq = k_chip_rate - code_error_filt_chips;
                            if (CF(CF_CARRIER_AIDING)) st_code_freq += div_by_constant_if<FAST>(st_doppler * k_chip_rate, k_carrier_freq, a.inv_signal_carrier_freq);
                            if (CF(CF_DOPPLER_CORRECTION) && !pull_in && !lk.corrected_doppler)  // trk.cc:1326-1346
                                {
                                    lk.dll_filt_sum += static_cast<double>(static_cast<float>(code_error_filt_chips));
                                    lk.dll_filt_count++;
                                    if (lk.dll_filt_count == 1000)
                                        {
                                            const float avg_code_error_chips_s = __fdiv_rn(static_cast<float>(lk.dll_filt_sum), 1000.0f);
                                            if (fabs(static_cast<double>(avg_code_error_chips_s)) > 1.0)
                                                {
                                                    const float carrier_doppler_error_hz =
                                                        __fdiv_rn(__fmul_rn(static_cast<float>(c.signal_carrier_freq), avg_code_error_chips_s), static_cast<float>(c.code_chip_rate));
                                                    const float f0 = __fsub_rn(static_cast<float>(st_doppler), carrier_doppler_error_hz);
                                                    if (s.pll.order == 3)  // Tracking_FLL_PLL_filter::initialize, T/tracking_FLL_PLL_filter.cc:57-69
                                                        {
                                                            s.pll.x = __fmul_rn(2.0f, f0);
                                                            s.pll.w = 0.0f;
                                                        }
                                                    else
                                                        {
                                                            s.pll.w = f0;
                                                            s.pll.x = 0.0f;
                                                        }
                                                    lk.corrected_doppler = 1;
                                                }
                                            lk.dll_filt_sum = 0.0;
                                            lk.dll_filt_count = 0;
                                        }
                                }
                        }

#ifdef GSH_TRK_PROFILE
                    t_c = clock64();
#endif
                    // ---- update_tracking_vars, trk.cc:1409-1483 (rate terms are zero outside high_dyn)
                    const double t_chip = 1.0 / st_code_freq;
                    const double t_prn = t_chip * static_cast<double>(k_code_length);
                    const double t_prn_samples = t_prn * k_fs_in;
                    const double k_blk = t_prn_samples + st_rem_code_samples;
                    prn_len = static_cast<int>(floor(k_blk));
                    st_phase_step = div_by_constant_if<FAST>(GNSS_TWO_PI_D * (st_doppler + k_cfo), k_fs_in, a.inv_fs_in);
                    if (HD)  // trk.cc:1425-1443
                        {
                            const int SL = static_cast<int>(c.smoother_length), cap = 2 * SL;
                            lk.carr_hist[lk.carr_pushes % cap][0] = st_phase_step;
                            lk.carr_hist[lk.carr_pushes % cap][1] = static_cast<double>(prn_len);
                            lk.carr_pushes++;
                            if (lk.carr_hist_n < cap) lk.carr_hist_n++;
                            if (lk.carr_hist_n == cap)
                                {
                                    const long long oldest = lk.carr_pushes % cap;
                                    double cp1 = 0.0, cp2 = 0.0, smp = 0.0;
                                    for (int k = 0; k < SL; k++)
                                        {
                                            cp1 += lk.carr_hist[(oldest + k) % cap][0];
                                            cp2 += lk.carr_hist[(oldest + cap - k - 1) % cap][0];
                                            smp += lk.carr_hist[(oldest + cap - k - 1) % cap][1];
                                        }
                                    cp1 /= static_cast<double>(SL);
                                    cp2 /= static_cast<double>(SL);
                                    st_phase_rate = (smp != 0.0) ? (cp2 - cp1) / smp : 0.0;
                                }
                        }
                    const double dphi = st_phase_step * static_cast<double>(prn_len) +
                                        0.5 * st_phase_rate * static_cast<double>(prn_len) * static_cast<double>(prn_len);
                    st_rem_carr += static_cast<float>(dphi);
                    st_rem_carr = static_cast<float>(fmod_two_pi<FAST>(static_cast<double>(st_rem_carr)));
                    st_acc_phase -= dphi;
                    st_code_step = div_by_constant_if<FAST>(st_code_freq, k_fs_in, a.inv_fs_in);
                    if (HD)  // trk.cc:1458-1480
                        {
                            const int SL = static_cast<int>(c.smoother_length), cap = 2 * SL;
                            lk.code_hist[lk.code_pushes % cap][0] = st_code_step;
                            lk.code_hist[lk.code_pushes % cap][1] = static_cast<double>(prn_len);
                            lk.code_pushes++;
                            if (lk.code_hist_n < cap) lk.code_hist_n++;
                            if (lk.code_hist_n == cap)
                                {
                                    const long long oldest = lk.code_pushes % cap;
                                    double cp1 = 0.0, cp2 = 0.0, smp = 0.0;
                                    for (int k = 0; k < SL; k++)
                                        {
                                            cp1 += lk.code_hist[(oldest + k) % cap][0];
                                            cp2 += lk.code_hist[(oldest + cap - k - 1) % cap][0];
                                            smp += lk.code_hist[(oldest + cap - k - 1) % cap][1];
                                        }
                                    cp1 /= static_cast<double>(SL);
                                    cp2 /= static_cast<double>(SL);
                                    if (smp >= 1.0) st_code_rate = (cp2 - cp1) / smp;
                                }
                        }
                    st_rem_code_samples = k_blk - static_cast<double>(prn_len);
                    st_rem_code_chips = div_by_constant_if<FAST>(st_code_freq * st_rem_code_samples, k_fs_in, a.inv_fs_in);
                    };
                    if (a.inv_fs_in != 0.0)
                        join_and_update(std::true_type{});
                    else
                        join_and_update(std::false_type{});
                    if constexpr (SEEDS)
                        {
                            mail.seed_step = static_cast<float>(st_phase_step);
                            mail.seed_rem = st_rem_carr;
                            lane_says(mail.step_seq, seq);
                        }

#ifdef GSH_TRK_PROFILE
                    const long long t_d = clock64();
#endif
                    // ---- symbol synchronisation (state 2, trk.cc:2026-2112) / symbol output (state 4, :2205-2246)
                    int rec_symbol_flags = 0;
                    float rec_pdata[2] = {0.0f, 0.0f};
                    if (CF(CF_SYMBOL_SYNC) && run_state == 3)
                        {
                            // trk.cc:2162-2194: a telemetry symbol may complete inside the coherent integration; then count the period
                            rec_pdata[0] = lk.p_data_accu[0];
                            rec_pdata[1] = lk.p_data_accu[1];
                            if (lk.current_data_symbol == 0)
                                {
                                    rec_symbol_flags |= 1;
                                    lk.p_data_accu[0] = lk.p_data_accu[1] = 0.0f;
                                }
                            if (lk.flag_pll_180) rec_symbol_flags |= 2;
                            lk.ext_count++;
                            if (lk.ext_count == extend - 1)
                                {
                                    lk.ext_count = 0;
                                    lk.state = 4;
                                }
                        }
                    else if (CF(CF_SYMBOL_SYNC))
                        {
                            if (run_state == 2)
                                {
                                    bool next_state = false;
                                    if (!pull_in)
                                        {
                                            if (!CF(CF_HAS_SECONDARY) && CF(CF_SYMBOLS_GT1) && lk.use_hist)  // trk.cc:2046-2072
                                                {
                                                    const bool lock_event = bit_sync_update_d(lk, c, out[PROMPT], true);
                                                    if (lock_event)
                                                        {
                                                            lk.wait_for_bit_edge = 1;
                                                            const long long k_now = lk.bs_epoch_count - 1;
                                                            const int B = c.symbols_per_bit;
                                                            // epochs_until_next_edge() - 1 (T/bit_synchronizer.cc:164-182)
                                                            int wait = ((lk.bs_edge_phase - static_cast<int>(k_now % B) + B) % B) - 1;
                                                            if (wait < 0) wait = wait + B;
                                                            lk.bs_target_epoch = k_now + wait;
                                                        }
                                                    if (lk.wait_for_bit_edge)
                                                        {
                                                            const long long k_now = lk.bs_epoch_count - 1;
                                                            if (k_now == lk.bs_target_epoch)
                                                                {
                                                                    next_state = true;
                                                                    lk.wait_for_bit_edge = 0;
                                                                    lk.use_hist = 0;  // disabled after its first lock (trk.cc:2067)
                                                                }
                                                        }
                                                }
                                            if (!next_state && (CF(CF_HAS_SECONDARY) || CF(CF_SYMBOLS_GT1)))
                                                {
                                                    const int len = c.secondary_code_length;
                                                    if (lk.ring_count < len)  // d_Prompt_circular_buffer.push_back(*d_Prompt)
                                                        {
                                                            lk.ring[2 * lk.ring_count] = out[PROMPT].x;
                                                            lk.ring[2 * lk.ring_count + 1] = out[PROMPT].y;
                                                            lk.ring_count++;
                                                        }
                                                    else if (len > 0)
                                                        {
                                                            lk.ring[2 * lk.ring_head] = out[PROMPT].x;
                                                            lk.ring[2 * lk.ring_head + 1] = out[PROMPT].y;
                                                            lk.ring_head = (lk.ring_head + 1) % len;
                                                        }
                                                    if (len > 0 && lk.ring_count == len)
                                                        {
                                                            int corr_value = 0;  // acquire_secondary, trk.cc:1118-1160
                                                            int idx = lk.ring_head;
                                                            for (int i = 0; i < len; i++)
                                                                {
                                                                    const bool zero = c.secondary_code[i] == '0';
                                                                    if (lk.ring[2 * idx] < 0.0f)
                                                                        corr_value += zero ? 1 : -1;
                                                                    else
                                                                        corr_value += zero ? -1 : 1;
                                                                    idx = (idx + 1 == len) ? 0 : idx + 1;
                                                                }
                                                            if (abs(corr_value) == len)
                                                                {
                                                                    lk.flag_pll_180 = corr_value < 0 ? 1 : 0;
                                                                    next_state = true;
                                                                }
                                                        }
                                                }
                                            if (!CF(CF_HAS_SECONDARY) && !CF(CF_SYMBOLS_GT1)) next_state = true;  // trk.cc:2091-2094
                                        }
                                    if (next_state)  // trk.cc:2101-2112, 2151-2154 (no extended integration)
                                        {
                                            lk.p_data_accu[0] = lk.p_data_accu[1] = 0.0f;
                                            lk.ring_count = lk.ring_head = 0;
                                            lk.current_symbol = 0;
                                            lk.current_data_symbol = 0;
#pragma unroll
                                            for (int t = 0; t < 5; t++) lk.accv[t] = fresh_zero2();
                                            if (extend > 1)  // trk.cc:2114-2149: stretch the integration, narrow the loops and the correlator spacing
                                                {
                                                    lk.ext_count = 0;
                                                    lk.corr_time = static_cast<double>(__fmul_rn(static_cast<float>(extend), static_cast<float>(code_period)));
                                                    lk.state = 3;
#pragma unroll
                                                    for (int k = 0; k < 4; k++)  // update_coefficients keeps the filter memories
                                                        {
                                                            s.dll.in_c[k] = lk.dll_narrow_in_c[k];
                                                            s.dll.out_c[k] = lk.dll_narrow_out_c[k];
                                                        }
                                                    s.dll.n_in = lk.dll_narrow_n_in;
                                                    s.dll.n_out = lk.dll_narrow_n_out;
                                                    const float keep_w = s.pll.w, keep_x = s.pll.x;  // set_params keeps d_pll_w / d_pll_x
                                                    s.pll = lk.pll_narrow;
                                                    s.pll.w = keep_w;
                                                    s.pll.x = keep_x;
                                                    lk.narrow = 1;
                                                    st_narrow = 1;
                                                    lk.spc_now = c.early_late_space_narrow_chips;
                                                }
                                            else
                                                {
                                                    lk.state = 4;
                                                }
                                        }
                                }
                            else
                                {
                                    if (!lk.acc_phase_initialized)  // check_carrier_phase_coherent_initialization, trk.cc:1350-1357
                                        {
                                            st_acc_phase = -static_cast<double>(st_rem_carr);
                                            lk.acc_phase_initialized = 1;
                                        }
                                    rec_pdata[0] = lk.p_data_accu[0];
                                    rec_pdata[1] = lk.p_data_accu[1];
                                    if (lk.current_data_symbol == 0)
                                        {
                                            rec_symbol_flags |= 1;
                                            lk.p_data_accu[0] = lk.p_data_accu[1] = 0.0f;
                                        }
#pragma unroll
                                    for (int t = 0; t < 5; t++) lk.accv[t] = fresh_zero2();  // trk.cc:2241-2246
                                    if (extend > 1) lk.state = 3;                                     // trk.cc:2247-2250
                                }
                            if (lk.flag_pll_180) rec_symbol_flags |= 2;
                        }

                    if (a.records != nullptr)
                        {
                            gsh_trk_epoch& r = rec_ref();  // written in place (every field is assigned)
                            rec_set<LIVE>(r.carrier_phase_rate_step_rad, st_phase_rate);
                            rec_set<LIVE>(r.code_phase_rate_step_chips, st_code_rate);
                            rec_set<LIVE>(r.symbol_flags, rec_symbol_flags);
                            rec_set<LIVE>(r.p_data_accu[0], rec_pdata[0]);
                            rec_set<LIVE>(r.p_data_accu[1], rec_pdata[1]);
                            rec_set<LIVE>(r.prn_length_samples, prn_len);
                            rec_set<LIVE>(r.rem_carr_phase_rad, st_rem_carr);
                            rec_set<LIVE>(r.carrier_doppler_hz, st_doppler);
                            rec_set<LIVE>(r.code_freq_chips, st_code_freq);
                            rec_set<LIVE>(r.carr_phase_error_hz, carr_phase_error_hz);
                            rec_set<LIVE>(r.carr_freq_error_hz, carr_freq_error_hz);
                            rec_set<LIVE>(r.carr_error_filt_hz, carr_error_filt_hz);
                            rec_set<LIVE>(r.rem_code_phase_samples, st_rem_code_samples);
                            rec_set<LIVE>(r.acc_carrier_phase_rad, st_acc_phase);
#ifdef GSH_TRK_PROFILE
                            if (NT == 3)  // phase durations in shader clocks, in the unused VE / VL slots
                                {
                                    r.corr[6] = static_cast<float>(t_corr_done - t_begin);
                                    const long long t_e = clock64();
                                    r.corr[7] = static_cast<float>(t_e - t_corr_done);
                                    r.corr[8] = static_cast<float>(t_join - t_corr_done);  // the four lanes side by side + their meeting (barrier or words)
                                    r.corr[9] = 0.0f;
                                    r.accu[6] = static_cast<float>(t_c - t_join);          // the join
                                    r.accu[7] = static_cast<float>(t_d - t_c);          // update_tracking_vars
                                    r.accu[8] = static_cast<float>(t_e - t_d);          // symbol bookkeeping + the record
                                }
#endif
                        }
                    const unsigned long long new_pos = pos + static_cast<unsigned long long>(prn_len);  // consume_each, trk.cc:2287
                    // the state goes back to LDS (for the launch's end, and for the lanes of the next period) ...
                    s.code_freq_chips = st_code_freq;
                    s.carrier_phase_step_rad = st_phase_step;
                    s.code_phase_step_chips = st_code_step;
                    s.rem_code_phase_samples = st_rem_code_samples;
                    s.rem_code_phase_chips = st_rem_code_chips;
                    s.acc_carrier_phase_rad = st_acc_phase;
                    s.rem_carr_phase_rad = st_rem_carr;
                    s.pos = new_pos;
                    if (HD)
                        {
                            s.carrier_phase_rate_step_rad = st_phase_rate;
                            s.code_phase_rate_step_chips = st_code_rate;
                        }
                    // ... and the next window is formed from the registers (publish(): do_correlation_step's casts, trk.cc:1237-1243; the channel is active here)
                    {
                        const float spcf = k_spcf;
                        win.pos = new_pos;
                        win.rem_carr = st_rem_carr;
                        win.phase_step = static_cast<float>(st_phase_step);
                        win.rem_code = __fmul_rn(static_cast<float>(st_rem_code_chips), spcf);
                        win.code_step = __fmul_rn(static_cast<float>(st_code_step), spcf);
                        win.phase_rate = static_cast<float>(st_phase_rate);
                        win.code_rate = __fmul_rn(static_cast<float>(st_code_rate), spcf);
                        win.go = (e + 1 < a.n_epochs && new_pos + k_vlen <= a.n_stream && new_pos >= a.ring_oldest) ? 1 : 0;
                        win.narrow = HD ? lk.narrow : st_narrow;
                        // (the tables are formed from what was said behind update_tracking_vars, and nothing between there and here assigns to st_phase_step or
                        // st_rem_carr -- the symbol machine and the record only read them --, so they are this window's: no look back into LDS on the lane the
                        // work-group waits for)
                        win.seed_ok = SEEDS ? 1 : 0;
                    }
                    if constexpr (LIVE) live_advance(a, new_pos, 1, win, lv, k_vlen);
#ifdef GSH_TRK_PROFILE
                    if (NT == 3 && a.records != nullptr) rec_ref().accu[9] = static_cast<float>(clock64() - t_corr_done);  // ... + publish
#if GSH_TRK_PROFILE == 3  // when each lane was done (the detector lanes' stamps are last period's when they run beside thread 0)
                    if (NT == 3 && a.records != nullptr)
                        {
                            gsh_trk_epoch& r = rec_ref();
                            r.accu[6] = static_cast<float>(mail.t_lane[0]);
                            r.accu[7] = static_cast<float>(mail.t_lane[1]);
                            r.accu[8] = static_cast<float>(mail.t_lane[2]);
                            r.corr[9] = static_cast<float>(mail.t_lane[3]);
                        }
#endif
#if GSH_TRK_PROFILE == 2  // the correlation phase instead of the serial section: window set-up, trips, wave sums, barrier, sum over the waves + barrier, the loop's own barrier
                    if (NT == 3 && a.records != nullptr)
                        {
                            gsh_trk_epoch& r = rec_ref();
                            using mcdev::cw_stamp;
                            r.corr[8] = static_cast<float>(cw_stamp[1] - t_begin);
                            r.corr[9] = static_cast<float>(cw_stamp[2] - cw_stamp[1]);
                            r.accu[6] = static_cast<float>(cw_stamp[3] - cw_stamp[2]);
                            r.accu[7] = static_cast<float>(cw_stamp[4] - cw_stamp[3]);
                            r.accu[8] = static_cast<float>(cw_stamp[5] - cw_stamp[4]);
                            r.accu[9] = static_cast<float>(t_corr_done - cw_stamp[5]);
                        }
#endif
#endif
                        }
                }
            if constexpr (SEEDS)
                {
                    // The seed tables of the NEXT window: two waves that have nothing else to do until the barrier that ends the period wait for thread 0's word
                    // (behind update_tracking_vars: carrier phase and phase step are final there; the symbol machine, the record and the next window -- a
                    // thousand clocks of thread 0 -- follow) and evaluate one factor per lane.  Nobody waits for them but that barrier.
                    int tl = tid;
                    asm volatile("" : "+v"(tl));
                    const int wv = tl >> 6;
                    if (GSH_TRK_SEED_TABLES != 3 && (wv == SERIAL_WAVES + 1 || wv == SERIAL_WAVES + 2))  // uniform over the wave
                        {
                            lane_waits(mail.step_seq, e + 1);
                            mcdev::seed_table_fill(seed_tab, mail.seed_step, mail.seed_rem, tl & 63, wv - (SERIAL_WAVES + 1));
                        }
                }
            done = e + 1;
            __syncthreads();
        }
    // the loop's last period ended with a barrier: state back to device memory for the next launch
    {
        // (the addresses are formed afresh -- laundered through an empty asm -- so that the ones of the prologue do not stay live, in
        // registers the correlator needs, across the whole launch)
        TrkChannel* chan_out = a.chan;
        LockState* lock_out = a.lock;
        asm volatile("" : "+s"(chan_out), "+s"(lock_out));
        unsigned* gs = reinterpret_cast<unsigned*>(chan_out + ch);
        unsigned* gl = reinterpret_cast<unsigned*>(lock_out + ch);
        const unsigned* ls = reinterpret_cast<const unsigned*>(&s);
        const unsigned* ll = reinterpret_cast<const unsigned*>(&lk);
        int t2 = tid;  // (laundered as well: the per-thread byte offset of the prologue's copy loops was kept for these -- 8 B of scratch in the tightest flavour)
        asm volatile("" : "+v"(t2));
        for (int i = t2; i < static_cast<int>(sizeof(TrkChannel) / 4); i += MC_THREADS) gs[i] = ls[i];
        for (int i = t2; i < static_cast<int>(sizeof(LockState) / 4); i += MC_THREADS) gl[i] = ll[i];
    }
    if (tid == 0)
        {
            if constexpr (LIVE)
                {
                    LiveTail* th = a.live.tail + ch;  // (pos, active and seq went out in the drain round that ended the loop)
                    __hip_atomic_store(&th->exit_reason, lv.exit_reason, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // no longer resident
                }
            else
                {
                    TrkTail tl;
                    tl.pos = s.pos;
                    tl.done = done;
                    tl.active = s.active;
                    a.tail[ch] = tl;
                }
        }
}

// ---- host: filter design in the reference's float / double mix ---------------------------------------------------
// Tracking_loop_filter::update_coefficients, T/tracking_loop_filter.cc:101-196 (Kaplan & Hegarty table 5.6, bilinear
// transform of the integrator cascade); include_last_integrator is false for the code loop (trk.cc:604)
void design_loop_filter(LoopFilterState& f, float T, float bw, int order, bool last_integrator)
{
    f = LoopFilterState{};
    const float zeta = 1.0F / std::sqrt(2.0F);
    float g1, g2, g3, wn;
    auto set_in = [&](std::initializer_list<double> v) {
        f.n_in = 0;
        for (double d : v) f.in_c[f.n_in++] = static_cast<float>(d);
    };
    auto set_out = [&](std::initializer_list<float> v) {
        f.n_out = 0;
        for (float d : v) f.out_c[f.n_out++] = d;
    };
    switch (order)
        {
        case 1:
            wn = bw * 4.0F;
            g1 = wn;
            if (last_integrator)
                {
                    set_in({g1 * T / 2.0, g1 * T / 2.0});
                    set_out({1.0F});
                }
            else
                {
                    set_in({static_cast<double>(g1)});
                    set_out({});
                }
            break;
        case 2:
            wn = bw * (8.0F * zeta) / (4.0F * zeta * zeta + 1.0F);
            g1 = wn * wn;
            g2 = wn * 2.0F * zeta;
            if (last_integrator)
                {
                    set_in({T / 2.0 * (g1 * T / 2.0 + g2), T * T / 2.0 * g1, T / 2.0 * (g1 * T / 2.0 - g2)});
                    set_out({2.0F, -1.0F});
                }
            else
                {
                    set_in({g1 * T / 2.0 + g2, g1 * T / 2.0 - g2});
                    set_out({1.0F});
                }
            break;
        default:
            {
                wn = bw / 0.7845F;
                const float a3 = 1.1;
                const float b3 = 2.4;
                g1 = wn * wn * wn;
                g2 = a3 * wn * wn;
                g3 = b3 * wn;
                if (last_integrator)
                    {
                        set_in({T / 2.0 * (g3 + T / 2.0 * (g2 + T / 2.0 * g1)), T / 2.0 * (-g3 + T / 2.0 * (g2 + 3.0 * T / 2.0 * g1)),
                            T / 2.0 * (-g3 - T / 2.0 * (g2 - 3.0 * T / 2.0 * g1)), T / 2.0 * (g3 - T / 2.0 * (g2 - T / 2.0 * g1))});
                        set_out({3.0F, -3.0F, 1.0F});
                    }
                else
                    {
                        set_in({g3 + T / 2.0 * (g2 + T / 2.0 * g1), g1 * T * T / 2.0 - 2.0 * g3, g3 + T / 2.0 * (-g2 + T / 2.0 * g1)});
                        set_out({2.0F, -1.0F});
                    }
            }
            break;
        }
    // initialize(0), T/tracking_loop_filter.cc:266-271
    for (int i = 0; i < 4; i++)
        {
            f.in_h[i] = 0.0F;
            f.out_h[i] = 0.0F;
        }
    f.idx = 0;  // (unused: the device keeps the histories newest first, loop_filter_apply)
}

// Tracking_FLL_PLL_filter::set_params + initialize, T/tracking_FLL_PLL_filter.cc:23-69
void design_fll_pll(FllPllState& f, float fll_bw_hz, float pll_bw_hz, int order, float acq_doppler_hz)
{
    f = FllPllState{};
    f.order = order;
    if (order == 3)
        {
            f.b3 = 2.400;
            f.a3 = 1.100;
            f.a2 = 1.414;
            f.w0p = pll_bw_hz / 0.7845F;
            f.w0p2 = f.w0p * f.w0p;
            f.w0p3 = f.w0p2 * f.w0p;
            f.w0f = fll_bw_hz / 0.53F;
            f.w0f2 = f.w0f * f.w0f;
            f.x = 2.0F * acq_doppler_hz;
            f.w = 0;
        }
    else
        {
            f.a2 = 1.414;
            f.w0p = pll_bw_hz / 0.53F;
            f.w0p2 = f.w0p * f.w0p;
            f.w0f = fll_bw_hz / 0.25F;
            f.w = acq_doppler_hz;
            f.x = 0;
        }
}
}  // namespace
}  // namespace gsh

struct gsh_trk
{
    int device{0};
    gsh_trk_conf conf{};
    int n_channels{0};
    int max_code_len{0};
    hipStream_t stream{nullptr};
    float* d_codes{nullptr};
    gsh::TrkChannel* d_chan{nullptr};
    gsh::TrkChannel* d_chan_backup{nullptr};
    gsh_trk_conf* d_conf{nullptr};           // device copy of conf (kernel argument by pointer)
    gsh::LockState* d_lock{nullptr};         // lock-detector state per channel (enable_lock_detectors)
    gsh::LockState* d_lock_backup{nullptr};
    std::vector<gsh::TrkChannel> h_chan;
    float2* d_stream_owned{nullptr};
    size_t stream_owned_cap{0};
    const float2* d_stream{nullptr};
    unsigned long long n_stream{0};
    gsh_stream* ring{nullptr};  // when set, the loop follows the live ring: positions are absolute sample indices
    gsh_trk_epoch* d_records{nullptr};
    size_t records_cap{0};
    gsh::TrkTail* d_tail{nullptr};
    hipEvent_t ev0{nullptr}, ev1{nullptr};
    // gsh_trk_run_begin / _end: results land in page-locked host memory so that the copies are true asynchronous DMAs
    gsh_trk_epoch* h_records{nullptr};
    size_t h_records_cap{0};
    gsh::TrkTail* h_tail{nullptr};           // n_channels: position / periods done / active flag after the run
    int pending_epochs{-1};                  // >= 0: a run has been begun and not ended
    bool host_records{true};                 // the kernel writes records and tails straight into the page-locked host buffers: no copies queued behind it, 4 - 20 us less
                                             // per launch (profiles/ab/r03/loop_host_records.txt).  GSH_TRK_HOST_RECORDS=0: device buffers + two copies, as before
    bool pending_records{false};
    // ---- live mode (gsh_trk_live_*): everything the resident kernel and the host share lies in coherent page-locked host memory
    hipStream_t live_stream{nullptr};               // lowest priority: a queue of its own kind, never the one a push travels in (sample_stream.hip)
    gsh::LiveTail* h_live_tail{nullptr};            // n_channels
    unsigned long long* h_live_consumed{nullptr};   // n_channels: records taken by the host
    int* h_live_quit{nullptr};
    gsh_trk_epoch* h_live_records{nullptr};         // n_channels * live_ring_len
    unsigned live_ring_len{0};
    std::vector<unsigned long long> live_next_window;  // first sample of the window behind the last record TAKEN (what the caller's block has to be offered next)
    hipEvent_t live_ev[2]{nullptr, nullptr};
    bool live_busy[2]{false, false};                // a residency has been queued and its event has not been seen complete yet
    unsigned live_idle_us{1000}, live_residency_us{20000};  // (round 4: 200 us / 5 ms until the quit word was polled often enough to end a residency within half a millisecond; a restart is ~0.25 ms of nobody advancing: +5 - 8 % through the blocks, profiles/ab/r04/dropin_push_notes.txt)
    std::shared_ptr<gsh::LiveFloor> live_floor;     // registered with the ring: pushes keep off what the channels still read
    std::atomic<bool> live_ready{false};            // live_setup has run: what gsh_trk_live_take (any thread) reads is in place
    // ---- cooperating work-groups (gsh_trk_set_split): launched standard-mode runs only
    int split{1};                                   // work-groups per channel
    unsigned long long* d_coop{nullptr};            // n_channels * coop_stride(split) + 1 tagged words (TrkArgs::coop_box)
    unsigned coop_seq{1};                           // tag of the next launch's first window
};

namespace
{
using gsh::set_error;

size_t trk_lds_bytes(const gsh_trk* t)
{
    const size_t tabs = static_cast<size_t>(gsh::mcdev::code_table_floats(t->max_code_len)) * (t->conf.track_pilot ? 2 : 1);
    return tabs * sizeof(float) + (gsh::mcdev::MC_WAVES + 1) * GSH_MAX_TAPS * sizeof(float2);  // outputs + one row of partial sums per wave (correlate_window)
}

// div_by_constant's preconditions for a whole configuration: the two divisors positive, finite, not with a significand of all ones, and every constant that
// enters a dividend (sampling rate, chip rate, carrier frequency, the front end's frequency offset) of a magnitude that keeps products and residuals inside the
// normal numbers whatever float the loop filters deliver.  Any receiver configuration passes; one that does not gets the plain divisions (reciprocals 0.0).
bool fast_division_applies(const gsh_trk_conf& c)
{
    auto sane = [](double v, double lo, double hi) { return v >= lo && v <= hi; };
    auto all_ones = [](double v) {
        unsigned long long bits;
        std::memcpy(&bits, &v, sizeof(bits));
        return (bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;
    };
    if (!sane(c.fs_in, 1.0, 1.0e12) || !sane(c.code_chip_rate, 1.0e-3, 1.0e12) || !sane(c.signal_carrier_freq, 1.0, 1.0e15)) return false;
    if (!(c.cfo_frequency_hz == 0.0 || sane(std::fabs(c.cfo_frequency_hz), 1.0e-30, 1.0e12))) return false;
    return !all_ones(c.fs_in) && !all_ones(c.signal_carrier_freq);
}

int trk_launch(gsh_trk* t, int n_epochs, gsh_trk_epoch* d_records, gsh::TrkTail* d_tail = nullptr, const gsh::LiveArgs* live = nullptr)
{
    gsh::TrkArgs a;
    a.live = gsh::LiveArgs{};
    if (live != nullptr) a.live = *live;
    a.conf = t->d_conf;
    a.stream = t->d_stream;
    a.n_stream = t->n_stream;
    a.ring_capacity = 0;
    a.ring_oldest = 0;
    if (live != nullptr)
        {
            // a residency learns how far the ring is complete from the ring's live words, not from the host: no event waits in either direction
            a.stream = t->ring->d_ring;
            a.ring_capacity = t->ring->capacity;
            a.n_stream = 0ull;
        }
    else if (t->ring != nullptr)
        {
            a.stream = t->ring->d_ring;
            a.ring_capacity = t->ring->capacity;
            a.ring_oldest = gsh::stream_oldest(t->ring);
            // The launch reads, and therefore waits for, only what its channels can reach in n_epochs periods -- not every push queued so far: the blocks of a
            // stream upload well ahead of what they consume (the scheduler offers them its whole buffer), and the upload of what the NEXT launch will read
            // runs under this one.  (A period is vector_length samples give or take the code Doppler; a channel that drifts past the estimate just stops a
            // period early in this launch and carries on in the next.)
            unsigned long long need = 0ull;
            bool any = false;
            for (int ch = 0; ch < t->n_channels; ch++)
                if (t->h_chan[ch].active)
                    {
                        any = true;
                        need = std::max(need, t->h_chan[ch].pos + static_cast<unsigned long long>(n_epochs) * (t->conf.vector_length + 8ull) + t->conf.vector_length);
                    }
            const unsigned long long limit = any ? std::min<unsigned long long>(t->ring->next, need) : t->ring->next;
            a.n_stream = limit;
            const int rcw = gsh::stream_wait_pushed(t->ring, limit, t->stream);  // conversions queued by the pushes that cover [.., limit)
            if (rcw != GSH_OK) return rcw;
        }
    a.codes = t->d_codes;
    a.code_stride = t->max_code_len;
    a.chan = t->d_chan;
    a.lock = t->d_lock;
    a.records = d_records;
    a.tail = d_tail != nullptr ? d_tail : t->d_tail;
    a.n_epochs = n_epochs;
    a.code_period = static_cast<double>(t->conf.code_length_chips) / t->conf.code_chip_rate;
    const bool fast_div = fast_division_applies(t->conf);
    a.inv_fs_in = fast_div ? 1.0 / t->conf.fs_in : 0.0;
    a.inv_signal_carrier_freq = fast_div ? 1.0 / t->conf.signal_carrier_freq : 0.0;
    {
        // trk.cc:1912-1915: the transitory lasts while !(pull_in_time_s < (samples since acquisition) / fs) in integer arithmetic, i.e. while the sample
        // count is below (pull_in_time_s + 1) * fs -- one comparison per period instead of a 64-bit division
        const unsigned long long f = static_cast<unsigned long long>(static_cast<int>(t->conf.fs_in));
        a.pull_in_limit = (static_cast<unsigned long long>(t->conf.pull_in_time_s) + 1ull) * f;
        // trk.cc:2002: limit < (samples since acquisition) / fs, the same integer arithmetic
        a.bit_sync_limit = (t->conf.enable_bit_sync_time_limit && t->conf.enable_symbol_sync)
                               ? (static_cast<unsigned long long>(t->conf.bit_synchronization_time_limit_s) + 1ull) * f
                               : ~0ull;
    }
    const size_t lds = trk_lds_bytes(t);
    const bool coop = live == nullptr && t->split > 1 && t->d_coop != nullptr && !t->conf.high_dyn;
    a.coop_box = coop ? t->d_coop : nullptr;
    a.coop_g = coop ? t->split : 1;
    a.coop_channels = t->n_channels;
    {
        static const int user = [] { const char* e = std::getenv("GSH_TRK_SPLIT_HELPER_TRIPS"); return e != nullptr ? std::atoi(e) : 0; }();
        a.coop_helper_trips = user;
    }
    a.coop_seq0 = t->coop_seq;
    if (coop) t->coop_seq += static_cast<unsigned>(n_epochs) + 2u;  // (tags never repeat within 2^32 periods)
    const dim3 grid(coop ? static_cast<unsigned>((t->n_channels + 7) / 8 * 8 * t->split) : static_cast<unsigned>(t->n_channels)), block(gsh::mcdev::MC_THREADS);
    hipStream_t st = live != nullptr ? t->live_stream : t->stream;
    if (coop)
        {
            if (t->conf.veml)
                hipLaunchKernelGGL((gsh::trk_loop_kernel<5, false, false, true>), grid, block, lds, st, a, a.conf);
            else
                hipLaunchKernelGGL((gsh::trk_loop_kernel<3, false, false, true>), grid, block, lds, st, a, a.conf);
        }
    else
    {
        using KernelFn = void (*)(gsh::TrkArgs, const gsh_trk_conf*);
        const bool L = live != nullptr;
        KernelFn fn;
        if (t->conf.veml)
            fn = t->conf.high_dyn ? (L ? gsh::trk_loop_kernel<5, true, true> : gsh::trk_loop_kernel<5, true, false>) : (L ? gsh::trk_loop_kernel<5, false, true> : gsh::trk_loop_kernel<5, false, false>);
        else
            fn = t->conf.high_dyn ? (L ? gsh::trk_loop_kernel<3, true, true> : gsh::trk_loop_kernel<3, true, false>) : (L ? gsh::trk_loop_kernel<3, false, true> : gsh::trk_loop_kernel<3, false, false>);
        hipLaunchKernelGGL(fn, grid, block, lds, st, a, a.conf);
    }
    GSH_HIP(hipGetLastError());
    if (live == nullptr && t->ring != nullptr)
        {
            // a later push waits for this launch only if it overwrites samples at or above the oldest window a running channel starts at
            unsigned long long lowest = ~0ull;
            for (const auto& c : t->h_chan)
                if (c.active && c.pos < lowest) lowest = c.pos;
            return gsh::stream_mark_read(t->ring, lowest, t->stream);
        }
    return GSH_OK;
}

// after a launched run has completed (the stream is idle): did a cooperating work-group give up on its partner?
int coop_check(gsh_trk* t)
{
    if (t->d_coop == nullptr || t->split <= 1) return GSH_OK;
    unsigned long long* err = t->d_coop + static_cast<size_t>(t->n_channels) * gsh::coop_stride(t->split);
    unsigned long long v = 0ull;
    GSH_HIP(hipMemcpy(&v, err, sizeof(v), hipMemcpyDeviceToHost));
    if (v == 0ull) return GSH_OK;
    GSH_HIP(hipMemset(err, 0, sizeof(v)));
    return set_error(GSH_ERR_STATE, "gsh_trk_run: a cooperating work-group did not answer within 0.2 s (gsh_trk_set_split needs every work-group of the launch resident at once: "
                                    "nothing else may occupy the device's compute units meanwhile); the channels stopped where they were");
}

// ---- live mode, host side ---------------------------------------------------------------------------------------------------------------------
// residencies whose event has completed are no longer in flight
int live_reap(gsh_trk* t)
{
    int n = 0;
    for (int i = 0; i < 2; i++)
        {
            if (!t->live_busy[i]) continue;
            const hipError_t e = hipEventQuery(t->live_ev[i]);
            if (e == hipSuccess)
                t->live_busy[i] = false;
            else if (e == hipErrorNotReady)
                n++;
            else
                {
                    (void)gsh::hip_fail(e, "hipEventQuery(residency)", __FILE__, __LINE__);
                    return -1;
                }
        }
    return n;
}

void live_release(gsh_trk* t)  // the handle's side of the registration with its ring, and the memory the registration points into
{
    if (t->live_floor)
        {
            std::lock_guard<std::mutex> lk(t->live_floor->m);
            t->live_floor->tails = nullptr;
            t->live_floor->n = 0;
        }
    t->live_floor.reset();
}

int live_quiesce(gsh_trk* t);

// gsh_stream_destroy found this handle still registered with the ring: the residencies leave (the ring's memory is about to be freed), the registration goes, the
// handle forgets the ring.  The record rings in host memory stay (block threads may be reading them); what the channels had finished can still be taken.  Whatever
// the handle is asked to do with a stream next fails with GSH_ERR_STATE until gsh_trk_set_stream_* gives it one.
void live_ring_gone(void* owner)
{
    gsh_trk* t = static_cast<gsh_trk*>(owner);
    (void)hipSetDevice(t->device);
    (void)live_quiesce(t);
    live_release(t);
    t->ring = nullptr;
    t->d_stream = nullptr;
    t->n_stream = 0;
}

// first use: the shared words and the record ring in coherent host memory, the stream, the registration with the ring
int live_setup(gsh_trk* t)
{
    if (t->h_live_tail != nullptr) return GSH_OK;
    GSH_HIP(hipSetDevice(t->device));
    if (const char* e = std::getenv("GSH_TRK_LIVE_IDLE_US")) t->live_idle_us = static_cast<unsigned>(std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("GSH_TRK_LIVE_RESIDENCY_US")) t->live_residency_us = static_cast<unsigned>(std::max(10, std::atoi(e)));
    // as many records as periods the ring can hold ahead of the slowest reader, and then some: the device never has to wait for the host in practice
    unsigned len = 64;
    const unsigned long long want = t->ring->capacity / std::max<unsigned long long>(t->conf.vector_length, 1ull) + 32ull;
    while (len < want && len < 4096u) len <<= 1;
    if (const char* e = std::getenv("GSH_TRK_LIVE_RECORDS"))
        {
            unsigned v = static_cast<unsigned>(std::max(2, std::atoi(e))), pw = 2;
            while (pw < v && pw < 65536u) pw <<= 1;
            len = pw;
        }
    const unsigned flags = hipHostMallocCoherent | hipHostMallocMapped;
    gsh::LiveTail* tails = nullptr;
    unsigned long long* consumed = nullptr;
    int* quit = nullptr;
    gsh_trk_epoch* recs = nullptr;
    auto undo = [&]() {
        if (tails) (void)hipHostFree(tails);
        if (consumed) (void)hipHostFree(consumed);
        if (quit) (void)hipHostFree(quit);
        if (recs) (void)hipHostFree(recs);
    };
    hipError_t e;
    if ((e = hipHostMalloc(&tails, sizeof(gsh::LiveTail) * t->n_channels, flags)) != hipSuccess ||
        (e = hipHostMalloc(&consumed, sizeof(unsigned long long) * t->n_channels, flags)) != hipSuccess ||
        (e = hipHostMalloc(&quit, sizeof(int) * 16, flags)) != hipSuccess ||
        (e = hipHostMalloc(&recs, sizeof(gsh_trk_epoch) * static_cast<size_t>(t->n_channels) * len, flags)) != hipSuccess)
        {
            undo();
            return gsh::hip_fail(e, "hipHostMalloc(live)", __FILE__, __LINE__);
        }
    if (t->live_stream == nullptr)
        {
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
            int prio = least;
            if (const char* pe = std::getenv("GSH_TRK_LIVE_PRIORITY")) prio = std::min(std::max(std::atoi(pe), greatest), least);  // (A/B runs)
            if ((e = hipStreamCreateWithPriority(&t->live_stream, hipStreamNonBlocking, prio)) != hipSuccess)
                {
                    undo();
                    return gsh::hip_fail(e, "hipStreamCreateWithPriority(live)", __FILE__, __LINE__);
                }
        }
    for (int i = 0; i < 2; i++)
        if (t->live_ev[i] == nullptr && (e = hipEventCreateWithFlags(&t->live_ev[i], hipEventDisableTiming)) != hipSuccess)
            {
                undo();
                return gsh::hip_fail(e, "hipEventCreate(live)", __FILE__, __LINE__);
            }
    std::memset(recs, 0, sizeof(gsh_trk_epoch) * static_cast<size_t>(t->n_channels) * len);
    *quit = 0;
    t->live_next_window.assign(static_cast<size_t>(t->n_channels), 0ull);
    for (int ch = 0; ch < t->n_channels; ch++)
        {
            tails[ch].pos = t->h_chan[ch].pos;
            tails[ch].seq = 0ull;
            tails[ch].active = t->h_chan[ch].active;
            tails[ch].exit_reason = gsh::LIVE_EXIT_IDLE;  // (no work-group is resident for the channel)
            consumed[ch] = 0ull;
            t->live_next_window[ch] = t->h_chan[ch].pos;
        }
    t->h_live_tail = tails;
    t->h_live_consumed = consumed;
    t->h_live_quit = quit;
    t->h_live_records = recs;
    t->live_ring_len = len;
    t->live_floor = std::make_shared<gsh::LiveFloor>();
    t->live_floor->tails = tails;
    t->live_floor->n = t->n_channels;
    t->live_floor->owner = t;
    t->live_floor->ring_gone = &live_ring_gone;
    t->ring->live_floors.push_back(t->live_floor);
    t->live_ready.store(true, std::memory_order_release);
    return GSH_OK;
}

// the host's copy of where the channels stand, after the device has gone quiet
void live_refresh_host_state(gsh_trk* t)
{
    if (t->h_live_tail == nullptr) return;
    for (int ch = 0; ch < t->n_channels; ch++)
        {
            t->h_chan[ch].pos = t->h_live_tail[ch].pos;
            t->h_chan[ch].active = t->h_live_tail[ch].active;
        }
}

int live_quiesce(gsh_trk* t)
{
    if (t->h_live_tail == nullptr || t->live_stream == nullptr) return GSH_OK;
    if (!t->live_busy[0] && !t->live_busy[1]) return GSH_OK;
    __atomic_store_n(t->h_live_quit, 1, __ATOMIC_RELEASE);
    const hipError_t e = hipStreamSynchronize(t->live_stream);
    __atomic_store_n(t->h_live_quit, 0, __ATOMIC_RELEASE);
    t->live_busy[0] = t->live_busy[1] = false;
    if (e != hipSuccess) return gsh::hip_fail(e, "hipStreamSynchronize(live)", __FILE__, __LINE__);
    live_refresh_host_state(t);
    return GSH_OK;
}

// live mode was set up against a ring the handle is leaving (another ring, a flat stream): residencies leave, the registration with the old ring goes, the host
// memory of the record rings is released; the next gsh_trk_live_begin starts from scratch
int live_wind_down(gsh_trk* t)
{
    if (t->h_live_tail == nullptr) return GSH_OK;
    GSH_HIP(hipSetDevice(t->device));
    const int rcq = live_quiesce(t);
    if (rcq != GSH_OK) return rcq;
    live_release(t);
    (void)hipHostFree(t->h_live_tail);
    (void)hipHostFree(t->h_live_consumed);
    (void)hipHostFree(t->h_live_quit);
    (void)hipHostFree(t->h_live_records);
    t->live_ready.store(false, std::memory_order_release);
    t->h_live_tail = nullptr;
    t->h_live_consumed = nullptr;
    t->h_live_quit = nullptr;
    t->h_live_records = nullptr;
    return GSH_OK;
}
}  // namespace

extern "C"
{
    int gsh_trk_create(int device, const gsh_trk_conf* conf, int n_channels, int max_code_length, gsh_trk_t** out)
    {
        GSH_REQUIRE(out != nullptr && conf != nullptr, "null argument");
        *out = nullptr;
        const gsh_trk_conf& c = *conf;
        GSH_REQUIRE(n_channels >= 1 && n_channels <= 65535, "n_channels %d outside 1..65535", n_channels);
        GSH_REQUIRE(max_code_length >= 1, "max_code_length %d", max_code_length);
        GSH_REQUIRE(c.fs_in >= 1.0 && c.code_chip_rate > 0.0 && c.signal_carrier_freq > 0.0, "fs_in, code_chip_rate and signal_carrier_freq must be positive");
        GSH_REQUIRE(c.code_length_chips >= 1 && c.code_samples_per_chip >= 1 && c.vector_length >= 1, "code_length_chips, code_samples_per_chip, vector_length must be >= 1");
        GSH_REQUIRE(static_cast<int>(static_cast<double>(c.code_length_chips) / c.code_chip_rate * 1000.0) >= 1,
            "code period %g s is shorter than 1 ms (cn0 smoother set-up divides by its whole milliseconds, trk.cc:620-627)", static_cast<double>(c.code_length_chips) / c.code_chip_rate);
        GSH_REQUIRE(c.pll_filter_order == 2 || c.pll_filter_order == 3, "pll_filter_order %d (2 or 3: T/tracking_FLL_PLL_filter.cc:23-54)", c.pll_filter_order);
        GSH_REQUIRE(c.dll_filter_order >= 1 && c.dll_filter_order <= 3, "dll_filter_order %d outside 1..3", c.dll_filter_order);
        GSH_REQUIRE(!c.high_dyn || (c.smoother_length >= 1 && c.smoother_length <= GSH_MAX_SMOOTHER), "smoother_length %u outside 1..%d", c.smoother_length, GSH_MAX_SMOOTHER);
        if (c.enable_symbol_sync)
            {
                GSH_REQUIRE(c.secondary_code_length >= 0 && c.secondary_code_length <= GSH_MAX_SECONDARY, "secondary_code_length %d outside 0..%d", c.secondary_code_length, GSH_MAX_SECONDARY);
                GSH_REQUIRE(c.data_secondary_code_length >= 0 && c.data_secondary_code_length <= GSH_MAX_SECONDARY, "data_secondary_code_length %d outside 0..%d", c.data_secondary_code_length, GSH_MAX_SECONDARY);
                GSH_REQUIRE(!c.has_secondary || c.secondary_code_length >= 1, "has_secondary needs a secondary code");
                GSH_REQUIRE(c.symbols_per_bit >= 0, "symbols_per_bit %d", c.symbols_per_bit);
                GSH_REQUIRE(!c.use_histogram_bit_sync || c.symbols_per_bit <= GSH_MAX_BITSYNC_BINS, "symbols_per_bit %d exceeds the %d histogram bins", c.symbols_per_bit, GSH_MAX_BITSYNC_BINS);
                GSH_REQUIRE(c.extend_correlation_symbols >= 0 && c.extend_correlation_symbols <= 1000, "extend_correlation_symbols %d", c.extend_correlation_symbols);
            }
        if (c.enable_lock_detectors)
            {
                GSH_REQUIRE(c.cn0_samples >= 1 && c.cn0_samples <= GSH_MAX_CN0_SAMPLES, "cn0_samples %d outside 1..%d", c.cn0_samples, GSH_MAX_CN0_SAMPLES);
                GSH_REQUIRE(c.max_code_lock_fail >= 0 && c.max_carrier_lock_fail >= 0, "lock fail limits must not be negative");
            }
        GSH_REQUIRE(static_cast<uint64_t>(c.code_length_chips) * c.code_samples_per_chip <= static_cast<uint64_t>(max_code_length), "max_code_length %d smaller than code_length_chips * code_samples_per_chip", max_code_length);
        gsh_trk probe;
        probe.conf = c;
        probe.max_code_len = max_code_length;
        GSH_REQUIRE(trk_lds_bytes(&probe) <= 160 * 1024, "local replica(s) of %d samples do not fit the 160 KiB LDS", max_code_length);
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh_trk* t = new (std::nothrow) gsh_trk();
        GSH_REQUIRE(t != nullptr, "out of host memory");
        t->device = device;
        t->conf = c;
        t->n_channels = n_channels;
        t->max_code_len = max_code_length;
        t->h_chan.assign(static_cast<size_t>(n_channels), gsh::TrkChannel{});
        auto fail = [&](hipError_t e, const char* what) {
            gsh::hip_fail(e, what, __FILE__, __LINE__);
            gsh_trk_destroy(t);
            return GSH_ERR_HIP;
        };
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
        if ((e = hipMalloc(&t->d_codes, sizeof(float) * 2 * static_cast<size_t>(n_channels) * max_code_length)) != hipSuccess) return fail(e, "hipMalloc(codes)");
        if ((e = hipMalloc(&t->d_chan, sizeof(gsh::TrkChannel) * n_channels)) != hipSuccess) return fail(e, "hipMalloc(state)");
        if ((e = hipMalloc(&t->d_chan_backup, sizeof(gsh::TrkChannel) * n_channels)) != hipSuccess) return fail(e, "hipMalloc(state)");
        if ((e = hipMemset(t->d_chan, 0, sizeof(gsh::TrkChannel) * n_channels)) != hipSuccess) return fail(e, "hipMemset(state)");
        if ((e = hipMalloc(&t->d_lock, sizeof(gsh::LockState) * n_channels)) != hipSuccess) return fail(e, "hipMalloc(lock)");
        if ((e = hipMalloc(&t->d_lock_backup, sizeof(gsh::LockState) * n_channels)) != hipSuccess) return fail(e, "hipMalloc(lock)");
        if ((e = hipMemset(t->d_lock, 0, sizeof(gsh::LockState) * n_channels)) != hipSuccess) return fail(e, "hipMemset(lock)");
        if ((e = hipMalloc(&t->d_tail, sizeof(gsh::TrkTail) * n_channels)) != hipSuccess) return fail(e, "hipMalloc(tail)");
        if ((e = hipHostMalloc(&t->h_tail, sizeof(gsh::TrkTail) * n_channels, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc(tail)");
        if (const char* hr = std::getenv("GSH_TRK_HOST_RECORDS")) t->host_records = (std::atoi(hr) != 0);
        if ((e = hipMalloc(&t->d_conf, sizeof(gsh_trk_conf))) != hipSuccess) return fail(e, "hipMalloc(conf)");
        if ((e = hipMemcpy(t->d_conf, &t->conf, sizeof(gsh_trk_conf), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(conf)");
        if ((e = hipEventCreate(&t->ev0)) != hipSuccess) return fail(e, "hipEventCreate");
        if ((e = hipEventCreate(&t->ev1)) != hipSuccess) return fail(e, "hipEventCreate");
        if (trk_lds_bytes(t) > 64 * 1024)
            {
                const int bytes = static_cast<int>(trk_lds_bytes(t));
                const void* fns[8] = {reinterpret_cast<const void*>(gsh::trk_loop_kernel<3, false, false>), reinterpret_cast<const void*>(gsh::trk_loop_kernel<5, false, false>),
                    reinterpret_cast<const void*>(gsh::trk_loop_kernel<3, true, false>), reinterpret_cast<const void*>(gsh::trk_loop_kernel<5, true, false>),
                    reinterpret_cast<const void*>(gsh::trk_loop_kernel<3, false, true>), reinterpret_cast<const void*>(gsh::trk_loop_kernel<5, false, true>),
                    reinterpret_cast<const void*>(gsh::trk_loop_kernel<3, true, true>), reinterpret_cast<const void*>(gsh::trk_loop_kernel<5, true, true>)};
                for (const void* fn : fns)
                    if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return fail(e, "hipFuncSetAttribute");
            }
        *out = t;
        return GSH_OK;
    }

    void gsh_trk_destroy(gsh_trk_t* t)
    {
        if (!t) return;
        (void)hipSetDevice(t->device);
        (void)live_quiesce(t);
        live_release(t);
        if (t->live_stream) (void)hipStreamSynchronize(t->live_stream);
        for (int i = 0; i < 2; i++)
            if (t->live_ev[i]) (void)hipEventDestroy(t->live_ev[i]);
        if (t->live_stream) (void)hipStreamDestroy(t->live_stream);
        if (t->h_live_tail) (void)hipHostFree(t->h_live_tail);
        if (t->h_live_consumed) (void)hipHostFree(t->h_live_consumed);
        if (t->h_live_quit) (void)hipHostFree(t->h_live_quit);
        if (t->h_live_records) (void)hipHostFree(t->h_live_records);
        if (t->stream) (void)hipStreamSynchronize(t->stream);
        if (t->d_codes) (void)hipFree(t->d_codes);
        if (t->d_chan) (void)hipFree(t->d_chan);
        if (t->d_chan_backup) (void)hipFree(t->d_chan_backup);
        if (t->d_lock) (void)hipFree(t->d_lock);
        if (t->d_lock_backup) (void)hipFree(t->d_lock_backup);
        if (t->d_stream_owned) (void)hipFree(t->d_stream_owned);
        if (t->d_records) (void)hipFree(t->d_records);
        if (t->d_tail) (void)hipFree(t->d_tail);
        if (t->d_coop) (void)hipFree(t->d_coop);
        if (t->d_conf) (void)hipFree(t->d_conf);
        if (t->h_records) (void)hipHostFree(t->h_records);
        if (t->h_tail) (void)hipHostFree(t->h_tail);
        if (t->ev0) (void)hipEventDestroy(t->ev0);
        if (t->ev1) (void)hipEventDestroy(t->ev1);
        if (t->stream) (void)hipStreamDestroy(t->stream);
        delete t;
    }

    int gsh_trk_set_stream_host(gsh_trk_t* t, const float* iq, uint64_t n_samples)
    {
        GSH_REQUIRE(t != nullptr && iq != nullptr && n_samples >= 1, "null / empty stream");
        GSH_HIP(hipSetDevice(t->device));
        {
            // (a residency still follows the ring the handle is leaving: it goes first -- it would keep reading the old ring, and its registration would keep
            // pushes into that ring off the channels' stale positions)
            const int rcq = live_wind_down(t);
            if (rcq != GSH_OK) return rcq;
        }
        if (n_samples + 2 > t->stream_owned_cap)
            {
                if (t->d_stream_owned) GSH_HIP(hipFree(t->d_stream_owned));
                t->d_stream_owned = nullptr;
                t->stream_owned_cap = 0;
                GSH_HIP(hipMalloc(&t->d_stream_owned, sizeof(float2) * (n_samples + 2)));  // +2: the last 16-byte load of an odd window
                t->stream_owned_cap = n_samples + 2;
            }
        GSH_HIP(hipMemsetAsync(t->d_stream_owned + n_samples, 0, sizeof(float2) * 2, t->stream));
        GSH_HIP(hipMemcpyAsync(t->d_stream_owned, iq, sizeof(float2) * n_samples, hipMemcpyHostToDevice, t->stream));
        GSH_HIP(hipStreamSynchronize(t->stream));
        t->d_stream = t->d_stream_owned;
        t->n_stream = n_samples;
        t->ring = nullptr;
        return GSH_OK;
    }

    int gsh_trk_set_stream_device(gsh_trk_t* t, const void* device_iq, uint64_t n_samples)
    {
        GSH_REQUIRE(t != nullptr && device_iq != nullptr && n_samples >= 1, "null / empty stream");
        GSH_REQUIRE((reinterpret_cast<uintptr_t>(device_iq) & 15u) == 0, "device stream must be 16-byte aligned");
        {
            const int rcq = live_wind_down(t);  // as in gsh_trk_set_stream_host
            if (rcq != GSH_OK) return rcq;
        }
        t->d_stream = static_cast<const float2*>(device_iq);
        t->n_stream = n_samples;
        t->ring = nullptr;
        return GSH_OK;
    }

    int gsh_trk_set_stream_ring(gsh_trk_t* t, gsh_stream_t* s)
    {
        GSH_REQUIRE(t != nullptr, "null handle");
        GSH_REQUIRE(s == nullptr || s->device == t->device, "the ring lives on device %d, the loop on device %d", s ? s->device : -1, t->device);
        GSH_REQUIRE(s == nullptr || s->max_window >= t->conf.vector_length, "the ring's max_window_samples %llu is shorter than vector_length %u",
            s ? s->max_window : 0ull, t->conf.vector_length);
        if (s != t->ring && t->h_live_tail != nullptr)
            {
                const int rcq = live_wind_down(t);  // live mode was set up against the old ring; the next gsh_trk_live_begin sets it up against the new one
                if (rcq != GSH_OK) return rcq;
            }
        t->ring = s;
        if (s != nullptr)
            {
                t->d_stream = s->d_ring;
                t->n_stream = s->next;
            }
        else
            {
                t->d_stream = nullptr;
                t->n_stream = 0;
            }
        return GSH_OK;
    }

    int gsh_trk_pull_in(const gsh_trk_conf* conf, uint64_t nitems_read, double acq_delay_samples, uint64_t acq_sample_stamp, double acq_carrier_doppler_hz,
        int32_t* samples_offset, int32_t* first_prn_length_samples, double* acc_carrier_phase_rad)
    {
        // dll_pll_veml_tracking::general_work, case 1 (trk.cc:1949-1978), statement by statement, in double as written there
        GSH_REQUIRE(conf != nullptr && samples_offset != nullptr, "null argument");
        GSH_REQUIRE(conf->fs_in > 0.0 && conf->code_chip_rate > 0.0 && conf->code_length_chips >= 1, "fs_in, code_chip_rate, code_length_chips must be positive");
        const int64_t acq_trk_diff_samples = static_cast<int64_t>(nitems_read) - static_cast<int64_t>(acq_sample_stamp);
        const double delta_trk_to_acq_prn_start_samples = static_cast<double>(acq_trk_diff_samples) - acq_delay_samples;
        const double code_freq_chips = conf->code_chip_rate;
        const double T_chip_mod_seconds = 1.0 / code_freq_chips;
        const double T_prn_mod_seconds = T_chip_mod_seconds * static_cast<double>(conf->code_length_chips);
        const double T_prn_mod_samples = T_prn_mod_seconds * conf->fs_in;
        const double acq_code_phase_samples = T_prn_mod_samples - std::fmod(delta_trk_to_acq_prn_start_samples, T_prn_mod_samples);
        const int32_t offset = static_cast<int32_t>(std::round(acq_code_phase_samples));
        *samples_offset = offset;
        if (first_prn_length_samples != nullptr) *first_prn_length_samples = static_cast<int32_t>(std::round(T_prn_mod_samples));
        if (acc_carrier_phase_rad != nullptr)
            {
                const double carrier_phase_step_rad = gsh::GNSS_TWO_PI_D * (acq_carrier_doppler_hz + conf->cfo_frequency_hz) / conf->fs_in;  // start_tracking, trk.cc:800-801; Glonass: + the FDMA channel offset, :1003
                *acc_carrier_phase_rad = 0.0 - carrier_phase_step_rad * static_cast<double>(offset);           // :1966 (d_acc_carrier_phase_rad is 0 after start_tracking)
            }
        return GSH_OK;
    }

    int gsh_trk_pull_in_over(const gsh_trk_conf* conf, uint64_t nitems_read, uint64_t acq_sample_stamp)
    {
        // trk.cc:1912 as written: uint32 < (uint64 - uint64) / int -- the int is converted to uint64 for the division
        if (conf == nullptr || static_cast<int>(conf->fs_in) <= 0) return 0;
        const uint64_t elapsed_s = (nitems_read - acq_sample_stamp) / static_cast<uint64_t>(static_cast<int>(conf->fs_in));
        return static_cast<uint64_t>(conf->pull_in_time_s) < elapsed_s ? 1 : 0;
    }

    int gsh_trk_start(gsh_trk_t* t, int channel, const float* code, const float* data_code, int code_length, uint64_t start_sample,
        uint64_t acq_sample_stamp, double acq_carrier_doppler_hz)
    {
        return gsh_trk_start_ex(t, channel, code, data_code, code_length, start_sample, acq_sample_stamp, acq_carrier_doppler_hz, 0.0);
    }

    int gsh_trk_stop(gsh_trk_t* t, int channel)
    {
        GSH_REQUIRE(t != nullptr, "null handle");
        GSH_REQUIRE(channel >= 0 && channel < t->n_channels, "channel %d outside 0..%d", channel, t->n_channels - 1);
        if (t->pending_epochs >= 0) return set_error(GSH_ERR_STATE, "gsh_trk_stop: a run has been begun and not ended");
        GSH_HIP(hipSetDevice(t->device));
        if (live_reap(t) != 0) return set_error(GSH_ERR_STATE, "gsh_trk_stop: a live residency is in flight (gsh_trk_live_quiesce first)");
        live_refresh_host_state(t);  // (a residency that ended by itself has moved the channels on)
        if (t->h_live_tail != nullptr)
            {
                t->h_live_tail[channel].active = 0;
                __atomic_store_n(&t->h_live_consumed[channel], t->h_live_tail[channel].seq, __ATOMIC_RELEASE);  // records not taken are dropped
            }
        t->h_chan[channel].active = 0;
        // only the flag is written: the rest of the channel's state (device-owned between launches) stays as the loop left it
        GSH_HIP(hipMemcpyAsync(reinterpret_cast<char*>(t->d_chan + channel) + offsetof(gsh::TrkChannel, active), &t->h_chan[channel].active, sizeof(int),
            hipMemcpyHostToDevice, t->stream));
        GSH_HIP(hipStreamSynchronize(t->stream));
        return GSH_OK;
    }

    int gsh_trk_start_ex(gsh_trk_t* t, int channel, const float* code, const float* data_code, int code_length, uint64_t start_sample,
        uint64_t acq_sample_stamp, double acq_carrier_doppler_hz, double initial_acc_carrier_phase_rad)
    {
        return gsh_trk_start_flags(t, channel, code, data_code, code_length, start_sample, acq_sample_stamp, acq_carrier_doppler_hz, initial_acc_carrier_phase_rad, 0U);
    }

    int gsh_trk_start_flags(gsh_trk_t* t, int channel, const float* code, const float* data_code, int code_length, uint64_t start_sample,
        uint64_t acq_sample_stamp, double acq_carrier_doppler_hz, double initial_acc_carrier_phase_rad, uint32_t flags)
    {
        GSH_REQUIRE(t != nullptr && code != nullptr, "null argument");
        GSH_REQUIRE((flags & ~GSH_TRK_START_PULL_IN_OVER) == 0U, "unknown start flags 0x%x", flags);
        GSH_REQUIRE(acq_sample_stamp < (1ull << 63), "acq_sample_stamp out of range");
        GSH_REQUIRE(channel >= 0 && channel < t->n_channels, "channel %d outside 0..%d", channel, t->n_channels - 1);
        GSH_REQUIRE(code_length >= gsh::mcdev::MC_MARGIN && code_length <= t->max_code_len, "code_length %d outside %d..%d", code_length, gsh::mcdev::MC_MARGIN, t->max_code_len);
        GSH_REQUIRE(!t->conf.track_pilot || data_code != nullptr, "track_pilot needs the data-component code");
        // (acq_sample_stamp may lie BEYOND start_sample: a tracking block more than a code period behind the acquisition's stamp starts on an earlier code period,
        //  trk.cc:1949-1978.  The loop's two elapsed-time tests -- pull-in transitory, bit-synchronisation time limit, trk.cc:1912, 2002 -- then see the unsigned
        //  difference wrapped round, here as there: such a channel is declared lost as soon as its C/N0 buffer has filled.  Found with 32 free-running channels.)
        if (t->pending_epochs >= 0) return set_error(GSH_ERR_STATE, "gsh_trk_start: a run has been begun and not ended");
        GSH_HIP(hipSetDevice(t->device));
        if (live_reap(t) != 0) return set_error(GSH_ERR_STATE, "gsh_trk_start: a live residency is in flight (gsh_trk_live_quiesce first)");
        live_refresh_host_state(t);  // (a residency that ended by itself has moved the channels on)
        const gsh_trk_conf& c = t->conf;
        gsh::TrkChannel s{};
        // start_tracking, trk.cc:803-826, and the pull-in state, :1956-1958
        s.carrier_doppler_hz = acq_carrier_doppler_hz;
        s.carrier_phase_step_rad = gsh::GNSS_TWO_PI_D * (s.carrier_doppler_hz + c.cfo_frequency_hz) / c.fs_in;  // trk.cc:801, :1003
        s.code_freq_chips = c.code_chip_rate;
        s.code_phase_step_chips = s.code_freq_chips / c.fs_in;
        s.rem_code_phase_samples = 0.0;
        s.rem_code_phase_chips = 0.0;
        s.acc_carrier_phase_rad = initial_acc_carrier_phase_rad;  // 0, or what the pull-in alignment left (gsh_trk_pull_in, trk.cc:1966)
        s.rem_carr_phase_rad = 0.0F;
        s.p_old_re = 0.0F;
        s.p_old_im = 0.0F;
        s.pos = start_sample;
        // d_pull_in_transitory is a LATCH in the reference (trk.cc:1910-1917: once false it stays false); the kernel re-evaluates pos - acq_stamp < limit every period.
        // A stamp beyond start_sample means the pull-in call's read pointer was below the stamp too (start_sample = nitems_read + offset >= nitems_read): the unsigned
        // difference wrapped there and released the latch at that very call -- without this the wrapped difference would read "over" only until pos passes the stamp
        // and then switch the FLL pull-in and the lock-counter gating back ON (round-5 review).
        if (acq_sample_stamp > start_sample) flags |= GSH_TRK_START_PULL_IN_OVER;
        s.acq_stamp = acq_sample_stamp | ((flags & GSH_TRK_START_PULL_IN_OVER) ? (1ull << 63) : 0ull);
        s.active = 1;
        s.code_len = code_length;
        const double code_period = static_cast<double>(c.code_length_chips) / c.code_chip_rate;
        gsh::design_loop_filter(s.dll, static_cast<float>(code_period), c.dll_bw_hz, c.dll_filter_order, false);     // trk.cc:604, 845-849
        gsh::design_fll_pll(s.pll, c.fll_bw_hz, c.pll_bw_hz, c.pll_filter_order, static_cast<float>(acq_carrier_doppler_hz));  // trk.cc:605, 844, 848
        t->h_chan[channel] = s;
        if (t->h_live_tail != nullptr)
            {
                t->h_live_tail[channel].pos = s.pos;
                t->h_live_tail[channel].active = 1;
                t->h_live_tail[channel].exit_reason = gsh::LIVE_EXIT_IDLE;
                __atomic_store_n(&t->h_live_consumed[channel], t->h_live_tail[channel].seq, __ATOMIC_RELEASE);  // records of the previous run that nobody took are dropped
                t->live_next_window[static_cast<size_t>(channel)] = s.pos;
            }
        float* dst = t->d_codes + static_cast<size_t>(channel) * 2 * t->max_code_len;
        GSH_HIP(hipMemcpyAsync(dst, code, sizeof(float) * code_length, hipMemcpyHostToDevice, t->stream));
        if (data_code) GSH_HIP(hipMemcpyAsync(dst + t->max_code_len, data_code, sizeof(float) * code_length, hipMemcpyHostToDevice, t->stream));
        GSH_HIP(hipMemcpyAsync(t->d_chan + channel, &t->h_chan[channel], sizeof(gsh::TrkChannel), hipMemcpyHostToDevice, t->stream));
        // lock detectors as the constructor / start_tracking leave them (trk.cc:676-692, 1039, 1073, 1970-1971)
        gsh::LockState lk{};
        auto init_smoother = [](gsh::SmootherState& sm, float alpha, int samples, float min_value, float offset) {
            sm = gsh::SmootherState{};
            sm.alpha = alpha < 0.0F ? 0.0F : (alpha > 1.0F ? 1.0F : alpha);  // set_alpha, T/exponential_smoother.cc:28-40
            sm.one_minus_alpha = 1.0F - sm.alpha;
            sm.samples_for_initialization = samples <= 0 ? 1 : samples;     // :49-58
            sm.min_value = min_value;
            sm.offset = offset;
            sm.initializing = 1;
        };
        int cn0_init = 200;  // T/exponential_smoother.h:66
        if (code_period > 0.0) cn0_init = c.cn0_smoother_samples / static_cast<int>(code_period * 1000.0);  // trk.cc:683-686
        init_smoother(lk.cn0_smoother, c.cn0_smoother_alpha, cn0_init, 25.0F, 12.0F);                        // class defaults, T/exponential_smoother.h:64-65
        init_smoother(lk.carrier_lock_test_smoother, c.carrier_lock_test_smoother_alpha, c.carrier_lock_test_smoother_samples, -1.0F, 0.0F);  // trk.cc:688-692
        lk.pull_in_latched = 1;
        lk.pull_in_latched_carr = 1;
        lk.carrier_lock_test = 1.0;  // d_carrier_lock_test(1.0): constructor and clear_tracking_vars (trk.cc:112, 1040)
        if (c.enable_symbol_sync && c.extend_correlation_symbols > 1)
            {
                // what set_update_interval / set_noise_bandwidth / set_params will install when extended integration starts (trk.cc:2126-2129)
                const float t_ext = static_cast<float>(c.extend_correlation_symbols) * static_cast<float>(code_period);
                gsh::LoopFilterState nd;
                gsh::design_loop_filter(nd, t_ext, c.dll_bw_narrow_hz, c.dll_filter_order, false);
                for (int k = 0; k < 4; k++)
                    {
                        lk.dll_narrow_in_c[k] = nd.in_c[k];
                        lk.dll_narrow_out_c[k] = nd.out_c[k];
                    }
                lk.dll_narrow_n_in = nd.n_in;
                lk.dll_narrow_n_out = nd.n_out;
                gsh::design_fll_pll(lk.pll_narrow, c.fll_bw_hz, c.pll_bw_narrow_hz, c.pll_filter_order, 0.0F);
            }
        lk.bs_edge_phase = -1;   // HistogramBitSynchronizer::reset (T/bit_synchronizer.cc:20-38)
        lk.bs_last_sign = +1;
        lk.use_hist = (c.enable_symbol_sync && c.use_histogram_bit_sync && !c.has_secondary && c.symbols_per_bit > 1) ? 1 : 0;  // trk.cc:1389
        lk.state = 2;            // pull-in hands over to state 2 (trk.cc:1963)
        lk.cloop = c.cloop;      // d_cloop = true at start_tracking (trk.cc:1072); conf.cloop lets a caller start four-quadrant
        GSH_HIP(hipMemcpyAsync(t->d_lock + channel, &lk, sizeof(lk), hipMemcpyHostToDevice, t->stream));
        GSH_HIP(hipStreamSynchronize(t->stream));
        return GSH_OK;
    }

    int gsh_trk_run_begin(gsh_trk_t* t, int n_epochs, int want_records)
    {
        GSH_REQUIRE(t != nullptr, "null handle");
        GSH_REQUIRE(n_epochs >= 0, "n_epochs %d", n_epochs);
        if (t->d_stream == nullptr) return set_error(GSH_ERR_STATE, "no IF stream attached (gsh_trk_set_stream_*)");
        if (t->pending_epochs >= 0) return set_error(GSH_ERR_STATE, "gsh_trk_run_begin: the previous run has not been ended");
        GSH_HIP(hipSetDevice(t->device));
        if (live_reap(t) != 0) return set_error(GSH_ERR_STATE, "gsh_trk_run_begin: a live residency is in flight (gsh_trk_live_quiesce first)");
        live_refresh_host_state(t);  // (a residency that ended by itself has moved the channels on)
        const size_t n_rec = want_records ? static_cast<size_t>(t->n_channels) * static_cast<size_t>(n_epochs) : 0;
        // (grown in steps of 64 periods per channel: a caller whose launches lengthen period by period -- the tracking runtime while its channels start --
        // would otherwise free and allocate page-locked memory at every launch, 0.3 ms each time: profiles/ab/r03/dropin_blocks_r03_final.txt)
        const size_t n_cap = static_cast<size_t>(t->n_channels) * ((static_cast<size_t>(n_epochs) + 63) / 64 * 64);
        if (!t->host_records && n_rec > t->records_cap)
            {
                if (t->d_records) GSH_HIP(hipFree(t->d_records));
                t->d_records = nullptr;
                t->records_cap = 0;
                GSH_HIP(hipMalloc(&t->d_records, sizeof(gsh_trk_epoch) * n_cap));
                t->records_cap = n_cap;
            }
        if (n_rec > t->h_records_cap)
            {
                if (t->h_records) GSH_HIP(hipHostFree(t->h_records));
                t->h_records = nullptr;
                t->h_records_cap = 0;
                GSH_HIP(hipHostMalloc(&t->h_records, sizeof(gsh_trk_epoch) * n_cap, hipHostMallocDefault));
                t->h_records_cap = n_cap;
            }
#ifdef GSH_TRACE_TRK_BEGIN
        static std::atomic<long long> acc_ns[2];
        static std::atomic<int> cnt{0};
        const auto t0 = std::chrono::steady_clock::now();
#endif
        gsh_trk_epoch* rec_dst = n_rec > 0 ? t->d_records : nullptr;
        gsh::TrkTail* tail_dst = nullptr;
        if (t->host_records)
            {
                void* p = nullptr;
                if (n_rec > 0)
                    {
                        GSH_HIP(hipHostGetDevicePointer(&p, t->h_records, 0));
                        rec_dst = static_cast<gsh_trk_epoch*>(p);
                    }
                GSH_HIP(hipHostGetDevicePointer(&p, t->h_tail, 0));
                tail_dst = static_cast<gsh::TrkTail*>(p);
            }
        int rc = trk_launch(t, n_epochs, rec_dst, tail_dst);
        if (rc != GSH_OK)
            {
                // the kernel may have been queued before the failure (the ring's reader fence): nothing of this launch may still be writing into the host
                // buffers, or be unknown to the ring, when the caller sees the error -- wait for the stream and take the positions the kernel left
                if (hipStreamSynchronize(t->stream) == hipSuccess && t->host_records)
                    for (int ch = 0; ch < t->n_channels; ch++)
                        if (t->h_tail[ch].done > 0)
                            {
                                t->h_chan[ch].pos = t->h_tail[ch].pos;
                                t->h_chan[ch].active = t->h_tail[ch].active;
                            }
                return rc;
            }
#ifdef GSH_TRACE_TRK_BEGIN
        const auto t1 = std::chrono::steady_clock::now();
#endif
        // what comes back: the records (periods a channel did not run are zeroed on the host in _end, not by a fill kernel in front of the launch) and one
        // 16-byte tail per channel -- written by the kernel itself into the host buffers, or copied
        if (!t->host_records)
            {
                if (n_rec > 0) GSH_HIP(hipMemcpyAsync(t->h_records, t->d_records, sizeof(gsh_trk_epoch) * n_rec, hipMemcpyDeviceToHost, t->stream));
                GSH_HIP(hipMemcpyAsync(t->h_tail, t->d_tail, sizeof(gsh::TrkTail) * t->n_channels, hipMemcpyDeviceToHost, t->stream));
            }
#ifdef GSH_TRACE_TRK_BEGIN
        const auto t2 = std::chrono::steady_clock::now();
        acc_ns[0] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
        acc_ns[1] += std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
        const int k = ++cnt;
        if (k % 40 == 0) std::fprintf(stderr, "gsh_trk_run_begin: launch %.1f us, two copies queued %.1f us (avg of %d)\n", acc_ns[0] * 1e-3 / k, acc_ns[1] * 1e-3 / k, k);
#endif
        t->pending_epochs = n_epochs;
        t->pending_records = n_rec > 0;
        return GSH_OK;
    }

    int gsh_trk_run_end(gsh_trk_t* t, gsh_trk_epoch* records, int32_t* epochs_done)
    {
        GSH_REQUIRE(t != nullptr, "null handle");
        if (t->pending_epochs < 0) return set_error(GSH_ERR_STATE, "gsh_trk_run_end without gsh_trk_run_begin");
        GSH_HIP(hipSetDevice(t->device));
        const int n_epochs = t->pending_epochs;
        t->pending_epochs = -1;
        GSH_HIP(hipStreamSynchronize(t->stream));
        const int rc_coop = coop_check(t);
        const size_t n_rec = static_cast<size_t>(t->n_channels) * static_cast<size_t>(n_epochs);
        if (records != nullptr && n_rec > 0)
            {
                if (!t->pending_records) return set_error(GSH_ERR_STATE, "gsh_trk_run_end: records asked for, but the run was begun without them");
                for (int ch = 0; ch < t->n_channels; ch++)
                    {
                        const int done = std::min(std::max(t->h_tail[ch].done, 0), n_epochs);
                        gsh_trk_epoch* dst = records + static_cast<size_t>(ch) * n_epochs;
                        if (done > 0) std::memcpy(dst, t->h_records + static_cast<size_t>(ch) * n_epochs, sizeof(gsh_trk_epoch) * static_cast<size_t>(done));
                        if (done < n_epochs) std::memset(dst + done, 0, sizeof(gsh_trk_epoch) * static_cast<size_t>(n_epochs - done));  // periods not run: zero records
                    }
            }
        for (int ch = 0; ch < t->n_channels; ch++)
            {
                if (epochs_done != nullptr) epochs_done[ch] = t->h_tail[ch].done;
                t->h_chan[ch].pos = t->h_tail[ch].pos;
                t->h_chan[ch].active = t->h_tail[ch].active;
                if (t->h_live_tail != nullptr)  // a handle that alternates between launches and residencies: the live view follows
                    {
                        t->h_live_tail[ch].pos = t->h_tail[ch].pos;
                        t->h_live_tail[ch].active = t->h_tail[ch].active;
                        if (t->h_live_consumed[ch] == t->h_live_tail[ch].seq) t->live_next_window[static_cast<size_t>(ch)] = t->h_tail[ch].pos;
                    }
            }
        return rc_coop;
    }

    int gsh_trk_set_split(gsh_trk_t* t, int work_groups_per_channel)
    {
        GSH_REQUIRE(t != nullptr, "null handle");
        GSH_REQUIRE(work_groups_per_channel >= 0 && work_groups_per_channel <= 8, "work_groups_per_channel %d outside 0..8", work_groups_per_channel);
        if (t->pending_epochs >= 0) return set_error(GSH_ERR_STATE, "gsh_trk_set_split: a run has been begun and not ended");
        GSH_HIP(hipSetDevice(t->device));
        if (live_reap(t) != 0) return set_error(GSH_ERR_STATE, "gsh_trk_set_split: a live residency is in flight (gsh_trk_live_quiesce first)");
        if (work_groups_per_channel == 0)
            {
                // by the window's length in trips and the compute units the channels leave free (measured, profiles/ab/r06/session17.txt: 7 trips -- 25 000 samples,
                // E/P/L -- 7.27 -> 6.74 us with two and no gain beyond; 62.5 trips -- 128 000 samples, 5 + 1 taps -- 36.5 -> 22.0 / 18.0 / 16.1 us with two / three / four)
                hipDeviceProp_t prop;
                GSH_HIP(hipGetDeviceProperties(&prop, t->device));
                const int trip = (!t->conf.veml && !t->conf.track_pilot) ? 4 * gsh::mcdev::MC_THREADS : 2 * gsh::mcdev::MC_THREADS;
                const int trips = (static_cast<int>(t->conf.vector_length) + trip - 1) / trip;
                const int room = prop.multiProcessorCount / ((t->n_channels + 7) / 8 * 8);
                const int want = trips >= 24 ? 4 : (trips >= 12 ? 3 : (trips >= 6 ? 2 : 1));
                work_groups_per_channel = t->conf.high_dyn ? 1 : std::max(1, std::min(want, room));
            }
        if (work_groups_per_channel > 1)
            {
                GSH_REQUIRE(!t->conf.high_dyn, "cooperating work-groups exist for the standard correlator (high_dyn = 0)");
                hipDeviceProp_t prop;
                GSH_HIP(hipGetDeviceProperties(&prop, t->device));
                const int blocks = (t->n_channels + 7) / 8 * 8 * work_groups_per_channel;
                // every work-group of a launch must be resident at once (they wait for each other): one 1 024-thread work-group per compute unit
                GSH_REQUIRE(blocks <= prop.multiProcessorCount, "%d channels x %d work-groups need %d compute units at once, the device has %d", t->n_channels, work_groups_per_channel,
                    blocks, prop.multiProcessorCount);
            }
        GSH_HIP(hipStreamSynchronize(t->stream));
        if (t->d_coop != nullptr) GSH_HIP(hipFree(t->d_coop));
        t->d_coop = nullptr;
        t->split = work_groups_per_channel;
        if (work_groups_per_channel > 1)
            {
                const size_t words = static_cast<size_t>(t->n_channels) * gsh::coop_stride(work_groups_per_channel) + 1 + 16;  // (+ 16: phase clocks of profiling builds)
                GSH_HIP(hipMalloc(&t->d_coop, words * sizeof(unsigned long long)));
                GSH_HIP(hipMemset(t->d_coop, 0, words * sizeof(unsigned long long)));  // tag 0 is never a period's (coop_seq starts at 1)
            }
        return GSH_OK;
    }

    // ---- live mode ------------------------------------------------------------------------------------------------------------------------------
    int gsh_trk_live_configure(gsh_trk_t* t, uint32_t idle_timeout_us, uint32_t residency_us)
    {
        GSH_REQUIRE(t != nullptr, "null handle");
        GSH_REQUIRE(idle_timeout_us >= 1 && idle_timeout_us <= 1000000u, "idle_timeout_us %u outside 1..1000000", idle_timeout_us);
        GSH_REQUIRE(residency_us >= 10 && residency_us <= 10000000u, "residency_us %u outside 10..10000000", residency_us);
        t->live_idle_us = idle_timeout_us;
        t->live_residency_us = residency_us;
        return GSH_OK;
    }

    int gsh_trk_live_begin(gsh_trk_t* t)
    {
        GSH_REQUIRE(t != nullptr, "null handle");
        if (t->ring == nullptr) return set_error(GSH_ERR_STATE, "live mode follows a sample ring (gsh_trk_set_stream_ring)");
        if (t->pending_epochs >= 0) return set_error(GSH_ERR_STATE, "gsh_trk_live_begin: a run has been begun and not ended");
        GSH_HIP(hipSetDevice(t->device));
        int rc = live_setup(t);
        if (rc != GSH_OK) return rc;
        const int in_flight = live_reap(t);
        if (in_flight < 0) return GSH_ERR_HIP;
        if (in_flight >= 2) return GSH_OK;  // one running, one queued behind it: nothing to add
        const int slot = t->live_busy[0] ? 1 : 0;
        gsh::LiveArgs L{};
        L.head = gsh::stream_live_words(t->ring);
        if (L.head == nullptr) return GSH_ERR_HIP;
        void* p = nullptr;
        GSH_HIP(hipHostGetDevicePointer(&p, t->h_live_tail, 0));
        L.tail = static_cast<gsh::LiveTail*>(p);
        GSH_HIP(hipHostGetDevicePointer(&p, t->h_live_consumed, 0));
        L.consumed = static_cast<const unsigned long long*>(p);
        GSH_HIP(hipHostGetDevicePointer(&p, t->h_live_quit, 0));
        L.quit = static_cast<const int*>(p);
        L.ring_len = t->live_ring_len;
        L.idle_ticks = 100ull * t->live_idle_us;            // wall_clock64() counts at 100 MHz
        L.residency_ticks = 100ull * t->live_residency_us;
        GSH_HIP(hipHostGetDevicePointer(&p, t->h_live_records, 0));
        rc = trk_launch(t, 0x7fffffff, static_cast<gsh_trk_epoch*>(p), nullptr, &L);
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipEventRecord(t->live_ev[slot], t->live_stream));
        t->live_busy[slot] = true;
        return GSH_OK;
    }

    int gsh_trk_live_in_flight(gsh_trk_t* t, int32_t* n)
    {
        GSH_REQUIRE(t != nullptr && n != nullptr, "null argument");
        const int k = live_reap(t);
        if (k < 0) return GSH_ERR_HIP;
        *n = k;
        return GSH_OK;
    }

    int gsh_trk_live_take(gsh_trk_t* t, int channel, uint64_t limit_end, int max_records, gsh_trk_epoch* out, int32_t* n_out, int32_t* pending,
        uint64_t* next_window, int32_t* active, int32_t* resident)
    {
        GSH_REQUIRE(t != nullptr && n_out != nullptr, "null argument");
        GSH_REQUIRE(channel >= 0 && channel < t->n_channels, "channel %d outside 0..%d", channel, t->n_channels - 1);
        GSH_REQUIRE(max_records == 0 || out != nullptr, "null record buffer");
        *n_out = 0;
        if (!t->live_ready.load(std::memory_order_acquire))
            {
                // no residency has ever been queued: nothing is finished; the channel stands where start / the last launch left it
                if (pending != nullptr) *pending = 0;
                if (next_window != nullptr) *next_window = t->h_chan[channel].pos;
                if (active != nullptr) *active = t->h_chan[channel].active;
                if (resident != nullptr) *resident = 0;
                return GSH_OK;
            }
        // No lock, no device call: the channel's tail and record ring are memory the kernel writes and this thread reads.  seq is stored by the device after
        // the records below it have left it (tracking_loop.hip, the drain rule); the loads here follow in program order.
        const volatile gsh::LiveTail* tail = t->h_live_tail + channel;
        const unsigned long long seq = __atomic_load_n(&t->h_live_tail[channel].seq, __ATOMIC_ACQUIRE);
        unsigned long long consumed = t->h_live_consumed[channel];
        const unsigned long long vlen = t->conf.vector_length;
        const gsh_trk_epoch* ring = t->h_live_records + static_cast<size_t>(channel) * t->live_ring_len;
        int n = 0;
        unsigned long long nw = t->live_next_window[static_cast<size_t>(channel)];
        while (n < max_records && consumed < seq)
            {
                const gsh_trk_epoch& r = ring[consumed & (t->live_ring_len - 1u)];
                const bool lost = (r.flags & 2) != 0;
                if (!lost)
                    {
                        const unsigned long long need = std::max<unsigned long long>(vlen, static_cast<unsigned long long>(std::max(r.prn_length_samples, 0)));
                        if (r.sample_counter + need > limit_end) break;  // the caller has not been offered these samples itself yet
                    }
                out[n++] = r;
                consumed++;
                if (lost) break;  // the channel has stopped: this is its last record
                nw = r.sample_counter + static_cast<unsigned long long>(std::max(r.prn_length_samples, 0));
            }
        if (n > 0)
            {
                t->live_next_window[static_cast<size_t>(channel)] = nw;
                __atomic_store_n(&t->h_live_consumed[channel], consumed, __ATOMIC_RELEASE);
            }
        *n_out = n;
        if (pending != nullptr) *pending = static_cast<int32_t>(std::min<unsigned long long>(seq - consumed, 0x7fffffffull));
        if (next_window != nullptr) *next_window = nw;
        if (active != nullptr) *active = tail->active;
        if (resident != nullptr) *resident = (tail->exit_reason == gsh::LIVE_EXIT_NONE) ? 1 : 0;
        return GSH_OK;
    }

    int gsh_trk_live_quiesce(gsh_trk_t* t)
    {
        GSH_REQUIRE(t != nullptr, "null handle");
        GSH_HIP(hipSetDevice(t->device));
        return live_quiesce(t);
    }

    int gsh_trk_run(gsh_trk_t* t, int n_epochs, gsh_trk_epoch* records, int32_t* epochs_done)
    {
        int rc = gsh_trk_run_begin(t, n_epochs, records != nullptr ? 1 : 0);
        if (rc != GSH_OK) return rc;
        return gsh_trk_run_end(t, records, epochs_done);
    }

    int gsh_trk_positions(gsh_trk_t* t, uint64_t* next_window, int32_t* active)
    {
        GSH_REQUIRE(t != nullptr, "null handle");
        for (int ch = 0; ch < t->n_channels; ch++)
            {
                if (next_window != nullptr) next_window[ch] = t->h_chan[ch].pos;
                if (active != nullptr) active[ch] = t->h_chan[ch].active;
            }
        return GSH_OK;
    }

    int gsh_trk_time_run(gsh_trk_t* t, int n_epochs, int reps, float* avg_ms)
    {
        GSH_REQUIRE(t != nullptr && avg_ms != nullptr, "null argument");
        GSH_REQUIRE(n_epochs >= 1 && reps >= 1, "n_epochs %d reps %d", n_epochs, reps);
        if (t->d_stream == nullptr) return set_error(GSH_ERR_STATE, "no IF stream attached (gsh_trk_set_stream_*)");
        GSH_HIP(hipSetDevice(t->device));
        const size_t bytes = sizeof(gsh::TrkChannel) * t->n_channels;
        const size_t lbytes = sizeof(gsh::LockState) * t->n_channels;
        GSH_HIP(hipMemcpyAsync(t->d_chan_backup, t->d_chan, bytes, hipMemcpyDeviceToDevice, t->stream));
        GSH_HIP(hipMemcpyAsync(t->d_lock_backup, t->d_lock, lbytes, hipMemcpyDeviceToDevice, t->stream));
        int rc = trk_launch(t, n_epochs, nullptr);  // warm-up
        if (rc != GSH_OK) return rc;
        float total = 0.0f;
        for (int i = 0; i < reps; i++)
            {
                GSH_HIP(hipMemcpyAsync(t->d_chan, t->d_chan_backup, bytes, hipMemcpyDeviceToDevice, t->stream));
                GSH_HIP(hipMemcpyAsync(t->d_lock, t->d_lock_backup, lbytes, hipMemcpyDeviceToDevice, t->stream));
                GSH_HIP(hipEventRecord(t->ev0, t->stream));
                rc = trk_launch(t, n_epochs, nullptr);
                if (rc != GSH_OK) return rc;
                GSH_HIP(hipEventRecord(t->ev1, t->stream));
                GSH_HIP(hipEventSynchronize(t->ev1));
                float ms = 0.0f;
                GSH_HIP(hipEventElapsedTime(&ms, t->ev0, t->ev1));
                total += ms;
            }
        GSH_HIP(hipMemcpyAsync(t->d_chan, t->d_chan_backup, bytes, hipMemcpyDeviceToDevice, t->stream));
        GSH_HIP(hipMemcpyAsync(t->d_lock, t->d_lock_backup, lbytes, hipMemcpyDeviceToDevice, t->stream));
        GSH_HIP(hipStreamSynchronize(t->stream));
        *avg_ms = total / static_cast<float>(reps);
        return coop_check(t);
    }
}

#ifdef GSH_COOP_PROFILE
// profiling builds only (profiles/ab/r06/coop_phases.py): the eight phase-clock sums channel 0 left behind the launch's error word, then cleared
extern "C" int gsh_debug_coop_profile(gsh_trk_t* t, unsigned long long* out8)
{
    if (t == nullptr || t->d_coop == nullptr || out8 == nullptr) return -1;
    unsigned long long* base = t->d_coop + static_cast<size_t>(t->n_channels) * gsh::coop_stride(t->split);
    if (hipMemcpy(out8, base, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return hipMemset(base + 1, 0, 7 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
