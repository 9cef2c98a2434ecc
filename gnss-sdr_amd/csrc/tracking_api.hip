// C-ABI tracking entry points (include/gnss_sdr_hip.h): the batched correlator bank
// (gsh_bank_*) and the one-to-one Cpu_Multicorrelator_Real_Codes replacement (gsh_mcorr_*)
// built on top of it.  Host-side bookkeeping only; the arithmetic is in multicorrelator.hip.
#include "multicorrelator.h"
#include "sample_stream.h"
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

struct gsh_bank
{
    int device{0};
    hipStream_t stream{nullptr};
    int n_slots{0};
    int max_code_len{0};
    float* d_codes{nullptr};
    int* d_code_lens{nullptr};
    std::vector<int> h_code_lens;
    float2* d_stream_owned{nullptr};
    size_t stream_owned_cap{0};
    const float2* d_stream{nullptr};
    unsigned long long stream_len{0};
    gsh_corr_job* d_jobs{nullptr};
    float2* d_out{nullptr};
    float2* d_partials{nullptr};
    size_t partials_cap{0};
    int jobs_cap{0};
    int n_jobs{0};
    int max_taps{0};
    int mode{0};
    int min_samples{0};
    unsigned long long max_end{0};
    int splits_user{0};
    unsigned long long sample_base{0};  // gsh_bank_set_sample_base
    unsigned long long ring_min_start{0}, ring_max_end{0};  // absolute sample range of the staged batch (ring-bound banks)
    // windowed code staging (multicorrelator.hip): possible when every job of the batch is mode 0 with code_step >= 0
    bool window_eligible{false};
    bool pair{false};                   // the staged batch's 2- / 3-tap jobs are all mcorr_pair_eligible (multicorrelator.h)
    double win_step_max{0.0};   // largest code_phase_step_chips of the batch (code samples per input sample)
    double win_code_span_max{0.0};  // largest code_phase_step_chips * n_samples of the batch: code samples one window walks
    double win_shift_span{0.0}; // largest (max shift - min shift) of the batch
    int max_samples{0};
    // pair fusion (multicorrelator.hip, AUX kernels): a single-tap job that follows a job with the same window and NCO parameters (the data
    // prompt track_pilot adds to a pilot channel, trk.cc:1246-1256) is computed by that job's work-groups instead of a second pass
    int fusion_user{1};
    int n_fused{0};
    int* d_aux{nullptr};
    int aux_cap{0};
    std::vector<int> h_aux;
    // per-flavour launch lists (jobs grouped by tap-count class, fused partners left out); used when a batch mixes classes or has fused jobs
    bool use_classes{false};
    bool lists_fused{false};
    gsh::McorrClassPlan class_plan{};
    int* d_list{nullptr};
    int list_cap{0};
    std::vector<int> h_list;
    hipEvent_t ev0{nullptr}, ev1{nullptr};
    gsh_stream* ring{nullptr};          // when set, job windows are absolute sample indices inside this ring
    gsh_corr_job* h_jobs{nullptr};      // pinned staging (ring translation; one-synchronisation gsh_bank_correlate)
    float2* h_out{nullptr};             // pinned
    int h_cap{0};
};

namespace
{
using gsh::set_error;

int bank_reserve_jobs(gsh_bank* b, int n)
{
    if (n <= b->jobs_cap) return GSH_OK;
    if (b->d_jobs) GSH_HIP(hipFree(b->d_jobs));
    if (b->d_out) GSH_HIP(hipFree(b->d_out));
    b->d_jobs = nullptr;
    b->d_out = nullptr;
    b->jobs_cap = 0;
    GSH_HIP(hipMalloc(&b->d_jobs, sizeof(gsh_corr_job) * static_cast<size_t>(n)));
    GSH_HIP(hipMalloc(&b->d_out, sizeof(float2) * GSH_MAX_TAPS * static_cast<size_t>(n)));
    b->jobs_cap = n;
    return GSH_OK;
}

int validate_job(const gsh_bank* b, const gsh_corr_job& j, int idx);

int bank_reserve_staging(gsh_bank* b, int n)
{
    if (n <= b->h_cap) return GSH_OK;
    if (b->h_jobs) GSH_HIP(hipHostFree(b->h_jobs));
    if (b->h_out) GSH_HIP(hipHostFree(b->h_out));
    b->h_jobs = nullptr;
    b->h_out = nullptr;
    b->h_cap = 0;
    const int cap = std::max(n, 64);
    GSH_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->h_jobs), sizeof(gsh_corr_job) * static_cast<size_t>(cap), hipHostMallocDefault));
    GSH_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->h_out), sizeof(float2) * GSH_MAX_TAPS * static_cast<size_t>(cap), hipHostMallocDefault));
    b->h_cap = cap;
    return GSH_OK;
}

// launch classes: a 3-tap job inside a 5-tap launch would pay for five taps, so a batch that mixes tap counts is launched per kernel flavour;
// with_fusion leaves the fused partners out (their leaders write their rows).  Reads the staged copy of the batch (h_jobs, h_aux).
int bank_build_lists(gsh_bank* b, bool with_fusion)
{
    const int n_jobs = b->n_jobs;
    const gsh_corr_job* jobs = b->h_jobs;
    int count[4] = {0, 0, 0, 0};
    bool has_aux[4] = {false, false, false, false};
    auto cls_of = [](int taps) { return taps <= 1 ? 0 : (taps <= 3 ? 1 : (taps <= 5 ? 2 : 3)); };
    for (int i = 0; i < n_jobs; i++)
        {
            if (with_fusion && b->h_aux[i] == -2) continue;
            const int c = cls_of(jobs[i].n_taps);
            count[c]++;
            if (with_fusion && b->h_aux[i] >= 0) has_aux[c] = true;
        }
    const int classes = (count[0] > 0) + (count[1] > 0) + (count[2] > 0) + (count[3] > 0);
    b->use_classes = (classes > 1) || with_fusion;
    b->lists_fused = with_fusion;
    if (!b->use_classes) return GSH_OK;
    int off[4];
    off[0] = 0;
    for (int c = 1; c < 4; c++) off[c] = off[c - 1] + count[c - 1];
    b->h_list.assign(static_cast<size_t>(n_jobs), 0);
    int fill[4] = {off[0], off[1], off[2], off[3]};
    for (int i = 0; i < n_jobs; i++)
        {
            if (with_fusion && b->h_aux[i] == -2) continue;
            b->h_list[static_cast<size_t>(fill[cls_of(jobs[i].n_taps)]++)] = i;
        }
    if (b->list_cap < n_jobs)
        {
            if (b->d_list) GSH_HIP(hipFree(b->d_list));
            b->d_list = nullptr;
            b->list_cap = 0;
            GSH_HIP(hipMalloc(&b->d_list, sizeof(int) * static_cast<size_t>(n_jobs)));
            b->list_cap = n_jobs;
        }
    GSH_HIP(hipMemcpyAsync(b->d_list, b->h_list.data(), sizeof(int) * static_cast<size_t>(n_jobs), hipMemcpyHostToDevice, b->stream));
    b->class_plan.list = b->d_list;
    for (int c = 0; c < 4; c++)
        {
            b->class_plan.offset[c] = off[c];
            b->class_plan.count[c] = count[c];
            b->class_plan.aux[c] = has_aux[c];
        }
    return GSH_OK;
}

// validate the batch, stage it in pinned memory (ring mode: absolute sample indices -> ring positions) and queue the
// host-to-device copy on the bank's stream.  No synchronisation.
int bank_stage_jobs(gsh_bank* b, const gsh_corr_job* jobs, int n_jobs)
{
    int max_taps = 0, mode = jobs[0].high_dyn, min_samples = jobs[0].n_samples, max_samples = 0;
    unsigned long long max_end = 0;
    bool window_eligible = (mode == 0);
    bool pair = true;  // every job the 3-tap launch will see may read its early tap next to the late one (multicorrelator.h mcorr_pair_eligible)
    double step_max = 0.0, shift_span = 0.0, code_span = 0.0;
    for (int i = 0; i < n_jobs; i++)
        {
            int rc = validate_job(b, jobs[i], i);
            if (rc != GSH_OK) return rc;
            if ((jobs[i].n_taps == 2 || jobs[i].n_taps == 3) && !gsh::mcorr_pair_eligible(jobs[i].n_taps, jobs[i].shifts_chips, jobs[i].code_phase_step_chips, jobs[i].high_dyn))
                pair = false;
            max_samples = std::max(max_samples, jobs[i].n_samples);
            if (!(jobs[i].code_phase_step_chips >= 0.0f)) window_eligible = false;
            step_max = std::max(step_max, static_cast<double>(jobs[i].code_phase_step_chips));
            code_span = std::max(code_span, static_cast<double>(jobs[i].code_phase_step_chips) * static_cast<double>(jobs[i].n_samples));
            float smin = jobs[i].shifts_chips[0], smax = jobs[i].shifts_chips[0];
            for (int t = 1; t < jobs[i].n_taps; t++)
                {
                    smin = std::min(smin, jobs[i].shifts_chips[t]);
                    smax = std::max(smax, jobs[i].shifts_chips[t]);
                }
            shift_span = std::max(shift_span, static_cast<double>(smax) - static_cast<double>(smin));
            if (jobs[i].high_dyn != mode)
                return set_error(GSH_ERR_UNSUPPORTED, "job %d: all jobs of one batch must share high_dyn (%d vs %d); split the batch", i, jobs[i].high_dyn, mode);
            max_taps = std::max(max_taps, jobs[i].n_taps);
            min_samples = std::min(min_samples, jobs[i].n_samples);
            max_end = std::max(max_end, static_cast<unsigned long long>(jobs[i].sample_offset) + static_cast<unsigned long long>(jobs[i].n_samples));
        }
    GSH_HIP(hipSetDevice(b->device));
    int rc = bank_reserve_jobs(b, n_jobs);
    if (rc == GSH_OK) rc = bank_reserve_staging(b, n_jobs);
    if (rc != GSH_OK) return rc;
    std::memcpy(b->h_jobs, jobs, sizeof(gsh_corr_job) * static_cast<size_t>(n_jobs));
    if (b->ring != nullptr)
        {
            unsigned long long lo = ~0ull;
            for (int i = 0; i < n_jobs; i++) lo = std::min<unsigned long long>(lo, jobs[i].sample_offset);
            b->ring_min_start = n_jobs > 0 ? lo : 0ull;
            b->ring_max_end = max_end;
            for (int i = 0; i < n_jobs; i++)
                {
                    const float2* w = nullptr;
                    rc = gsh::stream_window(b->ring, jobs[i].sample_offset, static_cast<unsigned long long>(jobs[i].n_samples), &w);
                    if (rc != GSH_OK) return rc;
                    b->h_jobs[i].sample_offset = static_cast<uint64_t>(w - b->ring->d_ring);
                }
            b->d_stream = b->ring->d_ring;
            b->stream_len = b->ring->capacity + b->ring->max_window;
            max_end = 0;  // residency was checked per job
        }
    GSH_HIP(hipMemcpyAsync(b->d_jobs, b->h_jobs, sizeof(gsh_corr_job) * static_cast<size_t>(n_jobs), hipMemcpyHostToDevice, b->stream));
    // ---- pair fusion: job i + 1 rides on job i when it is a single tap over exactly the same samples with exactly the same NCO parameters
    b->n_fused = 0;
    if (b->fusion_user && mode == 0 && max_taps >= 2 && max_taps <= 5)
        {
            b->h_aux.assign(static_cast<size_t>(n_jobs), -1);
            for (int i = 0; i + 1 < n_jobs; i++)
                {
                    const gsh_corr_job &p = jobs[i], &q = jobs[i + 1];
                    if (p.n_taps < 2 || q.n_taps != 1 || b->h_aux[i] == -2) continue;
                    if (p.sample_offset != q.sample_offset || p.n_samples != q.n_samples) continue;
                    if (std::memcmp(&p.rem_carr_phase_rad, &q.rem_carr_phase_rad, sizeof(float)) != 0 || std::memcmp(&p.phase_step_rad, &q.phase_step_rad, sizeof(float)) != 0 ||
                        std::memcmp(&p.rem_code_phase_chips, &q.rem_code_phase_chips, sizeof(float)) != 0 ||
                        std::memcmp(&p.code_phase_step_chips, &q.code_phase_step_chips, sizeof(float)) != 0)
                        continue;
                    b->h_aux[i] = i + 1;
                    b->h_aux[i + 1] = -2;
                    b->n_fused++;
                    i++;
                }
            if (b->n_fused > 0)
                {
                    if (b->aux_cap < n_jobs)
                        {
                            if (b->d_aux) GSH_HIP(hipFree(b->d_aux));
                            b->d_aux = nullptr;
                            b->aux_cap = 0;
                            GSH_HIP(hipMalloc(&b->d_aux, sizeof(int) * static_cast<size_t>(n_jobs)));
                            b->aux_cap = n_jobs;
                        }
                    GSH_HIP(hipMemcpyAsync(b->d_aux, b->h_aux.data(), sizeof(int) * static_cast<size_t>(n_jobs), hipMemcpyHostToDevice, b->stream));
                }
        }
    b->n_jobs = n_jobs;
    {
        int rc2 = bank_build_lists(b, b->n_fused > 0);
        if (rc2 != GSH_OK)
            {
                b->n_jobs = 0;
                return rc2;
            }
    }
    b->n_jobs = n_jobs;
    b->max_taps = max_taps;
    b->mode = mode;
    b->min_samples = min_samples;
    b->max_end = max_end;
    b->max_samples = max_samples;
    b->window_eligible = window_eligible;
    b->pair = pair;
    b->win_step_max = step_max;
    b->win_code_span_max = code_span;
    b->win_shift_span = shift_span;
    return GSH_OK;
}

int bank_splits(const gsh_bank* b)
{
    int s = b->splits_user;
    if (s <= 0)
        {
            // throughput mode once there are a few work-groups per CU; otherwise spread each
            // epoch over several CUs (latency mode for closed-loop tracking)
            s = (b->n_jobs >= 1024) ? 1 : (2048 + b->n_jobs - 1) / std::max(b->n_jobs, 1);
        }
    const int by_len = std::max(1, b->min_samples / 2048);  // keep >= 2048 samples per work-group
    if (b->splits_user <= 0 && b->window_eligible && static_cast<size_t>(b->max_code_len) * sizeof(float) > 20 * 1024)
        {
            // long codes (Galileo E1 / E5, GPS L5 / L2C): the whole code in LDS leaves room for 3-4 work-groups per compute unit.  Cut the
            // windows so that a work-group only walks ~2000 code samples and stages just those (bank_window_floats).
            const int want = static_cast<int>(std::ceil(b->win_code_span_max / 2000.0));
            s = std::max(s, std::min(want, 16));
        }
    s = std::min(s, by_len);
    s = std::min(s, 64);
    return std::max(s, 1);
}

// LDS floats per work-group when only the code samples a segment can touch are staged; 0 when the whole code is staged instead
// (batch not eligible, or the window would not be clearly smaller than the code)
int bank_window_floats(const gsh_bank* b, int splits)
{
    if (!b->window_eligible) return 0;
    // hi - lo + 1 <= step * (seg - 1) + (smax - smin) + 2 in exact arithmetic, seg the job's own segment length; the float evaluation on
    // the device moves either end by a few ulps of values below 2^24, far less than the slack added here (the kernel re-checks and
    // reports NaN if this were ever short)
    double walk = 0.0;
    for (int i = 0; i < b->n_jobs; i++)
        {
            int seg = (b->h_jobs[i].n_samples + splits - 1) / splits;
            seg = (seg + 1) & ~1;
            walk = std::max(walk, static_cast<double>(b->h_jobs[i].code_phase_step_chips) * static_cast<double>(seg));
        }
    const double need = walk + b->win_shift_span + 16.0;
    if (!(need < 1.0e6)) return 0;
    const int w = (static_cast<int>(std::ceil(need)) + 3) & ~3;
    const int full = b->max_code_len + 2 * 32;
    return (2 * w <= full) ? w : 0;
}

int validate_job(const gsh_bank* b, const gsh_corr_job& j, int idx)
{
    if (j.n_taps < 1 || j.n_taps > GSH_MAX_TAPS) return set_error(GSH_ERR_INVALID, "job %d: n_taps %d outside 1..%d", idx, j.n_taps, GSH_MAX_TAPS);
    if (j.n_samples < 1 || j.n_samples > (1 << 30)) return set_error(GSH_ERR_INVALID, "job %d: n_samples %d", idx, j.n_samples);
    if (j.code_slot < 0 || j.code_slot >= b->n_slots || b->h_code_lens[j.code_slot] <= 0)
        return set_error(GSH_ERR_STATE, "job %d: code slot %d has no local code", idx, j.code_slot);
    if (j.high_dyn < 0 || j.high_dyn > 2) return set_error(GSH_ERR_INVALID, "job %d: high_dyn %d", idx, j.high_dyn);
    if (j.high_dyn != 0)
        {
            // the reference derives taps 1..T-1 by rotating tap 0 by round(dshift/step) samples
            // (K/..high_dynamics_resampler..:82-90); a rotation outside [0, n] is undefined there.
            unsigned acc = 0;
            for (int t = 1; t < j.n_taps; t++)
                {
                    const float q = (j.shifts_chips[t] - j.shifts_chips[t - 1]) / j.code_phase_step_chips;
                    if (!std::isfinite(q)) return set_error(GSH_ERR_INVALID, "job %d: high-dynamics tap spacing / code step is not finite", idx);
                    acc += static_cast<unsigned>(static_cast<int>(std::round(q)));
                    if (acc > static_cast<unsigned>(j.n_samples))
                        return set_error(GSH_ERR_INVALID, "job %d: high-dynamics tap rotation %u outside [0, %d] (taps must ascend)", idx, acc, j.n_samples);
                }
        }
    return GSH_OK;
}
}  // namespace

extern "C"
{
    int gsh_bank_create(int device, int n_code_slots, int max_code_length, gsh_bank_t** out)
    {
        GSH_REQUIRE(out != nullptr, "null out pointer");
        *out = nullptr;
        GSH_REQUIRE(n_code_slots >= 1 && n_code_slots <= 65536, "n_code_slots %d", n_code_slots);
        GSH_REQUIRE(max_code_length >= 1, "max_code_length %d", max_code_length);
        GSH_REQUIRE(gsh::mcorr_lds_bytes(max_code_length) <= 64 * 1024, "max_code_length %d does not fit the LDS code table (limit %d samples)", max_code_length, 64 * 1024 / 4 - 128);
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh_bank* b = new (std::nothrow) gsh_bank();
        GSH_REQUIRE(b != nullptr, "out of host memory");
        b->device = device;
        b->n_slots = n_code_slots;
        b->max_code_len = max_code_length;
        b->h_code_lens.assign(n_code_slots, 0);
        auto fail = [&](hipError_t e, const char* what) {
            gsh::hip_fail(e, what, __FILE__, __LINE__);
            gsh_bank_destroy(b);
            return GSH_ERR_HIP;
        };
        hipError_t e;
        if ((e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
        if ((e = hipMalloc(&b->d_codes, sizeof(float) * static_cast<size_t>(n_code_slots) * max_code_length)) != hipSuccess) return fail(e, "hipMalloc(codes)");
        if ((e = hipMalloc(&b->d_code_lens, sizeof(int) * n_code_slots)) != hipSuccess) return fail(e, "hipMalloc(code_lens)");
        if ((e = hipMemset(b->d_code_lens, 0, sizeof(int) * n_code_slots)) != hipSuccess) return fail(e, "hipMemset");
        if ((e = hipEventCreate(&b->ev0)) != hipSuccess) return fail(e, "hipEventCreate");
        if ((e = hipEventCreate(&b->ev1)) != hipSuccess) return fail(e, "hipEventCreate");
        *out = b;
        return GSH_OK;
    }

    void gsh_bank_destroy(gsh_bank_t* b)
    {
        if (!b) return;
        (void)hipSetDevice(b->device);
        if (b->stream) (void)hipStreamSynchronize(b->stream);
        if (b->d_codes) (void)hipFree(b->d_codes);
        if (b->d_code_lens) (void)hipFree(b->d_code_lens);
        if (b->d_stream_owned) (void)hipFree(b->d_stream_owned);
        if (b->d_jobs) (void)hipFree(b->d_jobs);
        if (b->d_out) (void)hipFree(b->d_out);
        if (b->d_partials) (void)hipFree(b->d_partials);
        if (b->d_aux) (void)hipFree(b->d_aux);
        if (b->d_list) (void)hipFree(b->d_list);
        if (b->ev0) (void)hipEventDestroy(b->ev0);
        if (b->ev1) (void)hipEventDestroy(b->ev1);
        if (b->h_jobs) (void)hipHostFree(b->h_jobs);
        if (b->h_out) (void)hipHostFree(b->h_out);
        if (b->stream) (void)hipStreamDestroy(b->stream);
        delete b;
    }

    int gsh_bank_set_code(gsh_bank_t* b, int slot, const float* code, int code_length)
    {
        GSH_REQUIRE(b && code, "null argument");
        GSH_REQUIRE(slot >= 0 && slot < b->n_slots, "slot %d outside 0..%d", slot, b->n_slots - 1);
        GSH_REQUIRE(code_length >= 1 && code_length <= b->max_code_len, "code_length %d outside 1..%d", code_length, b->max_code_len);
        GSH_HIP(hipSetDevice(b->device));
        GSH_HIP(hipStreamSynchronize(b->stream));
        GSH_HIP(hipMemcpy(b->d_codes + static_cast<size_t>(slot) * b->max_code_len, code, sizeof(float) * code_length, hipMemcpyHostToDevice));
        GSH_HIP(hipMemcpy(b->d_code_lens + slot, &code_length, sizeof(int), hipMemcpyHostToDevice));
        b->h_code_lens[slot] = code_length;
        return GSH_OK;
    }

    int gsh_bank_set_stream_host(gsh_bank_t* b, const float* iq, uint64_t n_samples)
    {
        GSH_REQUIRE(b && iq, "null argument");
        GSH_REQUIRE(n_samples >= 1, "empty stream");
        GSH_HIP(hipSetDevice(b->device));
        GSH_HIP(hipStreamSynchronize(b->stream));
        const size_t need = static_cast<size_t>(n_samples) + 2;  // one spare pair: 16-byte loads may touch sample n
        if (need > b->stream_owned_cap)
            {
                if (b->d_stream_owned) GSH_HIP(hipFree(b->d_stream_owned));
                b->d_stream_owned = nullptr;
                b->stream_owned_cap = 0;
                GSH_HIP(hipMalloc(&b->d_stream_owned, sizeof(float2) * need));
                b->stream_owned_cap = need;
            }
        GSH_HIP(hipMemcpy(b->d_stream_owned, iq, sizeof(float2) * n_samples, hipMemcpyHostToDevice));
        GSH_HIP(hipMemset(b->d_stream_owned + n_samples, 0, sizeof(float2) * 2));
        b->d_stream = b->d_stream_owned;
        b->stream_len = n_samples;
        b->ring = nullptr;
        return GSH_OK;
    }

    int gsh_bank_set_stream_device(gsh_bank_t* b, const void* device_iq, uint64_t n_samples)
    {
        GSH_REQUIRE(b && device_iq, "null argument");
        GSH_REQUIRE(n_samples >= 1, "empty stream");
        GSH_REQUIRE((reinterpret_cast<uintptr_t>(device_iq) & 15u) == 0, "device stream must be 16-byte aligned");
        b->d_stream = static_cast<const float2*>(device_iq);
        b->stream_len = n_samples;
        b->ring = nullptr;
        return GSH_OK;
    }

    int gsh_bank_set_splits(gsh_bank_t* b, int splits)
    {
        GSH_REQUIRE(b != nullptr, "null bank");
        GSH_REQUIRE(splits >= 0 && splits <= 64, "splits %d outside 0..64", splits);
        b->splits_user = splits;
        return GSH_OK;
    }

    int gsh_bank_set_sample_base(gsh_bank_t* b, uint64_t sample_base)
    {
        GSH_REQUIRE(b != nullptr, "null bank");
        b->sample_base = sample_base;
        return GSH_OK;
    }

    int gsh_bank_set_pair_fusion(gsh_bank_t* b, int enable)
    {
        GSH_REQUIRE(b != nullptr, "null bank");
        b->fusion_user = enable ? 1 : 0;
        b->n_jobs = 0;  // the staged batch was analysed under the previous setting
        return GSH_OK;
    }

    int gsh_bank_upload_jobs(gsh_bank_t* b, const gsh_corr_job* jobs, int n_jobs)
    {
        GSH_REQUIRE(b != nullptr, "null bank");
        GSH_REQUIRE(n_jobs >= 0, "n_jobs %d", n_jobs);
        GSH_REQUIRE(n_jobs == 0 || jobs != nullptr, "null jobs");
        b->n_jobs = 0;
        if (n_jobs == 0) return GSH_OK;
        int rc = bank_stage_jobs(b, jobs, n_jobs);
        if (rc != GSH_OK)
            {
                b->n_jobs = 0;
                return rc;
            }
        GSH_HIP(hipStreamSynchronize(b->stream));
        return GSH_OK;
    }

    int gsh_bank_set_stream_ring(gsh_bank_t* b, gsh_stream_t* s)
    {
        GSH_REQUIRE(b != nullptr, "null bank");
        GSH_REQUIRE(s == nullptr || s->device == b->device, "the ring lives on device %d, the bank on device %d", s ? s->device : -1, b->device);
        b->ring = s;
        b->n_jobs = 0;  // an uploaded job table was translated against the previous attachment
        if (s != nullptr)
            {
                b->d_stream = s->d_ring;
                b->stream_len = s->capacity + s->max_window;
            }
        else
            {
                b->d_stream = nullptr;
                b->stream_len = 0;
            }
        return GSH_OK;
    }

    int gsh_bank_launch(gsh_bank_t* b, void* hip_stream)
    {
        GSH_REQUIRE(b != nullptr, "null bank");
        if (b->n_jobs == 0) return GSH_OK;
        if (b->d_stream == nullptr) return set_error(GSH_ERR_STATE, "no sample stream attached (gsh_bank_set_stream_*)");
        if (b->ring == nullptr)
            {
                if (b->max_end + b->sample_base > b->stream_len)  // sample_base shifts every window (gsh_bank_set_sample_base)
                    return set_error(GSH_ERR_INVALID, "a job window ends at sample %llu (sample_base %llu), past the %llu-sample stream", b->max_end + b->sample_base,
                        b->sample_base, b->stream_len);
            }
        else if (b->ring_max_end != 0)
            {
                // the shifted windows must be resident and pushed: a base that points at samples the ring no longer (or not yet) holds would make
                // the kernel read whatever lives at those ring positions now
                const unsigned long long lo = b->ring_min_start + b->sample_base, hi = b->ring_max_end + b->sample_base;
                if (lo < gsh::stream_oldest(b->ring) || hi > b->ring->next)
                    return set_error(GSH_ERR_INVALID, "windows [%llu, %llu) (sample_base %llu) are not resident (ring holds [%llu, %llu))", lo, hi, b->sample_base,
                        gsh::stream_oldest(b->ring), static_cast<unsigned long long>(b->ring->next));
            }
        GSH_HIP(hipSetDevice(b->device));
        const int splits = bank_splits(b);
        if (splits > 1)
            {
                const size_t need = static_cast<size_t>(b->n_jobs) * splits * GSH_MAX_TAPS;
                if (need > b->partials_cap)
                    {
                        if (b->d_partials) GSH_HIP(hipFree(b->d_partials));
                        b->d_partials = nullptr;
                        b->partials_cap = 0;
                        GSH_HIP(hipMalloc(&b->d_partials, sizeof(float2) * need));
                        b->partials_cap = need;
                    }
            }
        gsh::McorrArgs a;
        a.stream = b->d_stream;
        a.stream_len = b->stream_len;
        a.jobs = b->d_jobs;
        a.codes = b->d_codes;
        a.code_lens = b->d_code_lens;
        a.code_stride = b->max_code_len;
        a.out = b->d_out;
        a.partials = b->d_partials;
        a.n_jobs = b->n_jobs;
        a.splits = splits;
        a.window_floats = bank_window_floats(b, splits);
        a.packed = gsh::mcorr_packed_default();
        a.fac = gsh::mcorr_fac_default();
        a.pair = b->pair ? 1 : 0;
        a.sample_base = b->sample_base;
        a.ring_capacity = b->ring != nullptr ? b->ring->capacity : 0ull;
        // the second code table must not cost the occupancy the fusion is meant to win: only with windowed tables or short codes
        const bool fuse = b->n_fused > 0 && (a.window_floats > 0 || static_cast<size_t>(b->max_code_len) * sizeof(float) <= 10 * 1024);
        a.aux = fuse ? b->d_aux : nullptr;
        hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : b->stream;
        if (b->ring != nullptr)
            {
                // wait for the push that completed this batch's newest window -- not for a later block that may already be on its way
                int rcw = gsh::stream_wait_pushed(b->ring, b->ring_max_end ? b->ring_max_end + b->sample_base : ~0ull, s);
                if (rcw != GSH_OK) return rcw;
            }
        a.job_list = nullptr;
        a.n_launch = b->n_jobs;
        if (b->lists_fused != fuse)
            {
                // the LDS rule above changed its verdict since the batch was staged (set_splits in between): regroup
                int rc2 = bank_build_lists(b, fuse);
                if (rc2 != GSH_OK) return rc2;
                GSH_HIP(hipStreamSynchronize(b->stream));  // the list upload was queued on the bank's stream
            }
        const int rc_launch = b->use_classes ? gsh::mcorr_launch_classes(a, b->class_plan, b->mode, b->max_code_len, s)
                                             : gsh::mcorr_launch(a, b->max_taps, b->mode, b->max_code_len, s);
        if (rc_launch != GSH_OK) return rc_launch;
        if (b->ring != nullptr) return gsh::stream_mark_read(b->ring, b->ring_min_start + b->sample_base, s);  // pushes that would overwrite these windows wait
        return GSH_OK;
    }

    int gsh_bank_synchronize(gsh_bank_t* b)
    {
        GSH_REQUIRE(b != nullptr, "null bank");
        GSH_HIP(hipSetDevice(b->device));
        GSH_HIP(hipStreamSynchronize(b->stream));
        return GSH_OK;
    }

    int gsh_bank_read_outputs(gsh_bank_t* b, float* out_iq, int n_jobs)
    {
        GSH_REQUIRE(b && out_iq, "null argument");
        GSH_REQUIRE(n_jobs >= 0 && n_jobs <= b->n_jobs, "n_jobs %d outside 0..%d", n_jobs, b->n_jobs);
        if (n_jobs == 0) return GSH_OK;
        GSH_HIP(hipSetDevice(b->device));
        GSH_HIP(hipMemcpyAsync(out_iq, b->d_out, sizeof(float2) * GSH_MAX_TAPS * static_cast<size_t>(n_jobs), hipMemcpyDeviceToHost, b->stream));
        GSH_HIP(hipStreamSynchronize(b->stream));
        return GSH_OK;
    }

    int gsh_bank_correlate(gsh_bank_t* b, const gsh_corr_job* jobs, int n_jobs, float* out_iq)
    {
        // one synchronisation per batch: pinned job table -> H2D -> kernel -> D2H into pinned memory, all queued on the bank's stream
        GSH_REQUIRE(b != nullptr, "null bank");
        GSH_REQUIRE(n_jobs >= 0, "n_jobs %d", n_jobs);
        GSH_REQUIRE(n_jobs == 0 || (jobs != nullptr && out_iq != nullptr), "null argument");
        b->n_jobs = 0;
        if (n_jobs == 0) return GSH_OK;
        int rc = bank_stage_jobs(b, jobs, n_jobs);
        if (rc == GSH_OK) rc = gsh_bank_launch(b, nullptr);
        if (rc != GSH_OK)
            {
                (void)hipStreamSynchronize(b->stream);
                b->n_jobs = 0;
                return rc;
            }
        const size_t bytes = sizeof(float2) * GSH_MAX_TAPS * static_cast<size_t>(n_jobs);
        GSH_HIP(hipMemcpyAsync(b->h_out, b->d_out, bytes, hipMemcpyDeviceToHost, b->stream));
        GSH_HIP(hipStreamSynchronize(b->stream));
        std::memcpy(out_iq, b->h_out, bytes);
        return GSH_OK;
    }

    int gsh_bank_time_launches(gsh_bank_t* b, int reps, float* avg_ms)
    {
        GSH_REQUIRE(b && avg_ms, "null argument");
        GSH_REQUIRE(reps >= 1, "reps %d", reps);
        GSH_HIP(hipSetDevice(b->device));
        int rc = gsh_bank_launch(b, nullptr);  // warm-up, also validates state
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipEventRecord(b->ev0, b->stream));
        for (int i = 0; i < reps; i++)
            {
                rc = gsh_bank_launch(b, nullptr);
                if (rc != GSH_OK) return rc;
            }
        GSH_HIP(hipEventRecord(b->ev1, b->stream));
        GSH_HIP(hipEventSynchronize(b->ev1));
        float ms = 0.0f;
        GSH_HIP(hipEventElapsedTime(&ms, b->ev0, b->ev1));
        *avg_ms = ms / static_cast<float>(reps);
        return GSH_OK;
    }
}

// ------------------------------------------------------------------------------------------
// gsh_mcorr_*: Cpu_Multicorrelator_Real_Codes replacement (mcorr.h:37-61)
// ------------------------------------------------------------------------------------------
struct gsh_mcorr
{
    int device{0};
    gsh_bank* bank{nullptr};
    int max_len{0};
    int n_correlators{0};
    int code_len{0};
    float* shifts{nullptr};       // borrowed, re-read every call (mcorr.cc:58)
    const float* sig_in{nullptr}; // borrowed (mcorr.cc:69)
    float* corr_out{nullptr};     // borrowed (mcorr.cc:70)
    bool high_dyn{true};          // mcorr.h:60 default
    float2* h_pinned_in{nullptr};
    float2* h_pinned_out{nullptr};
    float2* d_in{nullptr};
};

namespace
{
int mcorr_run(gsh_mcorr* h, int mode, float rem_carr, float phase_step, float phase_rate, float rem_code, float code_step, float code_rate, int n)
{
    GSH_REQUIRE(h != nullptr, "null handle");
    if (h->bank == nullptr || h->d_in == nullptr) return set_error(GSH_ERR_STATE, "init() has not been called");
    if (h->code_len <= 0 || h->shifts == nullptr) return set_error(GSH_ERR_STATE, "set_local_code_and_taps() has not been called");
    if (h->sig_in == nullptr || h->corr_out == nullptr) return set_error(GSH_ERR_STATE, "set_input_output_vectors() has not been called");
    GSH_REQUIRE(n >= 1 && n <= h->max_len, "signal_length_samples %d outside 1..%d (init size)", n, h->max_len);
    gsh_bank* b = h->bank;
    GSH_HIP(hipSetDevice(h->device));

    gsh_corr_job j;
    std::memset(&j, 0, sizeof(j));
    j.sample_offset = 0;
    j.n_samples = n;
    j.code_slot = 0;
    j.rem_carr_phase_rad = rem_carr;
    j.phase_step_rad = phase_step;
    j.phase_rate_step_rad = phase_rate;
    j.rem_code_phase_chips = rem_code;
    j.code_phase_step_chips = code_step;
    j.code_phase_rate_step_chips = code_rate;
    j.n_taps = h->n_correlators;
    j.high_dyn = mode;
    for (int t = 0; t < h->n_correlators; t++) j.shifts_chips[t] = h->shifts[t];

    // H2D of the epoch through pinned staging, one launch, D2H of T complex values
    std::memcpy(h->h_pinned_in, h->sig_in, sizeof(float2) * static_cast<size_t>(n));
    GSH_HIP(hipMemcpyAsync(h->d_in, h->h_pinned_in, sizeof(float2) * static_cast<size_t>(n), hipMemcpyHostToDevice, b->stream));
    b->d_stream = h->d_in;
    b->stream_len = static_cast<unsigned long long>(n);
    // one synchronisation per call: pinned job record -> H2D, kernel and D2H are all queued on the bank's stream
    int rc = bank_stage_jobs(b, &j, 1);
    if (rc == GSH_OK) rc = gsh_bank_launch(b, nullptr);
    if (rc != GSH_OK)
        {
            (void)hipStreamSynchronize(b->stream);
            b->n_jobs = 0;
            return rc;
        }
    GSH_HIP(hipMemcpyAsync(h->h_pinned_out, b->d_out, sizeof(float2) * GSH_MAX_TAPS, hipMemcpyDeviceToHost, b->stream));
    GSH_HIP(hipStreamSynchronize(b->stream));
    std::memcpy(h->corr_out, h->h_pinned_out, sizeof(float2) * static_cast<size_t>(h->n_correlators));
    return GSH_OK;
}
}  // namespace

extern "C"
{
    int gsh_mcorr_create(int device, gsh_mcorr_t** out)
    {
        GSH_REQUIRE(out != nullptr, "null out pointer");
        *out = nullptr;
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh_mcorr* h = new (std::nothrow) gsh_mcorr();
        GSH_REQUIRE(h != nullptr, "out of host memory");
        h->device = device;
        *out = h;
        return GSH_OK;
    }

    int gsh_mcorr_free(gsh_mcorr_t* h)
    {
        GSH_REQUIRE(h != nullptr, "null handle");
        (void)hipSetDevice(h->device);
        if (h->bank) gsh_bank_destroy(h->bank);
        h->bank = nullptr;
        if (h->h_pinned_in) (void)hipHostFree(h->h_pinned_in);
        if (h->h_pinned_out) (void)hipHostFree(h->h_pinned_out);
        if (h->d_in) (void)hipFree(h->d_in);
        h->h_pinned_in = nullptr;
        h->h_pinned_out = nullptr;
        h->d_in = nullptr;
        h->max_len = 0;
        h->code_len = 0;
        return GSH_OK;
    }

    void gsh_mcorr_destroy(gsh_mcorr_t* h)
    {
        if (!h) return;
        gsh_mcorr_free(h);
        delete h;
    }

    int gsh_mcorr_init(gsh_mcorr_t* h, int max_signal_length_samples, int n_correlators)
    {
        GSH_REQUIRE(h != nullptr, "null handle");
        GSH_REQUIRE(max_signal_length_samples >= 1, "max_signal_length_samples %d", max_signal_length_samples);
        GSH_REQUIRE(n_correlators >= 1 && n_correlators <= GSH_MAX_TAPS, "n_correlators %d outside 1..%d", n_correlators, GSH_MAX_TAPS);
        gsh_mcorr_free(h);
        GSH_HIP(hipSetDevice(h->device));
        h->max_len = max_signal_length_samples;
        h->n_correlators = n_correlators;
        GSH_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_pinned_in), sizeof(float2) * static_cast<size_t>(max_signal_length_samples), hipHostMallocDefault));
        GSH_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_pinned_out), sizeof(float2) * GSH_MAX_TAPS, hipHostMallocDefault));
        GSH_HIP(hipMalloc(&h->d_in, sizeof(float2) * (static_cast<size_t>(max_signal_length_samples) + 2)));
        GSH_HIP(hipMemset(h->d_in, 0, sizeof(float2) * (static_cast<size_t>(max_signal_length_samples) + 2)));
        return GSH_OK;
    }

    int gsh_mcorr_set_local_code_and_taps(gsh_mcorr_t* h, int code_length_chips, const float* local_code_in, float* shifts_chips)
    {
        GSH_REQUIRE(h && local_code_in && shifts_chips, "null argument");
        GSH_REQUIRE(code_length_chips >= 1, "code_length_chips %d", code_length_chips);
        if (h->d_in == nullptr) return set_error(GSH_ERR_STATE, "init() has not been called");
        if (h->bank == nullptr || h->bank->max_code_len < code_length_chips)
            {
                if (h->bank) gsh_bank_destroy(h->bank);
                h->bank = nullptr;
                int rc = gsh_bank_create(h->device, 1, code_length_chips, &h->bank);
                if (rc != GSH_OK) return rc;
            }
        int rc = gsh_bank_set_code(h->bank, 0, local_code_in, code_length_chips);
        if (rc != GSH_OK) return rc;
        h->code_len = code_length_chips;
        h->shifts = shifts_chips;
        return GSH_OK;
    }

    int gsh_mcorr_set_input_output_vectors(gsh_mcorr_t* h, float* corr_out_iq, const float* sig_in_iq)
    {
        GSH_REQUIRE(h && corr_out_iq && sig_in_iq, "null argument");
        h->corr_out = corr_out_iq;
        h->sig_in = sig_in_iq;
        return GSH_OK;
    }

    int gsh_mcorr_set_high_dynamics_resampler(gsh_mcorr_t* h, int use_high_dynamics_resampler)
    {
        GSH_REQUIRE(h != nullptr, "null handle");
        h->high_dyn = use_high_dynamics_resampler != 0;
        return GSH_OK;
    }

    int gsh_mcorr_carrier_wipeoff_multicorrelator_resampler(gsh_mcorr_t* h, float rem_carrier_phase_in_rad, float phase_step_rad,
        float phase_rate_step_rad, float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips,
        int signal_length_samples)
    {
        GSH_REQUIRE(h != nullptr, "null handle");
        return mcorr_run(h, h->high_dyn ? 1 : 0, rem_carrier_phase_in_rad, phase_step_rad, phase_rate_step_rad, rem_code_phase_chips,
            code_phase_step_chips, code_phase_rate_step_chips, signal_length_samples);
    }

    int gsh_mcorr_carrier_wipeoff_multicorrelator_resampler6(gsh_mcorr_t* h, float rem_carrier_phase_in_rad, float phase_step_rad,
        float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips, int signal_length_samples)
    {
        GSH_REQUIRE(h != nullptr, "null handle");
        return mcorr_run(h, h->high_dyn ? 2 : 0, rem_carrier_phase_in_rad, phase_step_rad, 0.0f, rem_code_phase_chips,
            code_phase_step_chips, code_phase_rate_step_chips, signal_length_samples);
    }
}
