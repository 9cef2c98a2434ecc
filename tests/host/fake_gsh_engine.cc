// TEST INFRASTRUCTURE ONLY -- a CPU stand-in for the DEVICE half of the C ABI (include/gnss_sdr_hip.h), so that the host-side runtime
// (gnss-sdr_amd/host/hip_tracking_runtime.cc), the GNU Radio block shell and the adapters can be exercised -- threads, locks, bookkeeping,
// Gnss_Synchro items, dump, TOW -- in the CPU test suite and under ThreadSanitizer, where no GPU exists.  It is linked into TEST programs
// only (tests/host/*_fake), in front of libgnss_sdr_hip.so: the entry points defined here interpose the library's, everything else
// (gsh_trk_pull_in, gsh_trk_write_dump, gsh_last_error: host-only code) is the library's own.  The product has no CPU path; nothing under
// gnss-sdr_amd/ or include/ refers to this file.
//
// What stands in for the kernel is the ORACLE (oracle/gnss_oracle_loop.c, the restatement pinned to the reference block): a channel's whole
// trajectory over the test's stream is computed once at gsh_trk_start_ex -- the test hands the complete stream over beforehand
// (fake_gsh_set_reference_stream) -- and a run releases the periods whose windows are resident in the fake ring.  Every push is checked,
// sample for sample, against that stream at the absolute index the ring gives it: a duplicated or shifted push (two channel threads appending
// the same samples) fails loudly (fake_gsh_push_mismatches()).
#include "gnss_oracle.h"
#include "gnss_sdr_hip.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <mutex>
#include <vector>

namespace gsh
{
int set_error(int code, const char* fmt, ...);  // libgnss_sdr_hip.so (thread-local text behind gsh_last_error)
}

static_assert(sizeof(oracle_trk_conf) == sizeof(gsh_trk_conf), "oracle_trk_conf and gsh_trk_conf share one layout");
static_assert(sizeof(oracle_trk_epoch) == sizeof(gsh_trk_epoch), "oracle_trk_epoch and gsh_trk_epoch share one layout");

namespace
{
std::mutex g_ref_mutex;
const float* g_ref_iq = nullptr;
uint64_t g_ref_n = 0;
std::atomic<uint64_t> g_mismatch{0};
std::atomic<int> g_busy_handles{0};  // > 1 concurrent entries into one handle = a host-side locking bug
}  // namespace

struct gsh_stream
{
    int device{0};
    uint64_t capacity{0}, max_window{0};
    std::atomic<uint64_t> next{0}, origin{0};  // (atomic: live takes read them from the block threads while a push is in progress)
    std::atomic<int> inside{0};
};

struct gsh_stream_group
{
    std::vector<gsh_stream*> rings;
};

struct FakeChannel
{
    bool active{false};
    uint64_t start{0};
    std::vector<gsh_trk_epoch> all;  // the whole trajectory over the reference stream
    size_t released{0};
    uint64_t pos{0};
    // live mode: records the "device" has finished and the host has not taken; one block thread at a time per channel (checked)
    std::deque<gsh_trk_epoch> produced;
    uint64_t taken_next{0};
    std::atomic<int> inside{0};
    FakeChannel() = default;
    FakeChannel(const FakeChannel& o) : active(o.active), start(o.start), all(o.all), released(o.released), pos(o.pos), produced(o.produced), taken_next(o.taken_next) {}
    FakeChannel& operator=(const FakeChannel& o)
    {
        active = o.active;
        start = o.start;
        all = o.all;
        released = o.released;
        pos = o.pos;
        produced = o.produced;
        taken_next = o.taken_next;
        return *this;
    }
};

struct gsh_trk
{
    int device{0};
    gsh_trk_conf conf{};
    int n_channels{0}, max_code_len{0};
    gsh_stream* ring{nullptr};
    std::vector<FakeChannel> ch;
    std::vector<gsh_trk_epoch> pending_rec;
    std::vector<int32_t> pending_done;
    int pending_epochs{-1};
    std::atomic<int> inside{0};
    // live mode: a "residency" serves a bounded number of records and then ends by itself, so that the runtime's relaunch path is exercised
    std::atomic<bool> live_ready{false};
    std::atomic<int> live_left{0};  // records the residency in flight may still produce (0: none in flight)
    std::atomic<bool> dead_residency{false};  // fault injection: the residency in flight never produces a record
};

typedef std::complex<double> cd;

struct gsh_acq
{
    int device{0};
    gsh_acq_conf conf{};
    std::vector<std::vector<cd>> codes;    // conj(FFT(replica)) per slot
    std::vector<std::vector<float>> grids;  // |.|^2, bins x effective_fft_size, per slot (or per position for dwell_slots)
    std::atomic<int> inside{0};
};

namespace
{
// ---- fault injection (round 5: SURVEY section 5 "engine errors must surface as 'no detection' / loss of lock, never as an exception across the GNU Radio thread").
// fake_gsh_inject_fault(kind, after, count, channel): the next `count` (< 0: all) calls of that kind are failed with GSH_ERR_HIP once `after` more calls of the kind
// have passed; for the per-channel kinds (live take, start) only calls that name `channel` count (< 0: any channel).
enum Fault_Kind
{
    FAULT_PUSH = 0,       // gsh_stream_push* / gsh_stream_group_push
    FAULT_LIVE_TAKE = 1,  // gsh_trk_live_take
    FAULT_RESIDENCY = 2,  // gsh_trk_live_begin succeeds but the residency never reports a record
    FAULT_ACQ_DWELL = 3,  // gsh_acq_dwell*, gsh_acq_dwell_slots
    FAULT_TRK_START = 4,  // gsh_trk_start_ex
    FAULT_RUN = 5,        // gsh_trk_run_begin (launched mode)
    FAULT_ACQ_CREATE = 6,
    FAULT_KINDS = 7
};
struct Fault
{
    std::atomic<long> after{0}, count{0};
    std::atomic<int> channel{-1};
    std::atomic<long> hits{0};
};
Fault g_faults[FAULT_KINDS];

bool fake_fault_hit(int kind, int channel = -1)
{
    Fault& f = g_faults[kind];
    if (f.count.load() == 0) return false;
    const int want = f.channel.load();
    if (want >= 0 && channel != want) return false;
    // (several block threads come through here at once: exactly `after` calls pass, exactly `count` fail)
    long a = f.after.load();
    while (a > 0)
        if (f.after.compare_exchange_weak(a, a - 1)) return false;
    long c = f.count.load();
    for (;;)
        {
            if (c == 0) return false;
            if (c < 0) break;  // every call from now on
            if (f.count.compare_exchange_weak(c, c - 1)) break;
        }
    f.hits.fetch_add(1);
    return true;
}

// unnormalised DFT (forward: exp(-j...), inverse: exp(+j...)), decimation in time over the prime factors of the length
void fake_fft_rec(const cd* in, size_t stride, size_t n, cd* out, const std::vector<cd>& w, size_t wstep, bool inverse)
{
    if (n == 1)
        {
            out[0] = in[0];
            return;
        }
    size_t p = 2;
    while (n % p != 0) p++;
    const size_t m = n / p;
    std::vector<cd> sub(n);
    for (size_t r = 0; r < p; r++) fake_fft_rec(in + r * stride, stride * p, m, sub.data() + r * m, w, wstep * p, inverse);
    const size_t N = w.size();
    for (size_t k = 0; k < n; k++)
        {
            cd acc(0.0, 0.0);
            for (size_t r = 0; r < p; r++)
                {
                    const size_t idx = (r * k * wstep) % N;
                    const cd tw = inverse ? std::conj(w[idx]) : w[idx];
                    acc += sub[r * m + (k % m)] * tw;
                }
            out[k] = acc;
        }
}
void fake_fft(std::vector<cd>& x, bool inverse)
{
    static std::mutex mu;
    static std::map<size_t, std::vector<cd>> tables;
    const size_t n = x.size();
    const std::vector<cd>* w;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto& t = tables[n];
        if (t.empty())
            {
                t.resize(n);
                for (size_t k = 0; k < n; k++) t[k] = std::polar(1.0, -2.0 * M_PI * static_cast<double>(k) / static_cast<double>(n));
            }
        w = &t;
    }
    std::vector<cd> y(n);
    fake_fft_rec(x.data(), 1, n, y.data(), *w, 1, inverse);
    x.swap(y);
}

// one search of gsh_acq_dwell / _dwell_slots / _dwell_step2 (step_center != nullptr)
int fake_acq_search(gsh_acq* a, const float* in_iq, const std::vector<uint32_t>& slots, const float* step_center, const float* power_step_one, int accumulate,
    uint32_t dwell_count, gsh_acq_result* results, bool grid_by_position)
{
    const gsh_acq_conf& c = a->conf;
    const uint32_t N = c.fft_size, E = c.effective_fft_size;
    const double fs = static_cast<double>(c.fs_in);
    for (size_t i = 0; i < slots.size(); i++)
        {
            if (slots[i] >= a->codes.size() || a->codes[slots[i]].size() != N) return gsh::set_error(GSH_ERR_STATE, "fake: no local code in slot %u", slots[i]);
            const bool step2 = step_center != nullptr;
            const uint32_t bins = step2 ? c.num_doppler_bins_step2 : c.num_doppler_bins;
            std::vector<float>& grid = a->grids[grid_by_position ? i : slots[i]];
            if (!accumulate || grid.size() != static_cast<size_t>(bins) * E) grid.assign(static_cast<size_t>(bins) * E, 0.0F);
            float best = 0.0F;
            uint32_t best_bin = 0, best_t = 0;
            for (uint32_t d = 0; d < bins; d++)
                {
                    // acq.cc:284-301: the bin's frequency; the wipe-off phase step is formed in float there, which moves nothing a peak index could see
                    const double f = step2 ? static_cast<double>(step_center[i] + (static_cast<float>(d) - static_cast<float>(std::floor(c.num_doppler_bins_step2 / 2.0))) * c.doppler_step2)
                                           : static_cast<double>(c.doppler_bias + (-c.doppler_max + c.doppler_center + c.doppler_step * static_cast<int32_t>(d)));
                    std::vector<cd> x(N, cd(0.0, 0.0));
                    for (uint32_t k = 0; k < N && k < c.consumed_samples; k++)
                        x[k] = cd(in_iq[2 * k], in_iq[2 * k + 1]) * std::polar(1.0, -2.0 * M_PI * std::fmod(f * static_cast<double>(k) / fs, 1.0));
                    fake_fft(x, false);
                    for (uint32_t k = 0; k < N; k++) x[k] *= a->codes[slots[i]][k];
                    fake_fft(x, true);
                    float* row = grid.data() + static_cast<size_t>(d) * E;
                    float row_max = -1.0F;
                    uint32_t row_t = 0;
                    for (uint32_t k = 0; k < E; k++)
                        {
                            const float m = static_cast<float>(std::norm(x[k]));
                            row[k] = accumulate ? row[k] + m : m;
                            if (row[k] > row_max)
                                {
                                    row_max = row[k];
                                    row_t = k;
                                }
                        }
                    if (row_max > best)  // strict: the first bin wins ties (acq.cc:420)
                        {
                            best = row_max;
                            best_bin = d;
                            best_t = row_t;
                        }
                }
            gsh_acq_result& r = results[i];
            r = gsh_acq_result{};
            r.index_time = best_t;
            r.index_doppler = best_bin;
            r.peak = best;
            if (!step2)
                {
                    const uint32_t opp = (best_bin + bins / 2) % bins;  // acq.cc:428-431
                    float sum = 0.0F;
                    for (uint32_t k = 0; k < E; k++) sum += grid[static_cast<size_t>(opp) * E + k];
                    r.input_power = sum / static_cast<float>(E) / 2.0F / static_cast<float>(std::max<uint32_t>(dwell_count, 1));
                    r.doppler_hz = -c.doppler_max + c.doppler_center + c.doppler_step * static_cast<int32_t>(best_bin);
                }
            else
                {
                    r.input_power = power_step_one != nullptr ? power_step_one[i] : 0.0F;
                    r.doppler_hz = static_cast<int32_t>(step_center[i] + (static_cast<float>(best_bin) - static_cast<float>(std::floor(c.num_doppler_bins_step2 / 2.0))) * c.doppler_step2);
                }
            r.test_statistics = r.input_power < std::numeric_limits<float>::epsilon() ? 0.0F : best / r.input_power;
            r.acq_delay_samples = std::fmod(static_cast<float>(best_t), c.samples_per_code);
        }
    return GSH_OK;
}
}  // namespace

namespace
{
struct Guard  // one thread at a time per handle, as the ABI demands of its callers
{
    std::atomic<int>& f;
    explicit Guard(std::atomic<int>& flag) : f(flag)
    {
        if (f.fetch_add(1) != 0)
            {
                g_busy_handles++;
                std::fprintf(stderr, "FAKE ENGINE: two threads inside one handle at once\n");
            }
    }
    ~Guard() { f.fetch_sub(1); }
};

uint64_t oldest(const gsh_stream* s)
{
    const uint64_t next = s->next.load(), origin = s->origin.load();
    const uint64_t by_cap = next > s->capacity ? next - s->capacity : 0;
    return std::max(by_cap, origin);
}

int push_common(gsh_stream* s, const void* items, uint64_t n, int item_type, uint64_t* first_index)
{
    if (s == nullptr || (n != 0 && items == nullptr)) return gsh::set_error(GSH_ERR_INVALID, "fake: null argument");
    if (item_type != GSH_ITEM_GR_COMPLEX) return gsh::set_error(GSH_ERR_UNSUPPORTED, "fake: complex64 items only");
    if (n > s->capacity) return gsh::set_error(GSH_ERR_INVALID, "a push of %llu samples exceeds the ring capacity %llu", (unsigned long long)n, (unsigned long long)s->capacity);
    if (n != 0 && fake_fault_hit(FAULT_PUSH)) return gsh::set_error(GSH_ERR_HIP, "fake engine: injected failure of a push (hipMemcpyAsync: an illegal memory access was encountered)");
    Guard g(s->inside);
    const uint64_t at = s->next.load();
    if (first_index) *first_index = at;
    {
        std::lock_guard<std::mutex> lk(g_ref_mutex);
        if (g_ref_iq != nullptr)
            {
                if (at + n > g_ref_n || std::memcmp(items, g_ref_iq + 2 * at, sizeof(float) * 2 * n) != 0)
                    {
                        g_mismatch++;
                        std::fprintf(stderr, "FAKE ENGINE: push of %llu samples at absolute index %llu does not match the stream\n", (unsigned long long)n, (unsigned long long)at);
                    }
            }
    }
    s->next.store(at + n);
    return GSH_OK;
}
}  // namespace

extern "C"
{
    // ---- test hooks
    void fake_gsh_set_reference_stream(const float* iq, uint64_t n)
    {
        std::lock_guard<std::mutex> lk(g_ref_mutex);
        g_ref_iq = iq;
        g_ref_n = n;
    }
    uint64_t fake_gsh_push_mismatches(void) { return g_mismatch.load(); }
    void fake_gsh_inject_fault(int kind, long after, long count, int channel)
    {
        if (kind < 0 || kind >= FAULT_KINDS) return;
        g_faults[kind].after.store(after);
        g_faults[kind].channel.store(channel);
        g_faults[kind].count.store(count);
    }
    long fake_gsh_fault_hits(int kind) { return (kind >= 0 && kind < FAULT_KINDS) ? g_faults[kind].hits.load() : 0; }
    void fake_gsh_clear_faults(void)
    {
        for (auto& f : g_faults)
            {
                f.count.store(0);
                f.after.store(0);
                f.channel.store(-1);
                f.hits.store(0);
            }
    }
    int fake_gsh_concurrent_handle_entries(void) { return g_busy_handles.load(); }

    // FAKE_GSH_DEVICES=N in the environment: the stand-in pretends to N devices (the multi-GPU layout of the adapters, exercised on the CPU)
    int gsh_device_count(void)
    {
        static const int n = [] {
            const char* e = std::getenv("FAKE_GSH_DEVICES");
            return e != nullptr ? std::max(1, std::atoi(e)) : 1;
        }();
        return n;
    }

    // ---- sample ring
    int gsh_stream_create(int device, uint64_t capacity_samples, uint32_t max_window_samples, gsh_stream_t** out)
    {
        if (out == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null out pointer");
        *out = nullptr;
        if (device < 0 || device >= gsh_device_count()) return gsh::set_error(GSH_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, gsh_device_count() - 1);
        if (max_window_samples < 1 || capacity_samples < 2ull * max_window_samples) return gsh::set_error(GSH_ERR_INVALID, "fake: bad ring geometry");
        auto* s = new gsh_stream();
        s->device = device;
        s->capacity = capacity_samples + (capacity_samples & 1ull);
        s->max_window = max_window_samples;
        *out = s;
        return GSH_OK;
    }
    void gsh_stream_destroy(gsh_stream_t* s) { delete s; }
    int gsh_stream_push(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int, uint64_t* first_index) { return push_common(s, items, n, item_type, first_index); }
    int gsh_stream_push_staged(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int, uint64_t* first_index) { return push_common(s, items, n, item_type, first_index); }
    int gsh_stream_push_pinned(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int, uint64_t* first_index) { return push_common(s, items, n, item_type, first_index); }
    int gsh_stream_push_pinned_async(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int, uint64_t* first_index) { return push_common(s, items, n, item_type, first_index); }
    int gsh_stream_wait_copied(gsh_stream_t*) { return GSH_OK; }
    int gsh_stream_wait_copied_upto(gsh_stream_t* s, uint64_t, uint64_t* complete_upto)
    {
        if (s == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null stream");
        if (complete_upto) *complete_upto = s->next.load();
        return GSH_OK;
    }
    int gsh_host_register(int, void*, size_t) { return GSH_OK; }
    int gsh_host_unregister(void*) { return GSH_OK; }
    int gsh_stream_seek(gsh_stream_t* s, uint64_t next_index)
    {
        if (s == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null stream");
        Guard g(s->inside);
        s->next.store(next_index);
        s->origin.store(next_index);
        return GSH_OK;
    }
    int gsh_stream_range(gsh_stream_t* s, uint64_t* lo, uint64_t* hi)
    {
        if (s == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null stream");
        if (lo) *lo = oldest(s);
        if (hi) *hi = s->next.load();
        return GSH_OK;
    }

    // ---- the block-replication group (single process, several devices): one fake ring per device, every push lands in all of them -- and is checked
    // against the reference stream in each (push_common)
    int gsh_stream_group_create(const int* devices, int n_devices, uint64_t capacity_samples, uint32_t max_window_samples, int, gsh_stream_group_t** out)
    {
        if (out == nullptr || devices == nullptr || n_devices < 1) return gsh::set_error(GSH_ERR_INVALID, "fake: bad group arguments");
        *out = nullptr;
        auto* g = new gsh_stream_group();
        for (int i = 0; i < n_devices; i++)
            {
                gsh_stream_t* s = nullptr;
                if (gsh_stream_create(devices[i], capacity_samples, max_window_samples, &s) != GSH_OK)
                    {
                        gsh_stream_group_destroy(g);
                        return GSH_ERR_NO_DEVICE;
                    }
                g->rings.push_back(s);
            }
        *out = g;
        return GSH_OK;
    }
    void gsh_stream_group_destroy(gsh_stream_group_t* g)
    {
        if (g == nullptr) return;
        for (gsh_stream* s : g->rings) delete s;
        delete g;
    }
    int gsh_stream_group_size(const gsh_stream_group_t* g) { return g != nullptr ? static_cast<int>(g->rings.size()) : 0; }
    gsh_stream_t* gsh_stream_group_ring(gsh_stream_group_t* g, int i) { return (g != nullptr && i >= 0 && i < static_cast<int>(g->rings.size())) ? g->rings[static_cast<size_t>(i)] : nullptr; }
    int gsh_stream_group_push(gsh_stream_group_t* g, const void* host_items, uint64_t n, int item_type, int, uint64_t* first_index)
    {
        if (g == nullptr || g->rings.empty()) return gsh::set_error(GSH_ERR_INVALID, "null group");
        uint64_t first = 0;
        for (size_t i = 0; i < g->rings.size(); i++)
            {
                uint64_t f = 0;
                const int rc = push_common(g->rings[i], host_items, n, item_type, &f);
                if (rc != GSH_OK) return rc;
                if (i == 0) first = f;
                if (f != first)
                    {
                        g_mismatch++;
                        std::fprintf(stderr, "FAKE ENGINE: the rings of a group have drifted apart (%llu vs %llu)\n", (unsigned long long)f, (unsigned long long)first);
                    }
            }
        if (first_index) *first_index = first;
        return GSH_OK;
    }
    int gsh_stream_group_wait(gsh_stream_group_t* g) { return g != nullptr ? GSH_OK : gsh::set_error(GSH_ERR_INVALID, "null group"); }

    // ---- tracking loop
    int gsh_trk_create(int device, const gsh_trk_conf* conf, int n_channels, int max_code_length, gsh_trk_t** out)
    {
        if (out == nullptr || conf == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null argument");
        *out = nullptr;
        if (device < 0 || device >= gsh_device_count()) return gsh::set_error(GSH_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, gsh_device_count() - 1);
        auto* t = new gsh_trk();
        t->device = device;
        t->conf = *conf;
        t->n_channels = n_channels;
        t->max_code_len = max_code_length;
        t->ch.resize(static_cast<size_t>(n_channels));
        *out = t;
        return GSH_OK;
    }
    void gsh_trk_destroy(gsh_trk_t* t) { delete t; }
    int gsh_trk_set_stream_ring(gsh_trk_t* t, gsh_stream_t* s)
    {
        if (t == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null handle");
        if (s != nullptr && s->device != t->device) return gsh::set_error(GSH_ERR_INVALID, "the ring lives on device %d, the loop on device %d", s->device, t->device);
        if (s != nullptr && s->max_window < t->conf.vector_length)
            return gsh::set_error(GSH_ERR_INVALID, "the ring's max_window_samples %llu is shorter than vector_length %u", (unsigned long long)s->max_window, t->conf.vector_length);
        t->ring = s;
        return GSH_OK;
    }
    int gsh_trk_start_ex(gsh_trk_t* t, int channel, const float* code, const float* data_code, int code_length, uint64_t start_sample, uint64_t acq_sample_stamp,
        double acq_carrier_doppler_hz, double initial_acc_carrier_phase_rad)
    {
        return gsh_trk_start_flags(t, channel, code, data_code, code_length, start_sample, acq_sample_stamp, acq_carrier_doppler_hz, initial_acc_carrier_phase_rad, 0U);
    }
    int gsh_trk_start_flags(gsh_trk_t* t, int channel, const float* code, const float* data_code, int code_length, uint64_t start_sample, uint64_t acq_sample_stamp,
        double acq_carrier_doppler_hz, double initial_acc_carrier_phase_rad, uint32_t flags)
    {
        if (t == nullptr || code == nullptr || channel < 0 || channel >= t->n_channels) return gsh::set_error(GSH_ERR_INVALID, "fake: bad start arguments");
        if (fake_fault_hit(FAULT_TRK_START, channel)) return gsh::set_error(GSH_ERR_HIP, "fake engine: injected failure of gsh_trk_start");
        Guard g(t->inside);
        if (t->live_left.load() > 0) return gsh::set_error(GSH_ERR_STATE, "gsh_trk_start: a live residency is in flight (gsh_trk_live_quiesce first)");
        Guard gc(t->ch[static_cast<size_t>(channel)].inside);  // (a take of this channel at the same time is the caller's bug)
        const float* ref;
        uint64_t n_ref;
        {
            std::lock_guard<std::mutex> lk(g_ref_mutex);
            ref = g_ref_iq;
            n_ref = g_ref_n;
        }
        if (ref == nullptr) return gsh::set_error(GSH_ERR_STATE, "fake engine: fake_gsh_set_reference_stream() has not been called");
        FakeChannel& c = t->ch[static_cast<size_t>(channel)];
        c = FakeChannel{};
        c.start = start_sample;
        c.pos = start_sample;
        const uint64_t vlen = t->conf.vector_length;
        const uint64_t most = start_sample + vlen <= n_ref ? (n_ref - start_sample) / (vlen > 1 ? vlen - 1 : 1) + 2 : 0;
        c.all.resize(static_cast<size_t>(most));
        int got = 0;
        if (most > 0)
            got = oracle_trk_run_flags(reinterpret_cast<const oracle_trk_conf*>(&t->conf), code, t->conf.track_pilot ? data_code : nullptr, code_length, ref, n_ref, start_sample,
                acq_sample_stamp, acq_carrier_doppler_hz, static_cast<int>(most), reinterpret_cast<oracle_trk_epoch*>(c.all.data()),
                (flags & GSH_TRK_START_PULL_IN_OVER) ? ORACLE_TRK_PULL_IN_OVER : 0U);
        c.all.resize(static_cast<size_t>(std::max(got, 0)));
        // the oracle starts d_acc_carrier_phase_rad at 0; the pull-in alignment's contribution (trk.cc:1966) rides on it until the first narrow-tracking
        // period re-initialises the accumulator (check_carrier_phase_coherent_initialization, trk.cc:1350-1357)
        for (auto& r : c.all)
            {
                if (r.state == 4) break;
                r.acc_carrier_phase_rad += initial_acc_carrier_phase_rad;
            }
        c.active = true;
        c.taken_next = start_sample;
        return GSH_OK;
    }
    int gsh_trk_start(gsh_trk_t* t, int channel, const float* code, const float* data_code, int code_length, uint64_t start_sample, uint64_t acq_sample_stamp,
        double acq_carrier_doppler_hz)
    {
        return gsh_trk_start_ex(t, channel, code, data_code, code_length, start_sample, acq_sample_stamp, acq_carrier_doppler_hz, 0.0);
    }
    int gsh_trk_stop(gsh_trk_t* t, int channel)
    {
        if (t == nullptr || channel < 0 || channel >= t->n_channels) return gsh::set_error(GSH_ERR_INVALID, "fake: bad stop arguments");
        Guard g(t->inside);
        if (t->live_left.load() > 0) return gsh::set_error(GSH_ERR_STATE, "gsh_trk_stop: a live residency is in flight (gsh_trk_live_quiesce first)");
        Guard gc(t->ch[static_cast<size_t>(channel)].inside);
        t->ch[static_cast<size_t>(channel)].active = false;
        t->ch[static_cast<size_t>(channel)].produced.clear();
        return GSH_OK;
    }
    int gsh_trk_run_begin(gsh_trk_t* t, int n_epochs, int want_records)
    {
        if (t == nullptr || n_epochs < 0) return gsh::set_error(GSH_ERR_INVALID, "fake: bad run arguments");
        if (t->ring == nullptr) return gsh::set_error(GSH_ERR_STATE, "no IF stream attached (gsh_trk_set_stream_*)");
        if (t->pending_epochs >= 0) return gsh::set_error(GSH_ERR_STATE, "gsh_trk_run_begin: the previous run has not been ended");
        if (fake_fault_hit(FAULT_RUN)) return gsh::set_error(GSH_ERR_HIP, "fake engine: injected failure of a launch");
        Guard g(t->inside);
        if (t->ring->inside.load() != 0)
            {
                g_busy_handles++;
                std::fprintf(stderr, "FAKE ENGINE: a launch was queued while a push was inside the ring (the ring's lock was not held)\n");
            }
        (void)want_records;
        if (t->live_left.load() > 0) return gsh::set_error(GSH_ERR_STATE, "gsh_trk_run_begin: a live residency is in flight (gsh_trk_live_quiesce first)");
        const uint64_t vlen = t->conf.vector_length, next = t->ring->next.load(), old = oldest(t->ring);
        t->pending_rec.assign(static_cast<size_t>(t->n_channels) * static_cast<size_t>(n_epochs), gsh_trk_epoch{});
        t->pending_done.assign(static_cast<size_t>(t->n_channels), 0);
        for (int c = 0; c < t->n_channels; c++)
            {
                FakeChannel& C = t->ch[static_cast<size_t>(c)];
                int done = 0;
                while (C.active && done < n_epochs && C.released < C.all.size())
                    {
                        const gsh_trk_epoch& r = C.all[C.released];
                        if (r.sample_counter + vlen > next || r.sample_counter < old) break;
                        t->pending_rec[static_cast<size_t>(c) * static_cast<size_t>(n_epochs) + static_cast<size_t>(done)] = r;
                        done++;
                        C.released++;
                        if (r.flags & 2)
                            {
                                C.active = false;
                                break;
                            }
                        C.pos = r.sample_counter + static_cast<uint64_t>(r.prn_length_samples);
                    }
                t->pending_done[static_cast<size_t>(c)] = done;
            }
        t->pending_epochs = n_epochs;
        return GSH_OK;
    }
    int gsh_trk_run_end(gsh_trk_t* t, gsh_trk_epoch* records, int32_t* epochs_done)
    {
        if (t == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null handle");
        if (t->pending_epochs < 0) return gsh::set_error(GSH_ERR_STATE, "gsh_trk_run_end without gsh_trk_run_begin");
        Guard g(t->inside);
        if (records != nullptr && !t->pending_rec.empty()) std::memcpy(records, t->pending_rec.data(), sizeof(gsh_trk_epoch) * t->pending_rec.size());
        if (epochs_done != nullptr) std::memcpy(epochs_done, t->pending_done.data(), sizeof(int32_t) * t->pending_done.size());
        t->pending_epochs = -1;
        return GSH_OK;
    }
    int gsh_trk_run(gsh_trk_t* t, int n_epochs, gsh_trk_epoch* records, int32_t* epochs_done)
    {
        const int rc = gsh_trk_run_begin(t, n_epochs, records != nullptr);
        return rc != GSH_OK ? rc : gsh_trk_run_end(t, records, epochs_done);
    }
    // ---- live mode.  The stand-in "device" works at take time: what a resident kernel would have finished by now -- every period whose window lies in
    // the fake ring while a residency is in flight -- is moved to the channel's queue of finished records, then handed out under the same rules as the
    // library's gsh_trk_live_take.  A residency ends by itself after a bounded number of records (the kernel's time budget, in miniature).
    int gsh_trk_set_split(gsh_trk_t* t, int g) { return (t != nullptr && g >= 0 && g <= 8) ? GSH_OK : gsh::set_error(GSH_ERR_INVALID, "gsh_trk_set_split"); }
    int gsh_trk_live_configure(gsh_trk_t* t, uint32_t, uint32_t) { return t != nullptr ? GSH_OK : gsh::set_error(GSH_ERR_INVALID, "null handle"); }
    int gsh_trk_live_begin(gsh_trk_t* t)
    {
        if (t == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null handle");
        if (t->ring == nullptr) return gsh::set_error(GSH_ERR_STATE, "live mode follows a sample ring (gsh_trk_set_stream_ring)");
        if (t->pending_epochs >= 0) return gsh::set_error(GSH_ERR_STATE, "gsh_trk_live_begin: a run has been begun and not ended");
        Guard g(t->inside);
        if (t->ring->inside.load() != 0)
            {
                g_busy_handles++;
                std::fprintf(stderr, "FAKE ENGINE: a residency was queued while a push was inside the ring (the ring's lock was not held)\n");
            }
        t->live_ready.store(true);
        if (fake_fault_hit(FAULT_RESIDENCY)) t->dead_residency.store(true);  // the kernel is "in flight" and never writes a record (a hung device)
        if (t->live_left.load() <= 0) t->live_left.store(37 * std::max(1, t->n_channels / 8));
        return GSH_OK;
    }
    int gsh_trk_live_in_flight(gsh_trk_t* t, int32_t* n)
    {
        if (t == nullptr || n == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null argument");
        Guard g(t->inside);
        *n = t->live_left.load() > 0 ? 1 : 0;
        return GSH_OK;
    }
    int gsh_trk_live_quiesce(gsh_trk_t* t)
    {
        if (t == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null handle");
        Guard g(t->inside);
        t->live_left.store(0);
        t->dead_residency.store(false);
        return GSH_OK;
    }
    int gsh_trk_live_take(gsh_trk_t* t, int channel, uint64_t limit_end, int max_records, gsh_trk_epoch* out, int32_t* n_out, int32_t* pending, uint64_t* next_window,
        int32_t* active, int32_t* resident)
    {
        if (resident) *resident = (t != nullptr && t->live_left.load() > 0) ? 1 : 0;
        if (t == nullptr || n_out == nullptr || channel < 0 || channel >= t->n_channels) return gsh::set_error(GSH_ERR_INVALID, "fake: bad take arguments");
        *n_out = 0;
        if (fake_fault_hit(FAULT_LIVE_TAKE, channel)) return gsh::set_error(GSH_ERR_HIP, "fake engine: injected failure of gsh_trk_live_take (channel %d)", channel);
        FakeChannel& C = t->ch[static_cast<size_t>(channel)];
        if (!t->live_ready.load())
            {
                if (pending) *pending = 0;
                if (next_window) *next_window = C.pos;
                if (active) *active = C.active ? 1 : 0;
                return GSH_OK;
            }
        Guard gc(C.inside);  // one thread per channel at a time; other channels' takes, pushes and residency management run beside it
        const uint64_t vlen = t->conf.vector_length;
        while (C.active && C.released < C.all.size() && t->live_left.load() > 0 && !t->dead_residency.load())
            {
                const gsh_trk_epoch& r = C.all[C.released];
                if (r.sample_counter + vlen > t->ring->next.load() || r.sample_counter < oldest(t->ring)) break;
                t->live_left.fetch_sub(1);
                C.produced.push_back(r);
                C.released++;
                if (r.flags & 2)
                    {
                        C.active = false;
                        break;
                    }
                C.pos = r.sample_counter + static_cast<uint64_t>(r.prn_length_samples);
            }
        int n = 0;
        while (n < max_records && !C.produced.empty())
            {
                const gsh_trk_epoch& r = C.produced.front();
                const bool lost = (r.flags & 2) != 0;
                if (!lost && r.sample_counter + std::max<uint64_t>(vlen, static_cast<uint64_t>(std::max(r.prn_length_samples, 0))) > limit_end) break;
                out[n++] = r;
                if (!lost) C.taken_next = r.sample_counter + static_cast<uint64_t>(std::max(r.prn_length_samples, 0));
                C.produced.pop_front();
                if (lost) break;
            }
        *n_out = n;
        if (pending) *pending = static_cast<int32_t>(C.produced.size());
        if (next_window) *next_window = C.taken_next;
        if (active) *active = C.active ? 1 : 0;
        return GSH_OK;
    }
    int gsh_trk_positions(gsh_trk_t* t, uint64_t* next_window, int32_t* active)
    {
        if (t == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null handle");
        for (int c = 0; c < t->n_channels; c++)
            {
                if (next_window) next_window[c] = t->ch[static_cast<size_t>(c)].pos;
                if (active) active[c] = t->ch[static_cast<size_t>(c)].active ? 1 : 0;
            }
        return GSH_OK;
    }

    // ---- acquisition (round 5: the whole Channel -- acquisition adapter, ChannelFsm, tracking adapter -- runs on the CPU under ThreadSanitizer).  The stand-in
    // computes what include/gnss_sdr_hip.h documents for gsh_acq_*: replica placed per the padding rules, FFT, conjugate; per Doppler bin wipe-off, FFT, product,
    // inverse FFT, |.|^2 (+= for non-coherent dwells); the max-to-input-power statistic.  Double-precision mixed-radix transforms: slow and exact enough that every
    // peak index and Doppler bin equals the reference block's.  Plain and two-step CFAR searches of gr_complex / cshort items; nothing else (GSH_ERR_UNSUPPORTED).
    int gsh_acq_create(int device, const gsh_acq_conf* conf, gsh_acq_t** out)
    {
        if (out == nullptr || conf == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null argument");
        *out = nullptr;
        if (device < 0 || device >= gsh_device_count()) return gsh::set_error(GSH_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, gsh_device_count() - 1);
        if (conf->use_cfar == 0 || conf->bit_transition_flag != 0 || conf->fold > 1) return gsh::set_error(GSH_ERR_UNSUPPORTED, "fake engine: plain CFAR searches only");
        if (fake_fault_hit(FAULT_ACQ_CREATE)) return gsh::set_error(GSH_ERR_HIP, "fake engine: injected failure of gsh_acq_create");
        auto* a = new gsh_acq();
        a->device = device;
        a->conf = *conf;
        if (a->conf.num_doppler_bins == 0)
            a->conf.num_doppler_bins = static_cast<uint32_t>(std::ceil(2.0 * conf->doppler_max / static_cast<double>(conf->doppler_step)));
        a->codes.assign(conf->max_prn, std::vector<cd>());
        a->grids.assign(conf->max_prn, std::vector<float>());
        *out = a;
        return GSH_OK;
    }
    void gsh_acq_destroy(gsh_acq_t* a) { delete a; }
    int gsh_acq_set_local_code(gsh_acq_t* a, uint32_t slot, const float* code_iq)
    {
        if (a == nullptr || code_iq == nullptr || slot >= a->conf.max_prn) return gsh::set_error(GSH_ERR_INVALID, "fake: bad set_local_code arguments");
        Guard g(a->inside);
        const uint32_t N = a->conf.fft_size, C = a->conf.consumed_samples;
        std::vector<cd> x(N, cd(0.0, 0.0));
        // acq.cc:236-247: the replica at the front, or -- when the block integrates more than one code period -- behind N - C zeros
        const uint32_t off = (C < N) ? C : 0;
        for (uint32_t i = 0; i < C && off + i < N; i++) x[off + i] = cd(code_iq[2 * i], code_iq[2 * i + 1]);
        fake_fft(x, false);
        for (auto& v : x) v = std::conj(v);
        a->codes[slot] = std::move(x);
        return GSH_OK;
    }
    int gsh_acq_set_doppler_center(gsh_acq_t* a, int32_t doppler_center)
    {
        if (a == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null handle");
        Guard g(a->inside);
        a->conf.doppler_center = doppler_center;
        return GSH_OK;
    }
    int gsh_acq_set_doppler_bias(gsh_acq_t* a, int32_t doppler_bias)
    {
        if (a == nullptr) return gsh::set_error(GSH_ERR_INVALID, "null handle");
        Guard g(a->inside);
        a->conf.doppler_bias = doppler_bias;
        return GSH_OK;
    }
    int gsh_acq_dwell(gsh_acq_t* a, const float* in_iq, uint32_t n_prn, int accumulate, uint32_t dwell_count, gsh_acq_result* results)
    {
        if (a == nullptr || in_iq == nullptr || results == nullptr || n_prn > a->conf.max_prn) return gsh::set_error(GSH_ERR_INVALID, "fake: bad dwell arguments");
        Guard g(a->inside);
        if (fake_fault_hit(FAULT_ACQ_DWELL)) return gsh::set_error(GSH_ERR_HIP, "fake engine: injected failure of a dwell");
        std::vector<uint32_t> slots(n_prn);
        for (uint32_t i = 0; i < n_prn; i++) slots[i] = i;
        return fake_acq_search(a, in_iq, slots, nullptr, nullptr, accumulate, dwell_count, results, false);
    }
    int gsh_acq_dwell_slots(gsh_acq_t* a, const float* in_iq, uint32_t n, const uint32_t* prn_slots, gsh_acq_result* results)
    {
        if (a == nullptr || in_iq == nullptr || results == nullptr || prn_slots == nullptr) return gsh::set_error(GSH_ERR_INVALID, "fake: bad dwell arguments");
        Guard g(a->inside);
        if (fake_fault_hit(FAULT_ACQ_DWELL)) return gsh::set_error(GSH_ERR_HIP, "fake engine: injected failure of a dwell");
        return fake_acq_search(a, in_iq, std::vector<uint32_t>(prn_slots, prn_slots + n), nullptr, nullptr, 0, 1, results, true);
    }
    int gsh_acq_dwell_cshort(gsh_acq_t* a, const int16_t* in_iq16, uint32_t n_prn, int accumulate, uint32_t dwell_count, gsh_acq_result* results)
    {
        if (a == nullptr || in_iq16 == nullptr) return gsh::set_error(GSH_ERR_INVALID, "fake: bad dwell arguments");
        std::vector<float> x(2 * static_cast<size_t>(a->conf.consumed_samples));
        for (size_t i = 0; i < x.size(); i++) x[i] = static_cast<float>(in_iq16[i]);
        return gsh_acq_dwell(a, x.data(), n_prn, accumulate, dwell_count, results);
    }
    int gsh_acq_dwell_step2(gsh_acq_t* a, const float* in_iq, uint32_t n, const uint32_t* prn_slots, const float* doppler_center_step_two, const float* input_power_step_one,
        int accumulate, uint32_t dwell_count, gsh_acq_result* results)
    {
        if (a == nullptr || in_iq == nullptr || results == nullptr || prn_slots == nullptr || doppler_center_step_two == nullptr)
            return gsh::set_error(GSH_ERR_INVALID, "fake: bad step-two arguments");
        Guard g(a->inside);
        if (fake_fault_hit(FAULT_ACQ_DWELL)) return gsh::set_error(GSH_ERR_HIP, "fake engine: injected failure of a dwell");
        return fake_acq_search(a, in_iq, std::vector<uint32_t>(prn_slots, prn_slots + n), doppler_center_step_two, input_power_step_one, accumulate, dwell_count, results, false);
    }
    int gsh_acq_read_grid(gsh_acq_t* a, uint32_t prn_slot, float* grid)
    {
        if (a == nullptr || grid == nullptr || prn_slot >= a->grids.size()) return gsh::set_error(GSH_ERR_INVALID, "fake: bad read_grid arguments");
        Guard g(a->inside);
        if (a->grids[prn_slot].empty()) return gsh::set_error(GSH_ERR_STATE, "no dwell has filled this grid");
        std::memcpy(grid, a->grids[prn_slot].data(), sizeof(float) * a->grids[prn_slot].size());
        return GSH_OK;
    }
}
