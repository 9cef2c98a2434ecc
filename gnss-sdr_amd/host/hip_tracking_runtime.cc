/*!
 * \file hip_tracking_runtime.cc
 * \brief Channel-batching runtime behind the tracking blocks; see hip_tracking_runtime.h.
 */
#include "hip_tracking_runtime.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <sys/prctl.h>

namespace
{
constexpr size_t MAX_SLOTS = 4096;  // d_slots is reserved for this many once: readers without the lock (push) never see it move
int64_t now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace


Hip_Tracking_Runtime::Hip_Tracking_Runtime(int device, std::shared_ptr<Hip_Sample_Ring> ring, int periods_per_launch, int channels_per_group, bool live)
    : d_device(device),
      d_ring(std::move(ring)),
      d_periods_per_launch(std::min(std::max(periods_per_launch, 1), 256)),
      d_channels_per_group(std::min(std::max(channels_per_group, 1), 4096)),
      d_live(live)
{
    if (const char* e = std::getenv("GSH_TRK_LAUNCH_AHEAD")) d_launch_ahead = (std::atoi(e) != 0);
    if (const char* e = std::getenv("GSH_TRK_LIVE")) d_live = (std::atoi(e) != 0);
    if (const char* e = std::getenv("GSH_TRK_WORK_GROUPS")) set_work_groups_per_channel(std::atoi(e));  // (tests: <role>.hip_work_groups_per_channel)
    if (const char* e = std::getenv("GSH_TRK_LIVE_SPIN_US")) d_spin_us = d_spin_us_single = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("GSH_TRK_LIVE_SLEEP_US")) d_sleep_us = d_sleep_us_single = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("GSH_TRK_LIVE_SPIN_US_SINGLE")) d_spin_us_single = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("GSH_TRK_LIVE_SINGLE_MAX")) d_single_max_records = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("GSH_TRK_TIMER_SLACK_NS")) d_timer_slack_ns = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("GSH_TRK_PUSH_TRY")) d_push_try = (std::atoi(e) != 0);
    if (const char* e = std::getenv("GSH_TRK_PUSH_BATCH")) d_push_batch = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("GSH_TRK_PUSH_SPARE_SLOWEST")) d_push_spare_slowest = (std::atoi(e) != 0);
    d_slots.reserve(MAX_SLOTS);
}


Hip_Tracking_Runtime::~Hip_Tracking_Runtime()
{
    for (auto& g : d_groups)
        if (g->trk != nullptr) gsh_trk_destroy(g->trk);
}


// the group with this configuration that still has a free channel, or a new one (d_mutex held)
Hip_Tracking_Runtime::Group* Hip_Tracking_Runtime::group_for(const gsh_trk_conf& conf, int max_code_length, int* channel)
{
    for (auto& g : d_groups)
        {
            if (g->max_code_length != max_code_length || std::memcmp(&g->conf, &conf, sizeof(conf)) != 0) continue;
            for (size_t c = 0; c < g->slot_of_channel.size(); c++)
                if (g->slot_of_channel[c] < 0)
                    {
                        *channel = static_cast<int>(c);
                        return g.get();
                    }
        }
    auto g = std::make_unique<Group>();
    g->conf = conf;
    g->max_code_length = max_code_length;
    if (gsh_trk_create(d_device, &g->conf, d_channels_per_group, max_code_length, &g->trk) != GSH_OK)
        {
            d_error = std::string("gsh_trk_create: ") + gsh_last_error();
            return nullptr;
        }
    // (a stream that is resident in several GPUs -- Hip_Sample_Ring over a stream group -- has one ring per device: this runtime's)
    if (d_ring->handle_for(d_device) == nullptr || gsh_trk_set_stream_ring(g->trk, d_ring->handle_for(d_device)) != GSH_OK)
        {
            d_error = d_ring->handle_for(d_device) == nullptr ? "the sample ring is not resident on device " + std::to_string(d_device)
                                                               : std::string("gsh_trk_set_stream_ring: ") + gsh_last_error();
            gsh_trk_destroy(g->trk);
            return nullptr;
        }
    // cooperating work-groups (launched mode): a request the engine refuses for this handle leaves it with one work-group per channel
    if (!d_live && d_work_groups_per_channel != 1 && gsh_trk_set_split(g->trk, d_work_groups_per_channel) != GSH_OK) (void)gsh_trk_set_split(g->trk, 1);
    g->slot_of_channel.assign(static_cast<size_t>(d_channels_per_group), -1);
    g->records.resize(static_cast<size_t>(d_channels_per_group) * static_cast<size_t>(d_periods_per_launch));
    g->done.assign(static_cast<size_t>(d_channels_per_group), 0);
    d_groups.push_back(std::move(g));
    *channel = 0;
    return d_groups.back().get();
}


int Hip_Tracking_Runtime::attach(const gsh_trk_conf& conf, int max_code_length)
{
    if (!ok())
        {
            std::lock_guard<std::mutex> lk(d_mutex);
            d_error = d_ring ? "sample ring: " + d_ring->last_error() : std::string("no sample ring");
            return -1;
        }
    std::lock_guard<std::mutex> lk(d_mutex);
    int channel = -1;
    Group* g = group_for(conf, max_code_length, &channel);
    if (g == nullptr) return -1;
    size_t s = 0;
    while (s < d_slots.size() && d_slots[s]->used) s++;
    if (s == d_slots.size())
        {
            if (d_slots.size() >= MAX_SLOTS)
                {
                    d_error = "more than " + std::to_string(MAX_SLOTS) + " tracking blocks on one runtime";
                    return -1;
                }
            d_slots.push_back(std::make_unique<Slot>());
            d_n_slots.store(d_slots.size(), std::memory_order_release);
        }
    {
        uint64_t v = d_min_vlen.load(std::memory_order_relaxed);
        if (v == 0 || conf.vector_length < v) d_min_vlen.store(conf.vector_length, std::memory_order_relaxed);
    }
    Slot& S = *d_slots[s];
    S.tracking = false;
    S.device_active = false;
    S.live_tracking.store(false, std::memory_order_release);
    S.live_next_window.store(0, std::memory_order_release);
    S.generation++;
    S.next_window = 0;
    S.queue.clear();
    S.error.clear();
    S.group = g;
    S.channel = channel;
    S.used = true;
    g->slot_of_channel[static_cast<size_t>(channel)] = static_cast<int>(s);
    return static_cast<int>(s);
}


void Hip_Tracking_Runtime::detach(int slot)
{
    stop(slot);
    std::lock_guard<std::mutex> lk(d_mutex);
    if (slot < 0 || slot >= static_cast<int>(d_slots.size()) || !d_slots[slot]->used) return;
    Slot& S = *d_slots[slot];
    S.group->slot_of_channel[static_cast<size_t>(S.channel)] = -1;
    S.used = false;
    S.live_tracking.store(false, std::memory_order_release);
    S.generation++;
    d_lowest_next_window.store(lowest_next_window_locked(), std::memory_order_release);
}


bool Hip_Tracking_Runtime::start(int slot, const float* code, const float* data_code, int code_length, uint64_t nitems_read, double acq_delay_samples,
    double acq_doppler_hz, uint64_t acq_samplestamp_samples, int32_t* samples_offset, int32_t* first_prn_length)
{
    Group* g = nullptr;
    int channel = -1;
    {
        std::lock_guard<std::mutex> lk(d_mutex);
        if (slot < 0 || slot >= static_cast<int>(d_slots.size()) || !d_slots[slot]->used) return false;
        g = d_slots[slot]->group;
        channel = d_slots[slot]->channel;
    }
    // The handle's lock is held from the device-side start to the host-side bookkeeping: a launch of the group sees the channel either as
    // it was before (and its records, if any, are dropped by the generation check) or started with the slot ready to receive its records --
    // never the device running a channel whose slot does not know yet.
    std::lock_guard<std::mutex> hl(g->handle_mutex);  // waits for a launch of the group that is in flight
    struct Held_Off  // (the record watchdog of the group's other channels: this is the host holding the residencies off, not the device keeping quiet)
    {
        Group* g;
        explicit Held_Off(Group* gg) : g(gg) { g->held_off_ns.store(now_ns(), std::memory_order_release); }
        ~Held_Off() { g->held_off_ns.store(now_ns(), std::memory_order_release); }
    } held_off(g);
    if (g->begun) (void)end_and_file(g, nullptr);       // ... and one queued ahead comes in first: its records belong to the channels as they were
    if (d_live) quiesce_live(g);                        // residencies leave (the other channels' records stay in their rings; the next take brings a residency back)
    std::lock_guard<std::mutex> tl(d_slots[slot]->take_mutex);  // the block's own thread is not in the middle of a take of the old channel state
    int32_t offset = 0, first_len = 0;
    double acc0 = 0.0;
    std::string err;
    if (gsh_trk_pull_in(&g->conf, nitems_read, acq_delay_samples, acq_samplestamp_samples, acq_doppler_hz, &offset, &first_len, &acc0) != GSH_OK)
        err = std::string("gsh_trk_pull_in: ") + gsh_last_error();
    const uint64_t start_sample = nitems_read + static_cast<uint64_t>(std::max(offset, 0));
    // (the reference's pull-in latch is looked at in this very call, with the read pointer as it is now: behind the acquisition's stamp, the transitory is over at once)
    const uint32_t flags = gsh_trk_pull_in_over(&g->conf, nitems_read, acq_samplestamp_samples) ? GSH_TRK_START_PULL_IN_OVER : 0U;
    if (err.empty() && gsh_trk_start_flags(g->trk, channel, code, data_code, code_length, start_sample, acq_samplestamp_samples, acq_doppler_hz, acc0, flags) != GSH_OK)
        err = std::string("gsh_trk_start_flags: ") + gsh_last_error();
    std::lock_guard<std::mutex> lk(d_mutex);
    Slot& S = *d_slots[slot];
    S.generation++;
    S.queue.clear();
    S.error = err;
    S.tracking = err.empty();
    S.device_active = err.empty();
    S.starved_since_ns = 0;
    if (err.empty()) S.next_window = start_sample;
    S.live_next_window.store(start_sample, std::memory_order_release);
    S.live_tracking.store(err.empty(), std::memory_order_release);
    d_lowest_next_window.store(lowest_next_window_locked(), std::memory_order_release);
    if (!err.empty()) return false;
    if (samples_offset != nullptr) *samples_offset = offset;
    if (first_prn_length != nullptr) *first_prn_length = first_len;
    return true;
}


void Hip_Tracking_Runtime::stop(int slot)
{
    Group* g = nullptr;
    int channel = -1;
    {
        std::lock_guard<std::mutex> lk(d_mutex);
        if (slot < 0 || slot >= static_cast<int>(d_slots.size()) || !d_slots[slot]->used) return;
        Slot& S = *d_slots[slot];
        if (!S.tracking && !S.device_active)  // the device has stopped the channel itself (loss of lock) or it never ran: only the queue is left to drop
            {
                S.generation++;
                S.queue.clear();
                S.live_tracking.store(false, std::memory_order_release);
                return;
            }
        g = S.group;
        channel = S.channel;
    }
    std::lock_guard<std::mutex> hl(g->handle_mutex);  // as in start(): device state and slot change together, between two launches
    g->held_off_ns.store(now_ns(), std::memory_order_release);
    if (g->begun) (void)end_and_file(g, nullptr);
    if (d_live) quiesce_live(g);
    std::lock_guard<std::mutex> tl(d_slots[slot]->take_mutex);
    (void)gsh_trk_stop(g->trk, channel);
    g->held_off_ns.store(now_ns(), std::memory_order_release);
    std::lock_guard<std::mutex> lk(d_mutex);
    Slot& S = *d_slots[slot];
    S.tracking = false;
    S.device_active = false;
    S.live_tracking.store(false, std::memory_order_release);
    S.generation++;
    S.queue.clear();
    d_lowest_next_window.store(lowest_next_window_locked(), std::memory_order_release);
}


bool Hip_Tracking_Runtime::tracking(int slot) const
{
    std::lock_guard<std::mutex> lk(d_mutex);
    return slot >= 0 && slot < static_cast<int>(d_slots.size()) && d_slots[slot]->used && d_slots[slot]->tracking;
}


bool Hip_Tracking_Runtime::any_tracking() const
{
    std::lock_guard<std::mutex> lk(d_mutex);
    for (const auto& s : d_slots)
        if (s->used && s->tracking) return true;
    return false;
}


uint64_t Hip_Tracking_Runtime::lowest_next_window_locked() const
{
    uint64_t lowest = UINT64_MAX;
    for (const auto& s : d_slots)
        if (s->used && s->tracking) lowest = std::min(lowest, s->next_window);
    return lowest;
}


bool Hip_Tracking_Runtime::push(const std::complex<float>* samples, uint64_t first_index, uint64_t n, bool need_resident)
{
    if (!ok()) return false;
    if (first_index + n <= d_ring->next_index()) return true;  // somebody (another channel of the stream, an earlier call) has pushed them: all but the front-runner leave here
    // (kept up to date, under d_mutex, wherever a slot's next window or tracking flag changes: every block calls push in every general_work, and a scan of the
    // slots under the runtime's lock there is one more thing 32 threads queue up for.  Live mode: the blocks' takes move their windows without the lock; the
    // front-runner -- nobody else gets here -- scans their atomics)
    const uint64_t lowest = d_live ? lowest_next_window_live() : d_lowest_next_window.load(std::memory_order_acquire);
    // never push so far ahead that the window the slowest channel correlates next would be overwritten
    if (lowest != UINT64_MAX)
        {
            const uint64_t room_end = lowest + d_ring->capacity() - d_ring->max_window();
            if (first_index >= room_end) return true;  // nothing of this call fits yet; the block is offered the samples again
            n = std::min(n, room_end - first_index);
        }
    // a ring whose resident samples nobody is going to read may jump to the position of a caller that needs its own samples there
    const bool may_seek = need_resident && ((lowest == UINT64_MAX) || (lowest >= first_index));
    bool try_only = false;
    if (d_live && lowest != UINT64_MAX)
        {
            // Live mode, channels tracking: appending costs ~20 - 30 us of driver calls however little is appended, and every block of the stream is offered the
            // same new samples at the same moment.
            const uint64_t ring_next = d_ring->next_index();
            const uint64_t vlen = std::max<uint64_t>(d_min_vlen.load(std::memory_order_relaxed), 1);
            if (ring_next >= first_index)
                {
                    const uint64_t pending = first_index + n - ring_next;
                    const uint64_t runway = ring_next > lowest ? ring_next - lowest : 0;
                    // (2) a sliver is not worth the driver calls while the device still has periods in hand: the samples are offered again, with more behind them
                    if (pending < static_cast<uint64_t>(d_push_batch) * vlen && runway >= 3 * vlen) return true;
                    // (1) a small append by a block that finds the ring busy is left to the thread that is at it; behind a large one (several periods per call) the
                    // blocks queue up instead -- measured both ways, profiles/ab/r04/dropin_push_ab.txt
                    try_only = d_push_try && pending < 6 * vlen;
                    // (3) ... and not by the block that everybody is waiting for: the readers of a shared upstream buffer advance as fast as the slowest of them,
                    // and the slowest -- whose progress is what makes the scheduler offer new samples -- is the first to see them.  A sibling with periods in
                    // hand is microseconds behind; the 20 us of driver calls are better spent there.
                    if (try_only && d_push_spare_slowest && first_index < lowest + 2 * vlen && runway >= 3 * vlen) return true;
                }
        }
    uint64_t appended = 0, append_ns = 0;
    const bool ok_push = d_ring->push_from(first_index, samples, n, may_seek, std::chrono::milliseconds(need_resident ? 200 : 0), &appended, &append_ns, /*wait_copy=*/false, try_only);
    if (appended != 0)
        {
            d_push_ns.fetch_add(append_ns, std::memory_order_relaxed);
            d_pushed_samples.fetch_add(appended, std::memory_order_relaxed);
        }
    if (!ok_push)
        {
            std::lock_guard<std::mutex> lk(d_mutex);
            d_error = "sample ring: " + d_ring->last_error();
            return false;
        }
    return true;
}


int Hip_Tracking_Runtime::take(int slot, uint64_t limit_end, int max_records, gsh_trk_epoch* out)
{
    if (max_records <= 0 || out == nullptr) return 0;
    if (d_live)
        {
            if (slot < 0 || slot >= static_cast<int>(d_n_slots.load(std::memory_order_acquire))) return -1;
            return take_live(*d_slots[slot], limit_end, max_records, out);
        }
    std::unique_lock<std::mutex> lk(d_mutex);
    if (slot < 0 || slot >= static_cast<int>(d_slots.size()) || !d_slots[slot]->used) return -1;
    Slot& S = *d_slots[slot];
    bool waited = false;
    for (;;)
        {
            if (!S.error.empty()) return -1;
            Group* g = S.group;
            const uint64_t vlen = g->conf.vector_length;
            int n = 0;
            while (n < max_records && !S.queue.empty())
                {
                    const gsh_trk_epoch& r = S.queue.front();
                    if (!(r.flags & 2))
                        {
                            const uint64_t need = std::max<uint64_t>(vlen, static_cast<uint64_t>(std::max(r.prn_length_samples, 0)));
                            if (r.sample_counter + need > limit_end) break;  // the block has not been offered these samples itself yet
                        }
                    out[n++] = r;
                    const bool lost = (r.flags & 2) != 0;
                    S.queue.pop_front();
                    if (lost) break;
                }
            if (n > 0 && waited)
                {
                    d_stats.wake_ns += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - d_last_filed).count());
                    d_stats.wakes++;
                }
            if (n > 0 || !S.queue.empty() || !S.tracking) return n;
            // nothing filed for this channel: is its next window resident (and inside what the block itself has seen)?
            const uint64_t ring_next = d_ring->next_index();
            if (S.next_window + vlen > ring_next || S.next_window + vlen > limit_end) return 0;
            if (S.next_window < d_ring->oldest_index())
                {
                    // the channel fell more than the ring's capacity behind the stream: its samples are gone (the device would idle on it for ever)
                    S.error = "the channel's next window [" + std::to_string(S.next_window) + "..) is no longer resident (ring holds [" +
                              std::to_string(d_ring->oldest_index()) + ", " + std::to_string(ring_next) + "))";
                    S.tracking = false;
                    d_lowest_next_window.store(lowest_next_window_locked(), std::memory_order_release);
                    return -1;
                }
            if (g->in_flight)
                {
                    g->filed.wait(lk);  // the launch in flight may already cover this channel; look again when it has been filed
                    waited = true;
                    continue;
                }
            // ---- become the launcher for the whole group: end the launch that was queued ahead, or queue one and end it
            g->in_flight = true;
            lk.unlock();
            uint32_t filed = 0;
            bool failed = false;
            {
                std::lock_guard<std::mutex> hl(g->handle_mutex);  // start / stop of the group's channels happen between launches, never during one
                if (!g->begun) (void)begin_launch(g);
                const int ended_epochs = g->begun_epochs;
                uint64_t most = 0;
                if (g->begun)
                    filed = end_and_file(g, &most);
                else
                    failed = true;  // (begin_launch has filed the error into the slots)
                // launch-ahead: the blocks go through what has just been filed for a while; if the ring already holds a good part of another launch, let the
                // device start on it now
                if (d_launch_ahead && !failed && filed != 0 && most >= static_cast<uint64_t>(std::max(1, ended_epochs / 2))) (void)begin_launch(g);
            }
            lk.lock();
            g->in_flight = false;
            g->filed.notify_all();  // (those that found the group busy meanwhile)
            if (!failed && filed == 0 && S.queue.empty() && S.error.empty()) return 0;  // the device found nothing to do: do not spin on it
        }
}


// ------------------------------------------------------------------------------------------------ live mode
uint64_t Hip_Tracking_Runtime::lowest_next_window_live() const
{
    uint64_t lowest = UINT64_MAX;
    const size_t n = d_n_slots.load(std::memory_order_acquire);
    for (size_t i = 0; i < n; i++)
        {
            const Slot& S = *d_slots[i];
            if (S.live_tracking.load(std::memory_order_acquire)) lowest = std::min(lowest, S.live_next_window.load(std::memory_order_acquire));
        }
    return lowest;
}


// handle_mutex held
void Hip_Tracking_Runtime::quiesce_live(Group* g)
{
    if (gsh_trk_live_quiesce(g->trk) != GSH_OK)
        {
            std::lock_guard<std::mutex> lk(d_mutex);
            d_error = std::string("gsh_trk_live_quiesce: ") + gsh_last_error();
        }
}


// A block has work resident and no record: up to two residencies of the group's loop are kept queued (the second takes over when the first has used up
// its time on the device).  Whoever gets the handle does it; everybody else -- and anybody who comes while start / stop hold the handle -- just goes on polling.
void Hip_Tracking_Runtime::ensure_live(Group* g, bool wait_for_handle)
{
    std::unique_lock<std::mutex> hl(g->handle_mutex, std::defer_lock);
    if (wait_for_handle)
        hl.lock();
    else if (!hl.try_lock())
        return;  // somebody (start / stop, a sibling's check) has the handle right now.  (start / stop stamp held_off_ns themselves; a sibling's check must not:
                 // the channels of a group whose residency never reports would keep each other's watchdogs from ever firing)
    if (g->live_failed) return;
    int32_t n = 0;
    std::string err;
    if (gsh_trk_live_in_flight(g->trk, &n) != GSH_OK) err = std::string("gsh_trk_live_in_flight: ") + gsh_last_error();
    uint32_t begun = 0;
    while (err.empty() && n < 2)
        {
            // the ring's lock: the first residency registers the group with the ring (what pushes must keep off, the ring's live words), and pushes read that list
            std::lock_guard<std::mutex> rl(d_ring->mutex());
            if (gsh_trk_live_begin(g->trk) != GSH_OK)
                err = std::string("gsh_trk_live_begin: ") + gsh_last_error();
            else
                {
                    n++;
                    begun++;
                }
        }
    g->live_checked_ns.store(now_ns(), std::memory_order_release);
    if (begun != 0)
        {
            d_live_residencies.fetch_add(begun, std::memory_order_relaxed);
            std::lock_guard<std::mutex> lk(d_mutex);
            uint32_t serving = 0;
            for (const int sl : g->slot_of_channel)
                if (sl >= 0 && d_slots[sl]->tracking) serving++;
            d_stats.channels_served += static_cast<uint64_t>(begun) * serving;  // (a residency serves every channel of the group that is tracking)
        }
    if (!err.empty())
        {
            g->live_failed = true;
            std::lock_guard<std::mutex> lk(d_mutex);
            for (size_t c = 0; c < g->slot_of_channel.size(); c++)
                {
                    const int s = g->slot_of_channel[c];
                    if (s >= 0 && d_slots[s]->tracking) d_slots[s]->error = err;
                }
        }
}


int Hip_Tracking_Runtime::take_live(Slot& S, uint64_t limit_end, int max_records, gsh_trk_epoch* out)
{
    std::unique_lock<std::mutex> tl(S.take_mutex);  // (uncontended: start / stop of this very channel are the only other takers)
    Group* g = S.group;
    const uint64_t vlen = g->conf.vector_length;
    int64_t t_wait = 0, t_deadline = 0;
    bool asked_blocking = false;
    for (;;)
        {
            if (!S.live_tracking.load(std::memory_order_acquire))  // (looked at again after every wait: stop may have come in between)
                {
                    std::lock_guard<std::mutex> lk(d_mutex);
                    return (S.used && S.error.empty()) ? 0 : -1;
                }
            int32_t n = 0, pending = 0, active = 0, resident = 0;
            uint64_t nw = 0;
            if (gsh_trk_live_take(g->trk, S.channel, limit_end, max_records, out, &n, &pending, &nw, &active, &resident) != GSH_OK)
                {
                    std::lock_guard<std::mutex> lk(d_mutex);
                    S.error = std::string("gsh_trk_live_take: ") + gsh_last_error();
                    S.tracking = false;
                    S.live_tracking.store(false, std::memory_order_release);
                    return -1;
                }
            if (n > 0)
                {
                    S.starved_since_ns = 0;
                    S.live_next_window.store(nw, std::memory_order_release);
                    d_live_records.fetch_add(static_cast<uint64_t>(n), std::memory_order_relaxed);
                    if (t_wait != 0)
                        {
                            d_record_wait_ns.fetch_add(static_cast<uint64_t>(now_ns() - t_wait), std::memory_order_relaxed);
                            d_record_waits.fetch_add(1, std::memory_order_relaxed);
                        }
                    if (out[n - 1].flags & 2)  // loss of lock: the device has stopped the channel (trk.cc:2009-2014)
                        {
                            std::lock_guard<std::mutex> lk(d_mutex);
                            S.tracking = false;
                            S.device_active = false;
                            S.live_tracking.store(false, std::memory_order_release);
                        }
                    return n;
                }
            {
                std::lock_guard<std::mutex> lk(d_mutex);  // (rare path from here on: an error filed by ensure_live, or nothing to hand out)
                if (!S.error.empty()) return -1;
            }
            if (pending > 0 || !active)  // finished periods wait in the ring of records, but the block has not been offered their samples itself yet; or the device no
                {                        // longer advances the channel and there is no record left to say why: nothing will come
                    S.starved_since_ns = 0;
                    return 0;
                }
            const uint64_t ring_next = d_ring->next_index();
            if (nw + vlen > ring_next || nw + vlen > limit_end)  // the next window is not resident yet (or not the block's to consume)
                {
                    S.starved_since_ns = 0;
                    return 0;
                }
            if (nw < d_ring->oldest_index())
                {
                    std::lock_guard<std::mutex> lk(d_mutex);
                    S.error = "the channel's next window [" + std::to_string(nw) + "..) is no longer resident (ring holds [" + std::to_string(d_ring->oldest_index()) + ", " +
                              std::to_string(ring_next) + "))";
                    S.tracking = false;
                    S.live_tracking.store(false, std::memory_order_release);
                    return -1;
                }
            // The window is resident and its record is not there: a residency is working on it, or none is in flight.  Make sure of the latter now and then,
            // and look again: the device needs ~10 us per period.
            const int64_t now = now_ns();
            // (time in which start / stop of a sibling held the residencies off is not the device's silence: the clock starts again behind it)
            if (S.starved_since_ns == 0 || g->held_off_ns.load(std::memory_order_acquire) > S.starved_since_ns)
                S.starved_since_ns = now;
            else if (now - S.starved_since_ns > d_record_timeout_ns.load(std::memory_order_relaxed))
                {
                    // the window has been resident for a long time and the device has not delivered: a residency that never reports (a hung kernel, a lost device).  The
                    // block must not go on asking for ever -- the channel is given up the reference's way (the block publishes "events" 3, the FSM re-acquires)
                    std::lock_guard<std::mutex> lk(d_mutex);
                    S.error = "no record for the resident window [" + std::to_string(nw) + ", +" + std::to_string(vlen) + ") after " +
                              std::to_string((now - S.starved_since_ns) / 1000000) + " ms (" + (resident ? "a residency is running" : "no residency took it up") + ")";
                    S.tracking = false;
                    S.live_tracking.store(false, std::memory_order_release);
                    S.starved_since_ns = 0;
                    return -1;
                }
            if (t_wait == 0)
                {
                    t_wait = now;
                    t_deadline = now + 2000000;  // give up after 2 ms (start / stop of another channel hold the residencies off for a moment): the scheduler calls again
                }
            // (the channel's tail says whether its work-group is inside a residency right now: while it is, nobody needs to ask the driver -- 32 blocks polling
            // event states would get in the way of the thread that is appending samples; the look after a millisecond is a safety net only)
            if ((!resident || now - t_wait > 1000000) && now - g->live_checked_ns.load(std::memory_order_acquire) > 20000)
                {
                    const uint64_t before = d_live_residencies.load(std::memory_order_relaxed);
                    ensure_live(g);
                    // a residency queued just now -- the group's first sets up its host memory and loads the kernel, milliseconds -- gets its time: a block
                    // that comes back empty-handed although its samples are resident costs the scheduler a round trip (and call-for-call parity with the
                    // reference block, which the side-by-side tests hold it to)
                    if (d_live_residencies.load(std::memory_order_relaxed) != before) t_deadline = now_ns() + 50000000;
                }
            if (now >= t_deadline)
                {
                    if (!asked_blocking)
                        {
                            // nobody could make sure of a residency in all that time: start / stop of another channel hold the group's handle (they quiesce the
                            // residencies and restart a channel -- milliseconds, more with a slow engine).  Queue behind them this once, then give the device its time.
                            asked_blocking = true;
                            tl.unlock();  // (stop() of this channel takes the handle first and this slot's lock second: never wait for the handle with the slot held)
                            ensure_live(g, /*wait_for_handle=*/true);
                            tl.lock();
                            t_deadline = now_ns() + 50000000;
                            continue;
                        }
                    return 0;  // the scheduler calls again
                }
            // How a block waits for a record (profiles/ab/r05/dropin_wait_notes.txt).  Many periods per call: the records of a call arrive over tens of microseconds; polling
            // for 40 us and then dozing in 20 us steps costs the fewest wake-ups (2.9 M channel-periods/s at 20 per call against 2.4 M with the sleeping wait).  ONE period per
            // call, the reference's cadence (and up to four): the wait is ~30 us every time, for 32 block threads on the 16 CPUs the box's cgroup grants -- polling burns the
            // quota the appending thread needs, and the kernel's default timer slack turns a 20 us sleep into a 70 us one.  No polling, 25 us sleeps with 1 us of slack:
            // 0.86 - 0.89 M -> 1.50 M channel-periods/s.
            const bool single = max_records <= d_single_max_records;
            const int spin_us = single ? d_spin_us_single : d_spin_us;
            if (now - t_wait > static_cast<int64_t>(spin_us) * 1000)
                {
                    // (start / stop of this channel wait for the slot's lock with the group's handle held, and while they wait no sibling can make sure of a residency:
                    // the lock is not kept across the sleep.  Whatever changed meanwhile is looked at again at the top of the loop.)
                    tl.unlock();
                    if (d_timer_slack_ns > 0 && single)
                        {
                            // a 20 us sleep with the default 50 us timer slack is a 70 us sleep -- longer than the whole wait for a record.  Once per block thread.
                            static thread_local bool slack_set = false;
                            if (!slack_set)
                                {
                                    (void)prctl(PR_SET_TIMERSLACK, static_cast<unsigned long>(d_timer_slack_ns), 0, 0, 0);
                                    slack_set = true;
                                }
                        }
                    std::this_thread::sleep_for(std::chrono::microseconds(single ? d_sleep_us_single : d_sleep_us));
                    tl.lock();
                }
            else
                for (int k = 0; k < 64; k++) __builtin_ia32_pause();
        }
}


// handle_mutex held, d_mutex not
int Hip_Tracking_Runtime::begin_launch(Group* g)
{
    const uint64_t vlen = g->conf.vector_length;
    const auto t0 = std::chrono::steady_clock::now();
    int n_epochs = 1;
    g->begun_generation.assign(g->slot_of_channel.size(), 0);
    {
        // who takes part, and how far the newest sample lets the furthest-behind channel run: read with the handle locked, so that the
        // launch sees exactly the channels this snapshot describes
        std::lock_guard<std::mutex> lk(d_mutex);
        const uint64_t newest = d_ring->next_index();
        uint64_t most = 1;
        for (size_t c = 0; c < g->slot_of_channel.size(); c++)
            {
                const int s = g->slot_of_channel[c];
                if (s < 0) continue;
                const Slot& O = *d_slots[s];
                g->begun_generation[c] = O.generation;
                if (O.tracking && newest > O.next_window) most = std::max(most, (newest - O.next_window) / vlen);
            }
        n_epochs = static_cast<int>(std::min<uint64_t>(most, static_cast<uint64_t>(d_periods_per_launch)));
    }
    int rc;
    uint64_t ring_wait;
    {
        // pushes stay out only while the launch is queued (it reads the ring's newest index and files its reader fence)
        const auto t_ring = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> rl(d_ring->mutex());
        ring_wait = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_ring).count());
        rc = gsh_trk_run_begin(g->trk, n_epochs, 1);
    }
    const auto dt = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
    std::lock_guard<std::mutex> lk(d_mutex);
    d_stats.begin_ns += dt;
    d_stats.launch_ns += dt;
    d_stats.ring_wait_ns += ring_wait;
    if (rc != GSH_OK)
        {
            const std::string err = std::string("gsh_trk_run_begin: ") + gsh_last_error();
            for (size_t c = 0; c < g->slot_of_channel.size(); c++)
                {
                    const int s = g->slot_of_channel[c];
                    if (s >= 0 && d_slots[s]->generation == g->begun_generation[c] && d_slots[s]->tracking) d_slots[s]->error = err;
                }
            return 0;
        }
    g->begun = true;
    g->begun_epochs = n_epochs;
    return n_epochs;
}


// handle_mutex held, d_mutex not.  *most_resident: the most whole periods any channel of the group could run on what the ring holds once the records are filed.
uint32_t Hip_Tracking_Runtime::end_and_file(Group* g, uint64_t* most_resident)
{
    const uint64_t vlen = g->conf.vector_length;
    const int n_epochs = g->begun_epochs;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = gsh_trk_run_end(g->trk, g->records.data(), g->done.data());
    const std::string err = (rc != GSH_OK) ? std::string("gsh_trk_run_end: ") + gsh_last_error() : std::string();
    g->begun = false;
    const auto t_file = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> lk(d_mutex);
    d_stats.launch_ns += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(t_file - t0).count());
    const uint64_t newest = d_ring->next_index();
    uint64_t most = 0;
    uint32_t filed = 0, served = 0;
    for (size_t c = 0; c < g->slot_of_channel.size(); c++)
        {
            const int s = g->slot_of_channel[c];
            if (s < 0) continue;
            Slot& O = *d_slots[s];
            if (O.generation != g->begun_generation[c] || !O.tracking) continue;  // restarted / stopped while the launch ran
            if (rc != GSH_OK)
                {
                    O.error = err;
                    continue;
                }
            const int done = std::min(std::max(g->done[c], 0), n_epochs);
            for (int e = 0; e < done; e++)
                {
                    const gsh_trk_epoch& r = g->records[c * static_cast<size_t>(n_epochs) + static_cast<size_t>(e)];
                    O.queue.push_back(r);
                    filed++;
                    if (r.flags & 2)
                        {
                            O.tracking = false;  // loss of lock: the device has stopped the channel (trk.cc:2009-2014)
                            O.device_active = false;
                            break;
                        }
                    O.next_window = r.sample_counter + static_cast<uint64_t>(std::max(r.prn_length_samples, 0));
                }
            if (done > 0) served++;
            if (O.tracking && newest > O.next_window) most = std::max(most, (newest - O.next_window) / vlen);
        }
    if (most_resident != nullptr) *most_resident = most;
    d_lowest_next_window.store(lowest_next_window_locked(), std::memory_order_release);
    d_stats.file_ns += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_file).count());
    d_stats.launches++;
    d_stats.channel_periods += filed;
    d_stats.channels_served += served;
    d_stats.largest_launch = std::max(d_stats.largest_launch, filed);
    d_last_filed = std::chrono::steady_clock::now();
    g->filed.notify_all();
    return rc == GSH_OK ? filed : 0;
}


uint64_t Hip_Tracking_Runtime::next_window(int slot) const
{
    std::lock_guard<std::mutex> lk(d_mutex);
    if (slot < 0 || slot >= static_cast<int>(d_slots.size())) return 0;
    return d_live ? d_slots[slot]->live_next_window.load(std::memory_order_acquire) : d_slots[slot]->next_window;
}


std::string Hip_Tracking_Runtime::last_error(int slot) const
{
    std::lock_guard<std::mutex> lk(d_mutex);
    if (slot >= 0 && slot < static_cast<int>(d_slots.size()) && !d_slots[slot]->error.empty()) return d_slots[slot]->error;
    return d_error;
}


Hip_Tracking_Runtime::Stats Hip_Tracking_Runtime::stats() const
{
    std::lock_guard<std::mutex> lk(d_mutex);
    Stats s = d_stats;
    s.push_ns = d_push_ns.load(std::memory_order_relaxed);
    s.pushed_samples = d_pushed_samples.load(std::memory_order_relaxed);
    if (d_live)
        {
            s.residencies = d_live_residencies.load(std::memory_order_relaxed);
            s.launches += s.residencies;
            s.channel_periods += d_live_records.load(std::memory_order_relaxed);
            s.record_wait_ns = d_record_wait_ns.load(std::memory_order_relaxed);
            s.record_waits = d_record_waits.load(std::memory_order_relaxed);
        }
    return s;
}
