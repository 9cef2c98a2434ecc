/*!
 * \file hip_tracking_runtime.cc
 * \brief Channel-batching runtime behind the tracking blocks; see hip_tracking_runtime.h.
 */
#include "hip_tracking_runtime.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

Hip_Tracking_Runtime::Hip_Tracking_Runtime(int device, std::shared_ptr<Hip_Sample_Ring> ring, int periods_per_launch, int channels_per_group)
    : d_device(device),
      d_ring(std::move(ring)),
      d_periods_per_launch(std::min(std::max(periods_per_launch, 1), 256)),
      d_channels_per_group(std::min(std::max(channels_per_group, 1), 4096))
{
    if (const char* e = std::getenv("GSH_TRK_LAUNCH_AHEAD")) d_launch_ahead = (std::atoi(e) != 0);
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
    if (gsh_trk_set_stream_ring(g->trk, d_ring->handle()) != GSH_OK)
        {
            d_error = std::string("gsh_trk_set_stream_ring: ") + gsh_last_error();
            gsh_trk_destroy(g->trk);
            return nullptr;
        }
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
    if (s == d_slots.size()) d_slots.push_back(std::make_unique<Slot>());
    Slot& S = *d_slots[s];
    S = Slot{};
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
    if (g->begun) (void)end_and_file(g, nullptr);       // ... and one queued ahead comes in first: its records belong to the channels as they were
    int32_t offset = 0, first_len = 0;
    double acc0 = 0.0;
    std::string err;
    if (gsh_trk_pull_in(&g->conf, nitems_read, acq_delay_samples, acq_samplestamp_samples, acq_doppler_hz, &offset, &first_len, &acc0) != GSH_OK)
        err = std::string("gsh_trk_pull_in: ") + gsh_last_error();
    const uint64_t start_sample = nitems_read + static_cast<uint64_t>(std::max(offset, 0));
    if (err.empty() && gsh_trk_start_ex(g->trk, channel, code, data_code, code_length, start_sample, acq_samplestamp_samples, acq_doppler_hz, acc0) != GSH_OK)
        err = std::string("gsh_trk_start_ex: ") + gsh_last_error();
    std::lock_guard<std::mutex> lk(d_mutex);
    Slot& S = *d_slots[slot];
    S.generation++;
    S.queue.clear();
    S.error = err;
    S.tracking = err.empty();
    if (err.empty()) S.next_window = start_sample;
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
        if (!S.tracking)  // the device has stopped the channel itself (loss of lock) or it never ran: only the queue is left to drop
            {
                S.generation++;
                S.queue.clear();
                return;
            }
        g = S.group;
        channel = S.channel;
    }
    std::lock_guard<std::mutex> hl(g->handle_mutex);  // as in start(): device state and slot change together, between two launches
    if (g->begun) (void)end_and_file(g, nullptr);
    (void)gsh_trk_stop(g->trk, channel);
    std::lock_guard<std::mutex> lk(d_mutex);
    Slot& S = *d_slots[slot];
    S.tracking = false;
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
    // (kept up to date, under d_mutex, wherever a slot's next window or tracking flag changes: every block calls push in every general_work, and a scan of the
    // slots under the runtime's lock there is one more thing 32 threads queue up for)
    const uint64_t lowest = d_lowest_next_window.load(std::memory_order_acquire);
    // never push so far ahead that the window the slowest channel correlates next would be overwritten
    if (lowest != UINT64_MAX)
        {
            const uint64_t room_end = lowest + d_ring->capacity() - d_ring->max_window();
            if (first_index >= room_end) return true;  // nothing of this call fits yet; the block is offered the samples again
            n = std::min(n, room_end - first_index);
        }
    // a ring whose resident samples nobody is going to read may jump to the position of a caller that needs its own samples there
    const bool may_seek = need_resident && ((lowest == UINT64_MAX) || (lowest >= first_index));
    uint64_t appended = 0, append_ns = 0;
    const bool ok_push = d_ring->push_from(first_index, samples, n, may_seek, std::chrono::milliseconds(need_resident ? 200 : 0), &appended, &append_ns);
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
    return d_slots[slot]->next_window;
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
    return s;
}
