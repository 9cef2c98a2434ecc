/*!
 * \file hip_correlator_runtime.cc
 * \brief See the header.
 */
#include "hip_correlator_runtime.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace
{
// A timed wait on the STEADY clock (a wall-clock step from NTP or a manual set must not stretch or cut the 200 ms gap timeout); gcc 11's ThreadSanitizer does not
// understand pthread_cond_clockwait, which the steady-clock wait_for compiles to, so builds under TSAN take the system clock (round-5 review)
template <class Lock, class Rep, class Period, class Pred>
bool wait_with_timeout(std::condition_variable& cv, Lock& lk, const std::chrono::duration<Rep, Period>& timeout, Pred pred)
{
#if defined(__SANITIZE_THREAD__)
    return cv.wait_until(lk, std::chrono::system_clock::now() + timeout, pred);
#else
    return cv.wait_for(lk, timeout, pred);
#endif
}
}  // namespace

// ------------------------------------------------------------------------------------------------ Hip_Sample_Ring
Hip_Sample_Ring::Hip_Sample_Ring(int device, uint64_t capacity_samples, uint32_t max_window_samples)
    : d_device(device), d_capacity(capacity_samples), d_max_window(max_window_samples)
{
    d_devices.assign(1, device);
    if (gsh_stream_create(device, capacity_samples, max_window_samples, &d_handle) != GSH_OK)
        {
            set_error(gsh_last_error());
            d_handle = nullptr;
        }
}


Hip_Sample_Ring::Hip_Sample_Ring(const std::vector<int>& devices, uint64_t capacity_samples, uint32_t max_window_samples)
    : d_device(devices.empty() ? 0 : devices[0]), d_capacity(capacity_samples), d_max_window(max_window_samples)
{
    d_devices = devices;
    if (devices.size() <= 1)
        {
            if (gsh_stream_create(d_device, capacity_samples, max_window_samples, &d_handle) != GSH_OK)
                {
                    set_error(gsh_last_error());
                    d_handle = nullptr;
                }
            return;
        }
    // scatter + all-gather uses every xGMI link of the ingest GPU at once (csrc/stream_group.hip); GSH_GROUP_MODE=broadcast: ncclBroadcast
    const char* m = std::getenv("GSH_GROUP_MODE");
    const int mode = (m != nullptr && std::string(m) == "broadcast") ? GSH_GROUP_BROADCAST : GSH_GROUP_SCATTER_ALLGATHER;
    if (gsh_stream_group_create(devices.data(), static_cast<int>(devices.size()), capacity_samples, max_window_samples, mode, &d_group) != GSH_OK)
        {
            set_error(gsh_last_error());
            d_group = nullptr;
            return;
        }
    d_handle = gsh_stream_group_ring(d_group, 0);
}


Hip_Sample_Ring::~Hip_Sample_Ring()
{
    if (d_group != nullptr)
        gsh_stream_group_destroy(d_group);  // (the rings are the group's)
    else if (d_handle != nullptr)
        gsh_stream_destroy(d_handle);  // waits for the ring's queued copies
    for (void* p : d_registered) (void)gsh_host_unregister(p);
    for (auto& b : d_stage)
        if (!b.empty()) (void)gsh_host_unregister(b.data());
}


gsh_stream_t* Hip_Sample_Ring::handle_for(int device) const
{
    if (d_group == nullptr) return device == d_device ? d_handle : nullptr;
    for (size_t i = 0; i < d_devices.size(); i++)
        if (d_devices[i] == device) return gsh_stream_group_ring(d_group, static_cast<int>(i));
    return nullptr;
}


// d_mutex held.  Page-locked items: the DMA engine (or, in a group, the copy to the ingest GPU) reads them where they lie; the caller waits with
// wait_copied_upto before it hands the memory back.
int Hip_Sample_Ring::append_pinned(const std::complex<float>* items, uint64_t n, uint64_t* first)
{
    if (d_group == nullptr) return gsh_stream_push_pinned_async(d_handle, items, n, GSH_ITEM_GR_COMPLEX, 0, first);
    return gsh_stream_group_push(d_group, items, n, GSH_ITEM_GR_COMPLEX, 0, first);
}


// d_mutex held.  Anything else goes through a page-locked copy: the ring's own (gsh_stream_push_staged), or -- in a group -- one of four buffers here, each
// re-used once the append that read it last has reached the ring.
int Hip_Sample_Ring::append_pageable(const std::complex<float>* items, uint64_t n, uint64_t* first)
{
    if (d_group == nullptr) return gsh_stream_push_staged(d_handle, items, n, GSH_ITEM_GR_COMPLEX, 0, first);
    const int slot = d_stage_next;
    d_stage_next = (d_stage_next + 1) % NSTAGE;
    if (d_stage_end[slot] != 0)
        {
            const int rc = gsh_stream_wait_copied_upto(d_handle, d_stage_end[slot], nullptr);
            if (rc != GSH_OK) return rc;
        }
    if (d_stage[slot].size() < n)
        {
            if (!d_stage[slot].empty()) (void)gsh_host_unregister(d_stage[slot].data());
            d_stage[slot].assign(static_cast<size_t>(n + n / 2), std::complex<float>());
            (void)gsh_host_register(d_device, d_stage[slot].data(), d_stage[slot].size() * sizeof(std::complex<float>));  // (best effort: an unregistered buffer is copied by the runtime's own staging)
        }
    std::memcpy(d_stage[slot].data(), items, static_cast<size_t>(n) * sizeof(std::complex<float>));
    uint64_t at = 0;
    const int rc = gsh_stream_group_push(d_group, d_stage[slot].data(), n, GSH_ITEM_GR_COMPLEX, 0, &at);
    if (rc == GSH_OK)
        {
            d_stage_end[slot] = at + n;
            if (first != nullptr) *first = at;
        }
    return rc;
}


// end of the registered piece that holds address `at`, 0 when none does.  A DMA must lie inside ONE registration, so pieces are never merged.
uintptr_t Hip_Sample_Ring::piece_end_locked(uintptr_t at) const
{
    for (const auto& r : d_pinned)
        if (r.first <= at && at < r.second) return r.second;
    return 0;
}


// page-lock [a, b) (page aligned) minus what other pieces already hold; d_mutex held.  One registration per gap.
bool Hip_Sample_Ring::register_locked(uintptr_t a, uintptr_t b)
{
    std::vector<std::pair<uintptr_t, uintptr_t>> missing;
    uintptr_t at = a;
    for (const auto& r : d_pinned)  // sorted, disjoint
        {
            if (r.second <= at) continue;
            if (r.first >= b) break;
            if (r.first > at) missing.emplace_back(at, r.first);
            at = std::max(at, r.second);
        }
    if (at < b) missing.emplace_back(at, b);
    for (const auto& m : missing)
        {
            if (gsh_host_register(d_device, reinterpret_cast<void*>(m.first), static_cast<size_t>(m.second - m.first)) != GSH_OK)
                {
                    set_error(gsh_last_error());
                    return false;
                }
            d_registered.push_back(reinterpret_cast<void*>(m.first));
            d_pinned.emplace_back(m.first, m.second);
            std::sort(d_pinned.begin(), d_pinned.end());
        }
    return true;
}


bool Hip_Sample_Ring::register_host(const void* ptr, size_t bytes)
{
    if (d_handle == nullptr || ptr == nullptr || bytes == 0) return false;
    constexpr uintptr_t PAGE = 4096;
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr) & ~(PAGE - 1);
    const uintptr_t b = (reinterpret_cast<uintptr_t>(ptr) + bytes + PAGE - 1) & ~(PAGE - 1);
    std::lock_guard<std::mutex> lk(d_mutex);
    return register_locked(a, b);
}


uint64_t Hip_Sample_Ring::push_items(const void* items, uint64_t n, int item_type, bool inverted_spectrum)
{
    if (d_handle == nullptr) return UINT64_MAX;
    uint64_t first = 0;
    {
        std::lock_guard<std::mutex> lk(d_mutex);
        const int rc = d_group == nullptr ? gsh_stream_push(d_handle, items, n, item_type, inverted_spectrum ? 1 : 0, &first)
                                          : gsh_stream_group_push(d_group, items, n, item_type, inverted_spectrum ? 1 : 0, &first);
        if (rc != GSH_OK || (d_group != nullptr && gsh_stream_group_wait(d_group) != GSH_OK))  // (`items` is the caller's again on return)
            {
                set_error(gsh_last_error());
                return UINT64_MAX;
            }
        d_next.store(first + n, std::memory_order_release);
    }
    d_pushed.notify_all();
    return first;
}


bool Hip_Sample_Ring::push_from(uint64_t first_index, const std::complex<float>* samples, uint64_t n, bool may_seek, std::chrono::milliseconds gap_timeout,
    uint64_t* appended, uint64_t* append_ns, bool wait_copy, bool try_only)
{
    if (appended != nullptr) *appended = 0;
    if (append_ns != nullptr) *append_ns = 0;
    if (d_handle == nullptr) return false;
    if (n == 0) return true;
    bool dma_queued = false;
    const auto t_begin = std::chrono::steady_clock::now();
    {
        std::unique_lock<std::mutex> lk(d_mutex, std::defer_lock);
        if (try_only)
            {
                if (!lk.try_lock()) return true;
            }
        else
            lk.lock();
        uint64_t oldest = 0, next = 0;
        (void)gsh_stream_range(d_handle, &oldest, &next);
        if (first_index + n <= next) return true;  // somebody (another channel of the stream, an earlier call) has pushed them
        if (first_index > next)
            {
                if (next == oldest || may_seek)
                    {
                        // an empty ring starts wherever its first user is; a ring nobody reads any more follows the caller
                        const int rc_seek = seek_locked(first_index);
                        if (rc_seek == GSH_ERR_STATE && may_seek && next != oldest)
                            return true;  // a live tail is still winding down on the device (gsh_stream_seek refuses while one is active: the host's flags were cleared by a
                                          // watchdog or a failed take before the device's): not a ring error -- the block is offered its samples again (round-5 review)
                        if (rc_seek != GSH_OK)
                            {
                                set_error(gsh_last_error());
                                return false;
                            }
                        reposition_locked(first_index);
                        next = first_index;
                    }
                else
                    {
                        // the samples in between are in the input buffers of slower siblings: they push them when they get there
                        if (gap_timeout.count() <= 0) return true;  // a caller that does not need them resident itself (a block in standby) leaves it at that
                        // (wait_until on the system clock = pthread_cond_timedwait; wait_for is pthread_cond_clockwait, which gcc 11's ThreadSanitizer does not know releases the mutex)
                        const bool closed = wait_with_timeout(d_pushed, lk, gap_timeout, [&] { return d_next.load(std::memory_order_acquire) >= first_index; });
                        if (!closed)
                            {
                                set_error("push_from: samples " + std::to_string(first_index) + ".. leave a gap after the ring's " + std::to_string(next) +
                                          " that no other channel of the stream has closed");
                                return false;
                            }
                        (void)gsh_stream_range(d_handle, &oldest, &next);
                        if (first_index + n <= next) return true;
                    }
            }
        const uint64_t skip = next - first_index;
        // a call may offer more than the ring holds: the newest capacity's worth is all that can stay resident anyway
        uint64_t count = n - skip, from = skip;
        if (count > d_capacity)
            {
                from += count - d_capacity;
                if (seek_locked(first_index + from) != GSH_OK)
                    {
                        set_error(gsh_last_error());
                        return false;
                    }
                d_origin.store(first_index + from, std::memory_order_release);
                count = d_capacity;
            }
        // Page-locked input (registered by the caller, or here when auto-registration is on) goes to the DMA engine as it lies, one push per
        // registered piece it spans; anything else through the ring's page-locked staging buffers.
        int rc_push = GSH_OK;
        uint64_t done_items = 0, first = 0, first_piece = 0;
        while (done_items < count && rc_push == GSH_OK)
            {
                const std::complex<float>* cur = samples + from + done_items;
                const uintptr_t at = reinterpret_cast<uintptr_t>(cur);
                const uint64_t left = count - done_items;
                uintptr_t end = piece_end_locked(at);
                if (end == 0 && d_auto_register)
                    {
                        // a new stretch of the caller's buffer: page-lock what this call shows, and nothing beyond it -- the addresses behind it may belong to
                        // another mapping altogether.  A scheduler's buffer is the same memory for the whole run: after its first pass everything is registered.
                        constexpr uintptr_t PAGE = 4096;
                        const uintptr_t lo = at & ~(PAGE - 1), need_hi = (at + left * sizeof(std::complex<float>) + PAGE - 1) & ~(PAGE - 1);
                        uintptr_t hi = need_hi;
                        for (const auto& r : d_pinned)
                            if (r.first > at) hi = std::min(hi, r.first);  // up to the next piece
                        if (!register_locked(lo, std::max(hi, lo + PAGE))) d_auto_register = false;  // memory that cannot be registered: stop trying, use the staging copy
                        end = piece_end_locked(at);
                    }
                uint64_t chunk = left;
                if (end != 0)
                    {
                        chunk = std::min<uint64_t>(left, (end - at) / sizeof(std::complex<float>));
                        if (chunk == 0) end = 0;  // a sample straddles the piece's last bytes: copy it through staging with the rest
                    }
                if (end != 0)
                    {
                        // queued only: the wait for the DMA happens below, with the ring's lock released
                        rc_push = append_pinned(cur, chunk, &first_piece);
                        dma_queued = dma_queued || rc_push == GSH_OK;
                    }
                else
                    {
                        // not page-locked: up to the next registered piece (or everything) through staging
                        chunk = left;
                        for (const auto& r : d_pinned)
                            if (r.first > at) { chunk = std::min<uint64_t>(chunk, (r.first - at + sizeof(std::complex<float>) - 1) / sizeof(std::complex<float>)); break; }
                        rc_push = append_pageable(cur, chunk, &first_piece);
                    }
                if (done_items == 0) first = first_piece;
                if (rc_push == GSH_OK) done_items += chunk;
            }
        if (appended != nullptr) *appended = done_items;
        if (done_items > 0) d_next.store(first + done_items, std::memory_order_release);
        if (rc_push != GSH_OK)
            {
                set_error(gsh_last_error());
                d_pushed.notify_all();
                if (dma_queued) (void)gsh_stream_wait_copied(d_handle);
                return false;
            }
    }
    d_pushed.notify_all();
    // The caller gets its buffer back when the DMA engine has read it.  The ring's lock is free meanwhile: a launch that reads the ring can be
    // queued (it waits for the push on the device, not here) and the siblings' index comparisons go through.
    bool ok_wait = true;
    if (dma_queued && wait_copy && gsh_stream_wait_copied(d_handle) != GSH_OK)
        {
            std::lock_guard<std::mutex> lk(d_mutex);
            set_error(gsh_last_error());
            ok_wait = false;
        }
    if (append_ns != nullptr) *append_ns = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count());
    return ok_wait;
}


bool Hip_Sample_Ring::wait_copied_upto(uint64_t end)
{
    if (d_handle == nullptr) return false;
    if (end <= d_copied_upto.load(std::memory_order_acquire)) return true;  // the usual case: copied long ago
    if (end > d_next.load(std::memory_order_acquire)) end = d_next.load(std::memory_order_acquire);  // what has not been pushed has not been handed to the DMA engine either
    uint64_t upto = 0;
    if (gsh_stream_wait_copied_upto(d_handle, end, &upto) != GSH_OK)
        {
            std::lock_guard<std::mutex> lk(d_mutex);
            set_error(gsh_last_error());
            return false;
        }
    upto = std::max(upto, end);
    uint64_t seen = d_copied_upto.load(std::memory_order_relaxed);
    while (seen < upto && !d_copied_upto.compare_exchange_weak(seen, upto, std::memory_order_release)) {}
    return true;
}


// position every ring of the stream (d_mutex held)
int Hip_Sample_Ring::seek_locked(uint64_t next_index)
{
    if (d_group == nullptr) return gsh_stream_seek(d_handle, next_index);
    for (int i = 0; i < gsh_stream_group_size(d_group); i++)
        {
            const int rc = gsh_stream_seek(gsh_stream_group_ring(d_group, i), next_index);
            if (rc != GSH_OK) return rc;
        }
    for (uint64_t& e : d_stage_end) e = 0;  // (a seek waits for everything queued)
    return GSH_OK;
}


// after a successful seek (gsh_stream_seek has synchronised the ring's stream: nothing is in flight): what "has left the callers' buffers" means starts afresh at the
// new position -- a high-water mark kept from before a BACKWARD seek would let give_back() hand memory back while a DMA out of it is still queued
void Hip_Sample_Ring::reposition_locked(uint64_t next_index)
{
    d_next.store(next_index, std::memory_order_release);
    d_origin.store(next_index, std::memory_order_release);
    d_copied_upto.store(next_index, std::memory_order_release);
}


bool Hip_Sample_Ring::seek(uint64_t next_index)
{
    if (d_handle == nullptr) return false;
    std::lock_guard<std::mutex> lk(d_mutex);
    if (seek_locked(next_index) != GSH_OK)
        {
            set_error(gsh_last_error());
            return false;
        }
    reposition_locked(next_index);
    return true;
}


uint64_t Hip_Sample_Ring::push(const std::complex<float>* samples, uint64_t n, bool inverted_spectrum)
{
    return push_items(samples, n, GSH_ITEM_GR_COMPLEX, inverted_spectrum);
}


uint64_t Hip_Sample_Ring::push_ishort(const int16_t* iq, uint64_t n, bool inverted_spectrum)
{
    return push_items(iq, n, GSH_ITEM_SHORT, inverted_spectrum);
}


uint64_t Hip_Sample_Ring::push_ibyte(const int8_t* iq, uint64_t n, bool inverted_spectrum)
{
    return push_items(iq, n, GSH_ITEM_BYTE, inverted_spectrum);
}


void Hip_Sample_Ring::range(uint64_t* oldest, uint64_t* next) const
{
    std::lock_guard<std::mutex> lk(d_mutex);
    uint64_t lo = 0, hi = 0;
    if (d_handle != nullptr) (void)gsh_stream_range(d_handle, &lo, &hi);
    if (oldest) *oldest = lo;
    if (next) *next = hi;
}


bool Hip_Sample_Ring::wait_for(uint64_t end, std::chrono::milliseconds timeout) const
{
    if (d_next.load(std::memory_order_acquire) >= end) return true;  // the usual case: no lock, no convoy behind a running launch
    std::unique_lock<std::mutex> lk(d_mutex);
    return wait_with_timeout(d_pushed, lk, timeout, [&] { return d_next.load(std::memory_order_acquire) >= end; });
}


// ------------------------------------------------------------------------------------------------ Hip_Correlator_Runtime
Hip_Correlator_Runtime::Hip_Correlator_Runtime(Hip_Sample_Ring* ring, int max_channels, int max_code_length, std::chrono::microseconds max_wait, int spin_us)
    : d_ring(ring), d_max_wait(max_wait), d_current(std::make_shared<Batch>())
{
    d_current->jobs.reserve(static_cast<size_t>(std::max(max_channels, 1)));
    d_spin_us = spin_us >= 0 ? spin_us : (static_cast<int>(std::thread::hardware_concurrency()) >= 2 * max_channels ? 150 : 0);
    if (ring == nullptr || !ring->ok())
        {
            d_error = "no sample ring";
            return;
        }
    if (gsh_bank_create(ring->device(), max_channels, max_code_length, &d_bank) != GSH_OK || gsh_bank_set_stream_ring(d_bank, ring->handle()) != GSH_OK)
        {
            d_error = gsh_last_error();
            if (d_bank != nullptr) gsh_bank_destroy(d_bank);
            d_bank = nullptr;
            return;
        }
    d_slot_used.assign(static_cast<size_t>(max_channels), 0);
}


Hip_Correlator_Runtime::~Hip_Correlator_Runtime()
{
    if (d_bank != nullptr) gsh_bank_destroy(d_bank);
}


int Hip_Correlator_Runtime::register_channel()
{
    std::lock_guard<std::mutex> lk(d_mutex);
    for (size_t i = 0; i < d_slot_used.size(); i++)
        if (!d_slot_used[i])
            {
                d_slot_used[i] = 1;
                d_active++;
                return static_cast<int>(i);
            }
    return -1;
}


void Hip_Correlator_Runtime::unregister_channel(int channel)
{
    {
        std::lock_guard<std::mutex> lk(d_mutex);
        if (channel < 0 || channel >= static_cast<int>(d_slot_used.size()) || !d_slot_used[channel]) return;
        d_slot_used[channel] = 0;
        d_active--;
    }
    d_arrived.notify_all();  // a leader waiting for this channel must re-evaluate
}


bool Hip_Correlator_Runtime::set_code(int channel, const float* code, int code_length)
{
    if (d_bank == nullptr) return false;
    std::lock_guard<std::mutex> bl(d_bank_mutex);
    if (gsh_bank_set_code(d_bank, channel, code, code_length) != GSH_OK)
        {
            std::lock_guard<std::mutex> lk(d_mutex);
            d_error = gsh_last_error();
            return false;
        }
    return true;
}


std::shared_ptr<Hip_Correlator_Runtime::Batch> Hip_Correlator_Runtime::new_batch() const
{
    auto b = std::make_shared<Batch>();
    b->jobs.reserve(d_slot_used.size());  // arrivals never reallocate while they hold d_mutex
    return b;
}


// launch one closed batch and publish its results (called by exactly one thread per batch, without d_mutex)
void Hip_Correlator_Runtime::run_batch(const std::shared_ptr<Batch>& b, bool timed_out)
{
    const int n = static_cast<int>(b->jobs.size());
    b->out.assign(static_cast<size_t>(n) * GSH_MAX_TAPS * 2, 0.0F);
    int rc = GSH_OK;
    std::string err;
    {
        std::lock_guard<std::mutex> bl(d_bank_mutex);
        // pushes must not move the ring's residency window between the job translation and the launch
        std::lock_guard<std::mutex> rl(d_ring->d_mutex);
        // one launch per correlator flavour present in the batch (a launch shares one kernel specialisation; channels in
        // high-dynamics mode, trk.cc:675, are batched separately from the standard ones)
        bool uniform = true;
        for (int i = 1; i < n; i++) uniform = uniform && (b->jobs[i].high_dyn == b->jobs[0].high_dyn);
        if (uniform)
            {
                rc = gsh_bank_correlate(d_bank, b->jobs.data(), n, b->out.data());
            }
        else
            {
                std::vector<gsh_corr_job> part;
                std::vector<int> where;
                std::vector<float> part_out;
                for (int mode = 0; mode <= 2 && rc == GSH_OK; mode++)
                    {
                        part.clear();
                        where.clear();
                        for (int i = 0; i < n; i++)
                            if (b->jobs[i].high_dyn == mode)
                                {
                                    part.push_back(b->jobs[i]);
                                    where.push_back(i);
                                }
                        if (part.empty()) continue;
                        part_out.assign(part.size() * GSH_MAX_TAPS * 2, 0.0F);
                        rc = gsh_bank_correlate(d_bank, part.data(), static_cast<int>(part.size()), part_out.data());
                        for (size_t k = 0; k < where.size() && rc == GSH_OK; k++)
                            std::memcpy(&b->out[static_cast<size_t>(where[k]) * GSH_MAX_TAPS * 2], &part_out[k * GSH_MAX_TAPS * 2], sizeof(float) * GSH_MAX_TAPS * 2);
                    }
            }
        if (rc != GSH_OK) err = gsh_last_error();
    }
    {
        std::lock_guard<std::mutex> lk(d_mutex);
        d_stats.batches++;
        d_stats.jobs += static_cast<uint64_t>(n);
        d_stats.timeouts += timed_out ? 1u : 0u;
        d_stats.largest_batch = std::max<uint32_t>(d_stats.largest_batch, static_cast<uint32_t>(n));
    }
    b->status = rc;
    b->error = err;
    {
        std::lock_guard<std::mutex> bl(b->m);
        b->done.store(1, std::memory_order_release);
    }
    b->done_cv.notify_all();
}


bool Hip_Correlator_Runtime::correlate(int channel, const gsh_corr_job& job_in, std::complex<float>* out)
{
    const gsh_corr_job* jobs[1] = {&job_in};
    std::complex<float>* outs[1] = {out};
    return correlate_n(1, &channel, jobs, outs);
}


bool Hip_Correlator_Runtime::correlate_pair(int channel, const gsh_corr_job& job, std::complex<float>* out, int channel2, const gsh_corr_job& job2, std::complex<float>* out2)
{
    const int channels[2] = {channel, channel2};
    const gsh_corr_job* jobs[2] = {&job, &job2};
    std::complex<float>* outs[2] = {out, out2};
    return correlate_n(2, channels, jobs, outs);
}


// the jobs of ONE caller join the current batch side by side; a caller that brings two counts as two arrivals
bool Hip_Correlator_Runtime::correlate_n(int n_in, const int* channels, const gsh_corr_job* const* jobs_in, std::complex<float>* const* outs)
{
    if (d_bank == nullptr) return false;
    for (int k = 0; k < n_in; k++)
        if (outs[k] == nullptr) return false;
    std::shared_ptr<Batch> b;
    size_t idx;
    bool close_now = false, first = false;
    {
        std::unique_lock<std::mutex> lk(d_mutex);
        b = d_current;
        idx = b->jobs.size();
        for (int k = 0; k < n_in; k++)
            {
                b->jobs.push_back(*jobs_in[k]);
                b->jobs.back().code_slot = channels[k];
            }
        first = (idx == 0);
        if (static_cast<int>(b->jobs.size()) >= d_active)
            {
                // this arrival completes the batch: close it and launch from this very thread
                b->taken = true;
                d_current = new_batch();  // later arrivals start the next batch
                close_now = true;
            }
        else if (first)
            {
                // first arriver: bounded wait for the channels that are tracking right now; whoever completes the batch takes it,
                // otherwise this thread closes it with what has arrived
                const auto deadline = std::chrono::steady_clock::now() + d_max_wait;
                while (!b->taken && static_cast<int>(b->jobs.size()) < d_active)
                    {
                        if (d_arrived.wait_until(lk, deadline) == std::cv_status::timeout) break;
                    }
                if (!b->taken)
                    {
                        b->taken = true;
                        if (d_current == b) d_current = new_batch();
                        close_now = true;
                        lk.unlock();
                        run_batch(b, static_cast<int>(b->jobs.size()) < d_active);
                        close_now = false;  // already run
                    }
            }
    }
    if (close_now)
        {
            d_arrived.notify_all();  // the first arriver (if it is not this thread) stops its timed wait
            run_batch(b, false);
        }
    // wait for the batch's results: poll the flag for a short while (the launch is tens of microseconds), then block
    if (b->done.load(std::memory_order_acquire) == 0 && d_spin_us > 0)
        {
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(d_spin_us);
            while (b->done.load(std::memory_order_acquire) == 0 && std::chrono::steady_clock::now() < until)
                {
#if defined(__x86_64__) || defined(__i386__)
                    __builtin_ia32_pause();
#else
                    std::this_thread::yield();
#endif
                }
        }
    if (b->done.load(std::memory_order_acquire) == 0)
        {
            std::unique_lock<std::mutex> bl(b->m);
            b->done_cv.wait(bl, [&] { return b->done.load(std::memory_order_acquire) != 0; });
        }
    if (b->status != GSH_OK)
        {
            std::lock_guard<std::mutex> lk(d_mutex);
            d_error = b->error;
            return false;
        }
    for (int k = 0; k < n_in; k++)
        {
            const int taps = std::min(std::max(jobs_in[k]->n_taps, 0), GSH_MAX_TAPS);
            for (int t = 0; t < taps; t++)
                outs[k][t] = std::complex<float>(b->out[((idx + k) * GSH_MAX_TAPS + t) * 2], b->out[((idx + k) * GSH_MAX_TAPS + t) * 2 + 1]);
        }
    return true;
}


Hip_Correlator_Runtime::Stats Hip_Correlator_Runtime::stats() const
{
    std::lock_guard<std::mutex> lk(d_mutex);
    return d_stats;
}


// ------------------------------------------------------------------------------------------------ Hip_Multicorrelator_Batched
bool Hip_Multicorrelator_Batched::init(int max_signal_length_samples, int n_correlators)
{
    if (d_runtime == nullptr || !d_runtime->ok())
        {
            d_error = d_runtime ? d_runtime->last_error() : "no runtime";
            return false;
        }
    if (n_correlators < 1 || n_correlators > GSH_MAX_TAPS || max_signal_length_samples < 1)
        {
            d_error = "init: n_correlators outside 1..GSH_MAX_TAPS or empty signal length";
            return false;
        }
    if (d_channel < 0) d_channel = d_runtime->register_channel();
    if (d_channel < 0)
        {
            d_error = "init: the runtime has no free channel slot";
            return false;
        }
    d_max_len = max_signal_length_samples;
    d_n_correlators = n_correlators;
    return true;
}


bool Hip_Multicorrelator_Batched::set_local_code_and_taps(int code_length_chips, const float* local_code_in, float* shifts_chips)
{
    if (d_channel < 0 || local_code_in == nullptr || shifts_chips == nullptr)
        {
            d_error = "set_local_code_and_taps before init, or null argument";
            return false;
        }
    d_shifts = shifts_chips;
    if (!d_runtime->set_code(d_channel, local_code_in, code_length_chips))
        {
            d_error = d_runtime->last_error();
            return false;
        }
    return true;
}


bool Hip_Multicorrelator_Batched::set_input_output_vectors(std::complex<float>* corr_out, const std::complex<float>* /*sig_in*/)
{
    d_corr_out = corr_out;
    return corr_out != nullptr;
}


bool Hip_Multicorrelator_Batched::run(int mode, float rem_carr, float phase_step, float phase_rate, float rem_code, float code_step, float code_rate, int n)
{
    if (d_channel < 0 || d_shifts == nullptr || d_corr_out == nullptr)
        {
            d_error = "correlate before init / set_local_code_and_taps / set_input_output_vectors";
            return false;
        }
    if (n < 1 || n > d_max_len)
        {
            d_error = "signal_length_samples outside 1..init size";
            return false;
        }
    gsh_corr_job j;
    fill_job(&j, mode, rem_carr, phase_step, phase_rate, rem_code, code_step, code_rate, n);
    if (d_served)
        {
            // a leader (set_companion) computed this very call while it held the window: same window, parameters and taps -> done
            d_served = false;
            if (std::memcmp(&j, &d_served_job, sizeof(j)) == 0) return true;
        }
    Hip_Multicorrelator_Batched* cmp = d_companion;
    if (cmp != nullptr && cmp->ready() && cmp->d_runtime == d_runtime && mode == 0 && n <= cmp->d_max_len)
        {
            gsh_corr_job j2;
            cmp->fill_job(&j2, mode, rem_carr, phase_step, phase_rate, rem_code, code_step, code_rate, n);
            j2.sample_offset = d_sample_index;  // the companion reads the window this call reads (trk.cc:1246-1256 passes the same pointer)
            if (!d_runtime->correlate_pair(d_channel, j, d_corr_out, cmp->d_channel, j2, cmp->d_corr_out))
                {
                    d_error = d_runtime->last_error();
                    return false;
                }
            cmp->d_served = true;
            cmp->d_served_job = j2;
            return true;
        }
    if (!d_runtime->correlate(d_channel, j, d_corr_out))
        {
            d_error = d_runtime->last_error();
            return false;
        }
    return true;
}


void Hip_Multicorrelator_Batched::fill_job(gsh_corr_job* j, int mode, float rem_carr, float phase_step, float phase_rate, float rem_code, float code_step, float code_rate,
    int n) const
{
    std::memset(j, 0, sizeof(*j));
    j->sample_offset = d_sample_index;
    j->n_samples = n;
    j->rem_carr_phase_rad = rem_carr;
    j->phase_step_rad = phase_step;
    j->phase_rate_step_rad = phase_rate;
    j->rem_code_phase_chips = rem_code;
    j->code_phase_step_chips = code_step;
    j->code_phase_rate_step_chips = code_rate;
    j->n_taps = d_n_correlators;
    j->high_dyn = mode;
    for (int t = 0; t < d_n_correlators; t++) j->shifts_chips[t] = d_shifts[t];  // re-read every call: the caller mutates them in place (trk.cc:2132-2146)
}


bool Hip_Multicorrelator_Batched::Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float phase_rate_step_rad,
    float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips, int signal_length_samples)
{
    // mcorr.cc:103-126: the flag picks the high-dynamics resampler + rotator pair
    return run(d_use_high_dynamics_resampler ? 1 : 0, rem_carrier_phase_in_rad, phase_step_rad, phase_rate_step_rad, rem_code_phase_chips, code_phase_step_chips,
        code_phase_rate_step_chips, signal_length_samples);
}


bool Hip_Multicorrelator_Batched::Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float rem_code_phase_chips,
    float code_phase_step_chips, float code_phase_rate_step_chips, int signal_length_samples)
{
    // mcorr.cc:129-144: the six-argument overload keeps the standard rotator whatever the flag says
    return run(d_use_high_dynamics_resampler ? 2 : 0, rem_carrier_phase_in_rad, phase_step_rad, 0.0F, rem_code_phase_chips, code_phase_step_chips,
        code_phase_rate_step_chips, signal_length_samples);
}


bool Hip_Multicorrelator_Batched::free()
{
    if (d_channel >= 0 && d_runtime != nullptr) d_runtime->unregister_channel(d_channel);
    d_channel = -1;
    d_shifts = nullptr;
    d_corr_out = nullptr;
    return true;
}
