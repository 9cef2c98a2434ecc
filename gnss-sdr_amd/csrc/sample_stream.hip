// gsh_stream_*: the IF sample stream of one front-end as a ring in device memory, addressed by absolute sample index.
// See include/gnss_sdr_hip.h for the contract and the reference pieces it stands in for.
#include "sample_stream.h"
#include "sample_convert.h"
#include <algorithm>
#include <cstdlib>
#include <new>

namespace gsh
{
int stream_window(const gsh_stream* s, unsigned long long index, unsigned long long n, const float2** ptr)
{
    if (n > s->max_window) return set_error(GSH_ERR_INVALID, "window of %llu samples exceeds the ring's max_window_samples %llu", n, s->max_window);
    if (index < stream_oldest(s) || index + n > s->next)
        return set_error(GSH_ERR_INVALID, "samples [%llu, %llu) are not resident (ring holds [%llu, %llu))", index, index + n, stream_oldest(s), s->next);
    *ptr = s->d_ring + (index % s->capacity);
    return GSH_OK;
}
int stream_mark_read(gsh_stream* s, unsigned long long min_index, hipStream_t st)
{
    const int slot = s->read_count % gsh_stream::HIST;
    if (s->read_ev[slot] == nullptr) GSH_HIP(hipEventCreateWithFlags(&s->read_ev[slot], hipEventDisableTiming));
    if (s->read_count >= gsh_stream::HIST)
        {
            // the slot still holds the fence of a launch 16 records back: fold it into the ring's stream instead of dropping it
            if (s->read_fold == nullptr) GSH_HIP(hipEventCreateWithFlags(&s->read_fold, hipEventDisableTiming));
            GSH_HIP(hipStreamWaitEvent(s->stream, s->read_ev[slot], 0));
            GSH_HIP(hipEventRecord(s->read_fold, s->stream));
            s->has_fold = true;
        }
    GSH_HIP(hipEventRecord(s->read_ev[slot], st));
    s->read_min[slot] = min_index;
    s->read_count++;
    return GSH_OK;
}

int stream_wait_pushed(gsh_stream* s, unsigned long long need_end, hipStream_t st)
{
    const int n = s->push_count < gsh_stream::HIST ? s->push_count : gsh_stream::HIST;
    // oldest recorded push whose end covers need_end (ends grow with the push count)
    for (int k = n; k >= 1; k--)
        {
            const int slot = (s->push_count - k) % gsh_stream::HIST;
            if (s->push_end[slot] >= need_end)
                {
                    GSH_HIP(hipStreamWaitEvent(st, s->push_ev[slot], 0));
                    return GSH_OK;
                }
        }
    if (s->pushed != nullptr) GSH_HIP(hipStreamWaitEvent(st, s->pushed, 0));  // not in the history (or "everything"): the latest push
    return GSH_OK;
}

// One thread, queued behind a push's copies and conversion on the stream that carried them: the ring is complete up to `next`.  A kernel that is
// already resident (trk_loop_kernel in live mode) polls word 0 with agent-scope loads; the dispatch of THIS kernel is what orders the DMA's writes
// before the store, exactly as it would for any kernel launched behind a copy.
__global__ void publish_live_kernel(unsigned long long* live, unsigned long long next, unsigned long long origin)
{
    __hip_atomic_store(live + 1, origin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(live, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int stream_publish_live(gsh_stream* s, unsigned long long next, hipStream_t st)
{
    if (s->d_live == nullptr) return GSH_OK;
    hipLaunchKernelGGL(publish_live_kernel, dim3(1), dim3(1), 0, st, s->d_live, next, s->origin);
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}

unsigned long long* stream_live_words(gsh_stream* s)
{
    if (s->d_live != nullptr) return s->d_live;
    if (hipSetDevice(s->device) != hipSuccess) return nullptr;
    unsigned long long* w = nullptr;
    hipError_t e = hipMalloc(&w, 2 * sizeof(unsigned long long));
    if (e != hipSuccess)
        {
            hip_fail(e, "hipMalloc(live words)", __FILE__, __LINE__);
            return nullptr;
        }
    s->d_live = w;
    // what is resident now, behind everything queued so far
    if (stream_publish_live(s, s->next, s->stream) != GSH_OK || hipStreamSynchronize(s->stream) != hipSuccess)
        {
            (void)hipFree(w);
            s->d_live = nullptr;
            set_error(GSH_ERR_HIP, "the ring's live words could not be published");
            return nullptr;
        }
    return s->d_live;
}

unsigned long long stream_live_floor(gsh_stream* s)
{
    unsigned long long lowest = ~0ull;
    for (auto it = s->live_floors.begin(); it != s->live_floors.end();)
        {
            LiveFloor& f = **it;
            std::lock_guard<std::mutex> lk(f.m);
            if (f.tails == nullptr)
                {
                    it = s->live_floors.erase(it);  // the handle is gone
                    continue;
                }
            for (int c = 0; c < f.n; c++)
                if (f.tails[c].active && f.tails[c].pos < lowest) lowest = f.tails[c].pos;
            ++it;
        }
    return lowest;
}
}  // namespace gsh

namespace
{
using gsh::set_error;

// GSH_STREAM_DIRECT_DMA=0: page-locked gr_complex items go through the device staging buffer like every other item type (the path of rounds 2-3, for A/B runs)
bool stream_direct_dma()
{
    static const bool on = [] {
        const char* e = std::getenv("GSH_STREAM_DIRECT_DMA");
        return e == nullptr || std::atoi(e) != 0;
    }();
    return on;
}

// a staging buffer that has been outgrown.  hipFree waits for the WHOLE device -- with a live loop resident on it (tracking_loop.hip) that is until the
// loop's residency ends, milliseconds during which the loop may be waiting for the very push that wants to free: while live readers are registered the
// buffer is parked instead and released with the ring.
hipError_t release_buffer(gsh_stream* s, void* p, bool host)
{
    if (!s->live_floors.empty())
        {
            (host ? s->parked_host : s->parked_device).push_back(p);
            return hipSuccess;
        }
    return host ? hipHostFree(p) : hipFree(p);
}

// queue the conversion of n items at d_src into ring positions of absolute indices [first, first + n) on `st`.  host_src: d_src is page-locked HOST memory
// holding gr_complex items to be taken as they are -- the ring positions are then the destination of the DMA itself (no staging buffer, no second copy);
// *copied_after (an event), when given, is recorded behind the last read of d_src.
int write_items(gsh_stream* s, const void* d_src, unsigned long long n, int item_type, int conj, hipStream_t st, bool host_src = false, hipEvent_t copied_after = nullptr)
{
    const size_t isz = gsh::item_bytes(item_type);
    if (!s->live_floors.empty())
        {
            // a resident loop cannot be waited for by an event (it would wait for this very push): what its channels still read is off limits instead
            const unsigned long long floor = gsh::stream_live_floor(s);
            if (floor != ~0ull && s->next + n > floor + s->capacity)
                return set_error(GSH_ERR_STATE, "a push of %llu samples at %llu would overwrite sample %llu, which a live tracking channel has not correlated yet (ring capacity %llu)",
                    n, s->next, floor, s->capacity);
        }
    {
        // everything below `bound` is overwritten by this push: wait for the launches that still read below it
        const unsigned long long end = s->next + n;
        const unsigned long long bound = end > s->capacity ? end - s->capacity : 0ull;
        const int m = s->read_count < gsh_stream::HIST ? s->read_count : gsh_stream::HIST;
        if (s->has_fold && st != s->stream) GSH_HIP(hipStreamWaitEvent(st, s->read_fold, 0));  // launches older than the history (on the ring's own stream: already ordered)
        for (int k = 1; k <= m; k++)
            {
                const int slot = (s->read_count - k) % gsh_stream::HIST;
                if (s->read_min[slot] < bound) GSH_HIP(hipStreamWaitEvent(st, s->read_ev[slot], 0));
            }
    }
    const unsigned long long C = s->capacity, M = s->max_window;
    unsigned long long done = 0;
    while (done < n)
        {
            const unsigned long long p = (s->next + done) % C;
            const unsigned long long len = std::min(n - done, C - p);
            const char* src = static_cast<const char*>(d_src) + done * isz;
            if (host_src)
                {
                    GSH_HIP(hipMemcpyAsync(s->d_ring + p, src, sizeof(float2) * len, hipMemcpyHostToDevice, st));
                    if (done + len == n && copied_after != nullptr) GSH_HIP(hipEventRecord(copied_after, st));  // (before the mirror copy: that one reads the ring)
                }
            else
                {
                    int rc = gsh::convert_to_complex(src, item_type, conj, s->d_ring + p, len, st);
                    if (rc != GSH_OK) return rc;
                }
            if (p < M)  // keep the mirror behind the end in step
                {
                    const unsigned long long ml = std::min(len, M - p);
                    GSH_HIP(hipMemcpyAsync(s->d_ring + C + p, s->d_ring + p, sizeof(float2) * ml, hipMemcpyDeviceToDevice, st));
                }
            done += len;
        }
    return GSH_OK;
}
}  // namespace

namespace
{
// after a push's device work has been queued on `st`: the "latest" event and the (end index, event) history entry
int record_push(gsh_stream* s, unsigned long long end_index, hipStream_t st)
{
    {
        // ONE event per push (every driver call here is ~6 us on the thread that appends, and the blocks of a stream wait for it): the history entry; `pushed`
        // -- "the latest push" -- is whichever entry was recorded last
        std::lock_guard<std::mutex> lk(s->hist_mutex);
        const int slot = s->push_count % gsh_stream::HIST;
        if (s->push_ev[slot] == nullptr) GSH_HIP(hipEventCreateWithFlags(&s->push_ev[slot], hipEventDisableTiming));
        GSH_HIP(hipEventRecord(s->push_ev[slot], st));
        s->push_end[slot] = end_index;
        s->push_count++;
        s->pushed = s->push_ev[slot];
    }
    return gsh::stream_publish_live(s, end_index, st);
}
}  // namespace

namespace gsh
{
int stream_write_device_items(gsh_stream* s, const void* d_src, unsigned long long n, int item_type, int conj, hipStream_t st)
{
    int rc = write_items(s, d_src, n, item_type, conj, st);
    if (rc != GSH_OK) return rc;
    rc = record_push(s, s->next + n, st);
    if (rc != GSH_OK) return rc;
    s->next += n;
    return GSH_OK;
}
}  // namespace gsh

extern "C"
{
    int gsh_stream_create(int device, uint64_t capacity_samples, uint32_t max_window_samples, gsh_stream_t** out)
    {
        GSH_REQUIRE(out != nullptr, "null out pointer");
        *out = nullptr;
        GSH_REQUIRE(max_window_samples >= 1, "max_window_samples must be positive");
        GSH_REQUIRE(capacity_samples >= 2ull * max_window_samples, "capacity_samples %llu must be at least twice max_window_samples %u",
            static_cast<unsigned long long>(capacity_samples), max_window_samples);
        GSH_REQUIRE(capacity_samples <= (1ull << 34), "capacity_samples %llu too large", static_cast<unsigned long long>(capacity_samples));
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        gsh_stream* s = new (std::nothrow) gsh_stream();
        GSH_REQUIRE(s != nullptr, "out of host memory");
        s->device = device;
        s->capacity = capacity_samples + (capacity_samples & 1ull);  // even: windows keep their 16-byte phase across the wrap
        s->max_window = max_window_samples;
        auto fail = [&](hipError_t e, const char* what) {
            gsh::hip_fail(e, what, __FILE__, __LINE__);
            gsh_stream_destroy(s);
            return GSH_ERR_HIP;
        };
        hipError_t e;
        {
            // The ring's stream gets the highest priority the device offers: the runtime keeps separate hardware queues per priority level, so a push can
            // never end up in a queue BEHIND a resident live loop (normal / low priority) that is waiting for that very push (tracking_loop.hip, live mode).
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
            int prio = greatest;
            if (const char* pe = std::getenv("GSH_STREAM_PRIORITY")) prio = std::min(std::max(std::atoi(pe), greatest), least);  // (A/B runs)
            if ((e = hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, prio)) != hipSuccess) return fail(e, "hipStreamCreate");
        }
        const size_t total = static_cast<size_t>(s->capacity + s->max_window + 2);
        if ((e = hipMalloc(&s->d_ring, sizeof(float2) * total)) != hipSuccess) return fail(e, "hipMalloc(ring)");
        if ((e = hipMemset(s->d_ring, 0, sizeof(float2) * total)) != hipSuccess) return fail(e, "hipMemset(ring)");
        *out = s;
        return GSH_OK;
    }

    void gsh_stream_destroy(gsh_stream_t* s)
    {
        if (!s) return;
        (void)hipSetDevice(s->device);
        // live tracking handles that still follow this ring: their resident kernels read d_ring and the live words -- they are told to leave and to forget the ring
        // BEFORE the memory goes (their next gsh_trk_live_begin / run then fails with GSH_ERR_STATE: the block gives its channel up the reference's way)
        {
            std::vector<std::shared_ptr<gsh::LiveFloor>> floors;
            floors.swap(s->live_floors);
            for (auto& f : floors)
                {
                    void (*gone)(void*) = nullptr;
                    void* owner = nullptr;
                    {
                        std::lock_guard<std::mutex> lk(f->m);
                        if (f->tails != nullptr)
                            {
                                gone = f->ring_gone;
                                owner = f->owner;
                            }
                    }
                    if (gone != nullptr) gone(owner);
                }
        }
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        for (void* p : s->parked_device) (void)hipFree(p);
        for (void* p : s->parked_host) (void)hipHostFree(p);
        if (s->d_ring) (void)hipFree(s->d_ring);
        if (s->d_live) (void)hipFree(s->d_live);
        if (s->d_raw) (void)hipFree(s->d_raw);
        for (int i = 0; i < 2; i++)
            {
                if (s->d_raw2[i]) (void)hipFree(s->d_raw2[i]);
                if (s->raw2_done[i]) (void)hipEventDestroy(s->raw2_done[i]);
            }
        if (s->read_fold) (void)hipEventDestroy(s->read_fold);
        for (int i = 0; i < gsh_stream::NSTAGE; i++)
            {
                if (s->stage_done[i]) (void)hipEventDestroy(s->stage_done[i]);
                if (s->h_stage[i]) (void)hipHostFree(s->h_stage[i]);
                if (s->d_stage[i]) (void)hipFree(s->d_stage[i]);
            }
        for (int i = 0; i < gsh_stream::HIST; i++)
            {
                if (s->push_ev[i]) (void)hipEventDestroy(s->push_ev[i]);
                if (s->read_ev[i]) (void)hipEventDestroy(s->read_ev[i]);
            }
        if (s->stream) (void)hipStreamDestroy(s->stream);
        delete s;
    }

    int gsh_stream_push_device(gsh_stream_t* s, const void* device_items, uint64_t n, int item_type, int inverted_spectrum, void* hip_stream,
        uint64_t* first_index)
    {
        GSH_REQUIRE(s != nullptr, "null stream");
        GSH_REQUIRE(n == 0 || device_items != nullptr, "null items");
        GSH_REQUIRE(gsh::item_bytes(item_type) != 0, "unknown item type %d", item_type);
        GSH_REQUIRE(n <= s->capacity, "a push of %llu samples exceeds the ring capacity %llu", static_cast<unsigned long long>(n), s->capacity);
        if (first_index) *first_index = s->next;
        if (n == 0) return GSH_OK;
        GSH_HIP(hipSetDevice(s->device));
        hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->stream;
        int rc = write_items(s, device_items, n, item_type, inverted_spectrum ? 1 : 0, st);
        if (rc != GSH_OK) return rc;
        rc = record_push(s, s->next + n, st);
        if (rc != GSH_OK) return rc;
        s->next += n;
        if (!hip_stream) GSH_HIP(hipStreamSynchronize(st));
        return GSH_OK;
    }

    int gsh_stream_push(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index)
    {
        GSH_REQUIRE(s != nullptr, "null stream");
        GSH_REQUIRE(n == 0 || items != nullptr, "null items");
        const size_t isz = gsh::item_bytes(item_type);
        GSH_REQUIRE(isz != 0, "unknown item type %d", item_type);
        GSH_REQUIRE(n <= s->capacity, "a push of %llu samples exceeds the ring capacity %llu", static_cast<unsigned long long>(n), s->capacity);
        if (first_index) *first_index = s->next;
        if (n == 0) return GSH_OK;
        GSH_HIP(hipSetDevice(s->device));
        const size_t bytes = static_cast<size_t>(n) * isz;
        if (bytes > s->raw_cap)
            {
                if (s->d_raw) GSH_HIP(release_buffer(s, s->d_raw, false));
                s->d_raw = nullptr;
                s->raw_cap = 0;
                GSH_HIP(hipMalloc(&s->d_raw, bytes));
                s->raw_cap = bytes;
            }
        GSH_HIP(hipMemcpyAsync(s->d_raw, items, bytes, hipMemcpyHostToDevice, s->stream));
        int rc = write_items(s, s->d_raw, n, item_type, inverted_spectrum ? 1 : 0, s->stream);
        if (rc != GSH_OK) return rc;
        rc = record_push(s, s->next + n, s->stream);
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipStreamSynchronize(s->stream));
        s->next += n;
        return GSH_OK;
    }

    int gsh_stream_push_async(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index)
    {
        // H2D copy + conversion queued on the ring's stream; returns without waiting.  Two device staging buffers alternate, so the copy of
        // block k + 1 can run while the conversion of block k still reads its staging buffer, and -- the point of it -- while the
        // correlators work on block k on their own streams (they wait on `pushed`, not on the host).
        GSH_REQUIRE(s != nullptr, "null stream");
        GSH_REQUIRE(n == 0 || items != nullptr, "null items");
        const size_t isz = gsh::item_bytes(item_type);
        GSH_REQUIRE(isz != 0, "unknown item type %d", item_type);
        GSH_REQUIRE(n <= s->capacity, "a push of %llu samples exceeds the ring capacity %llu", static_cast<unsigned long long>(n), s->capacity);
        if (first_index) *first_index = s->next;
        if (n == 0) return GSH_OK;
        GSH_HIP(hipSetDevice(s->device));
        const int slot = s->raw2_next;
        s->raw2_next ^= 1;
        const size_t bytes = static_cast<size_t>(n) * isz;
        if (s->raw2_done[slot] == nullptr) GSH_HIP(hipEventCreateWithFlags(&s->raw2_done[slot], hipEventDisableTiming));
        if (bytes > s->raw2_cap[slot])
            {
                GSH_HIP(hipEventSynchronize(s->raw2_done[slot]));  // nothing queued may still read the buffer being replaced
                if (s->d_raw2[slot]) GSH_HIP(release_buffer(s, s->d_raw2[slot], false));
                s->d_raw2[slot] = nullptr;
                s->raw2_cap[slot] = 0;
                GSH_HIP(hipMalloc(&s->d_raw2[slot], bytes));
                s->raw2_cap[slot] = bytes;
            }
        // same stream as the previous conversion out of this slot (two pushes ago): ordered without an explicit wait
        GSH_HIP(hipMemcpyAsync(s->d_raw2[slot], items, bytes, hipMemcpyHostToDevice, s->stream));
        int rc = write_items(s, s->d_raw2[slot], n, item_type, inverted_spectrum ? 1 : 0, s->stream);
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipEventRecord(s->raw2_done[slot], s->stream));
        rc = record_push(s, s->next + n, s->stream);
        if (rc != GSH_OK) return rc;
        s->next += n;
        return GSH_OK;
    }

    int gsh_stream_push_staged(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index)
    {
        // items -> page-locked staging (host memcpy: the caller's buffer is free on return, whatever kind of memory it is) -> the ring.  gr_complex items that
        // need no conversion are copied by the DMA engine straight from the staging buffer into their ring positions; every other item type goes
        // staging -> device staging -> conversion kernel.  A staging pair is re-used four pushes later, after the device work that read it has finished.
        GSH_REQUIRE(s != nullptr, "null stream");
        GSH_REQUIRE(n == 0 || items != nullptr, "null items");
        const size_t isz = gsh::item_bytes(item_type);
        GSH_REQUIRE(isz != 0, "unknown item type %d", item_type);
        GSH_REQUIRE(n <= s->capacity, "a push of %llu samples exceeds the ring capacity %llu", static_cast<unsigned long long>(n), s->capacity);
        if (first_index) *first_index = s->next;
        if (n == 0) return GSH_OK;
        GSH_HIP(hipSetDevice(s->device));
        const int slot = s->stage_next;
        s->stage_next = (s->stage_next + 1) % gsh_stream::NSTAGE;
        const size_t bytes = static_cast<size_t>(n) * isz;
        if (s->stage_done[slot] == nullptr)
            GSH_HIP(hipEventCreateWithFlags(&s->stage_done[slot], hipEventDisableTiming));
        else
            GSH_HIP(hipEventSynchronize(s->stage_done[slot]));  // the push that used this pair four pushes ago
        if (bytes > s->stage_cap[slot])
            {
                if (s->h_stage[slot]) GSH_HIP(release_buffer(s, s->h_stage[slot], true));
                if (s->d_stage[slot]) GSH_HIP(release_buffer(s, s->d_stage[slot], false));
                s->h_stage[slot] = nullptr;
                s->d_stage[slot] = nullptr;
                s->stage_cap[slot] = 0;
                const size_t cap = bytes + bytes / 2;
                GSH_HIP(hipHostMalloc(&s->h_stage[slot], cap, hipHostMallocDefault));
                GSH_HIP(hipMalloc(&s->d_stage[slot], cap));
                s->stage_cap[slot] = cap;
            }
        std::memcpy(s->h_stage[slot], items, bytes);
        int rc;
        if (item_type == GSH_ITEM_GR_COMPLEX && !inverted_spectrum && stream_direct_dma())
            {
                // the ring's own format: the DMA's destination is the ring (profiles/ab/r03/dropin_direct_dma.txt); its SOURCE is the page-locked staging
                // copy, never the caller's pageable buffer (an asynchronous copy out of pageable memory is only safe if the runtime happens to stage it itself)
                rc = write_items(s, s->h_stage[slot], n, item_type, 0, s->stream, true, nullptr);
            }
        else
            {
                GSH_HIP(hipMemcpyAsync(s->d_stage[slot], s->h_stage[slot], bytes, hipMemcpyHostToDevice, s->stream));
                rc = write_items(s, s->d_stage[slot], n, item_type, inverted_spectrum ? 1 : 0, s->stream);
            }
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipEventRecord(s->stage_done[slot], s->stream));
        rc = record_push(s, s->next + n, s->stream);
        if (rc != GSH_OK) return rc;
        s->next += n;
        return GSH_OK;
    }

    int gsh_stream_wait_copied(gsh_stream_t* s)
    {
        GSH_REQUIRE(s != nullptr, "null stream");
        hipEvent_t ev;
        {
            std::lock_guard<std::mutex> lk(s->hist_mutex);
            ev = s->pushed;
        }
        if (ev == nullptr) return GSH_OK;    // nothing has been queued yet
        GSH_HIP(hipEventSynchronize(ev));    // the latest push has reached the ring: every DMA out of the callers' memory lies before that (no hipSetDevice: an event knows its device)
        return GSH_OK;
    }

    int gsh_stream_wait_copied_upto(gsh_stream_t* s, uint64_t end_index, uint64_t* complete_upto)
    {
        GSH_REQUIRE(s != nullptr, "null stream");
        hipEvent_t ev = nullptr;
        unsigned long long covered = 0;
        {
            // the oldest recorded push that reaches end_index (ends grow with the push count); pushes that have dropped out of the history were queued
            // before every one in it on streams that complete in order for this purpose: the oldest entry stands for them
            std::lock_guard<std::mutex> lk(s->hist_mutex);
            const int n = s->push_count < gsh_stream::HIST ? s->push_count : gsh_stream::HIST;
            for (int k = n; k >= 1; k--)
                {
                    const int slot = (s->push_count - k) % gsh_stream::HIST;
                    ev = s->push_ev[slot];
                    covered = s->push_end[slot];
                    if (s->push_end[slot] >= end_index) break;
                }
        }
        if (ev != nullptr) GSH_HIP(hipEventSynchronize(ev));  // (an event that has been re-recorded for a later push meanwhile only makes the wait longer)
        if (complete_upto != nullptr) *complete_upto = covered;
        return GSH_OK;
    }

    int gsh_stream_push_pinned(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index)
    {
        int rc = gsh_stream_push_pinned_async(s, items, n, item_type, inverted_spectrum, first_index);
        if (rc != GSH_OK || n == 0) return rc;
        return gsh_stream_wait_copied(s);  // `items` is free again; the conversion and the readers' waits stay asynchronous
    }

    int gsh_stream_push_pinned_async(gsh_stream_t* s, const void* items, uint64_t n, int item_type, int inverted_spectrum, uint64_t* first_index)
    {
        // page-locked items -> device staging (DMA straight out of the caller's memory) -> conversion into the ring; nothing waits here
        GSH_REQUIRE(s != nullptr, "null stream");
        GSH_REQUIRE(n == 0 || items != nullptr, "null items");
        const size_t isz = gsh::item_bytes(item_type);
        GSH_REQUIRE(isz != 0, "unknown item type %d", item_type);
        GSH_REQUIRE(n <= s->capacity, "a push of %llu samples exceeds the ring capacity %llu", static_cast<unsigned long long>(n), s->capacity);
        if (first_index) *first_index = s->next;
        if (n == 0) return GSH_OK;
        GSH_HIP(hipSetDevice(s->device));
        if (item_type == GSH_ITEM_GR_COMPLEX && !inverted_spectrum && stream_direct_dma())
            {
                // items that are the ring's own format: the DMA's destination is the ring (profiles/ab/r03/dropin_direct_dma.txt)
                int rc = write_items(s, items, n, item_type, 0, s->stream, true, nullptr);
                if (rc != GSH_OK) return rc;
                rc = record_push(s, s->next + n, s->stream);
                if (rc != GSH_OK) return rc;
                s->next += n;
                return GSH_OK;
            }
        const int slot = s->stage_next;
        s->stage_next = (s->stage_next + 1) % gsh_stream::NSTAGE;
        const size_t bytes = static_cast<size_t>(n) * isz;
        if (s->stage_done[slot] == nullptr)
            GSH_HIP(hipEventCreateWithFlags(&s->stage_done[slot], hipEventDisableTiming));
        else
            GSH_HIP(hipEventSynchronize(s->stage_done[slot]));  // the conversion that read this device buffer four pushes ago
        if (bytes > s->stage_cap[slot])
            {
                if (s->h_stage[slot]) GSH_HIP(release_buffer(s, s->h_stage[slot], true));
                if (s->d_stage[slot]) GSH_HIP(release_buffer(s, s->d_stage[slot], false));
                s->h_stage[slot] = nullptr;
                s->d_stage[slot] = nullptr;
                s->stage_cap[slot] = 0;
                const size_t cap = bytes + bytes / 2;
                GSH_HIP(hipHostMalloc(&s->h_stage[slot], cap, hipHostMallocDefault));  // (kept in step with the staged path, which shares the slots)
                GSH_HIP(hipMalloc(&s->d_stage[slot], cap));
                s->stage_cap[slot] = cap;
            }
        GSH_HIP(hipMemcpyAsync(s->d_stage[slot], items, bytes, hipMemcpyHostToDevice, s->stream));
        int rc = write_items(s, s->d_stage[slot], n, item_type, inverted_spectrum ? 1 : 0, s->stream);
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipEventRecord(s->stage_done[slot], s->stream));
        rc = record_push(s, s->next + n, s->stream);
        if (rc != GSH_OK) return rc;
        s->next += n;
        return GSH_OK;
    }

    int gsh_host_register(int device, void* ptr, size_t bytes)
    {
        GSH_REQUIRE(ptr != nullptr && bytes > 0, "null / empty range");
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        GSH_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
        return GSH_OK;
    }

    int gsh_host_unregister(void* ptr)
    {
        GSH_REQUIRE(ptr != nullptr, "null pointer");
        GSH_HIP(hipHostUnregister(ptr));
        return GSH_OK;
    }

    int gsh_stream_wait(gsh_stream_t* s)
    {
        GSH_REQUIRE(s != nullptr, "null stream");
        GSH_HIP(hipSetDevice(s->device));
        GSH_HIP(hipStreamSynchronize(s->stream));
        return GSH_OK;
    }

    int gsh_stream_seek(gsh_stream_t* s, uint64_t next_index)
    {
        GSH_REQUIRE(s != nullptr, "null stream");
        if (!s->live_floors.empty())
            {
                // a resident loop cannot be fenced by events: a seek under it would let the next push overwrite a window it is correlating right now.  While a live channel
                // is active the ring only moves forward (the caller quiesces the loops -- gsh_trk_live_quiesce / gsh_trk_stop -- or pushes the gap).
                const unsigned long long floor = gsh::stream_live_floor(s);
                if (floor != ~0ull)
                    return set_error(GSH_ERR_STATE, "gsh_stream_seek to %llu while a live tracking channel still reads the ring at %llu (stop or quiesce the loops first)",
                        static_cast<unsigned long long>(next_index), floor);
            }
        GSH_HIP(hipSetDevice(s->device));
        GSH_HIP(hipStreamSynchronize(s->stream));  // queued pushes (and the folded reader fences)
        {
            // launches on other streams that still read the ring: after the seek nothing protects what they read
            const int m = s->read_count < gsh_stream::HIST ? s->read_count : gsh_stream::HIST;
            for (int k = 1; k <= m; k++) GSH_HIP(hipEventSynchronize(s->read_ev[(s->read_count - k) % gsh_stream::HIST]));
        }
        s->has_fold = false;
        s->next = next_index;
        s->origin = next_index;  // nothing older is resident
        {
            std::lock_guard<std::mutex> lk(s->hist_mutex);
            s->push_count = 0;
        }
        s->read_count = 0;
        if (s->d_live != nullptr)
            {
                int rc = gsh::stream_publish_live(s, next_index, s->stream);
                if (rc != GSH_OK) return rc;
                GSH_HIP(hipStreamSynchronize(s->stream));
            }
        return GSH_OK;
    }

    int gsh_stream_range(gsh_stream_t* s, uint64_t* oldest, uint64_t* next)
    {
        GSH_REQUIRE(s != nullptr, "null stream");
        if (oldest) *oldest = gsh::stream_oldest(s);
        if (next) *next = s->next;
        return GSH_OK;
    }

    int gsh_stream_read(gsh_stream_t* s, uint64_t index, uint64_t n, float* out_iq)
    {
        GSH_REQUIRE(s != nullptr && (n == 0 || out_iq != nullptr), "null argument");
        GSH_REQUIRE(index >= gsh::stream_oldest(s) && index + n <= s->next, "samples [%llu, %llu) are not resident (ring holds [%llu, %llu))",
            static_cast<unsigned long long>(index), static_cast<unsigned long long>(index + n), gsh::stream_oldest(s), s->next);
        GSH_HIP(hipSetDevice(s->device));
        GSH_HIP(hipStreamSynchronize(s->stream));
        unsigned long long done = 0;
        while (done < n)
            {
                const unsigned long long p = (index + done) % s->capacity;
                const unsigned long long len = std::min<unsigned long long>(n - done, s->capacity - p);
                GSH_HIP(hipMemcpy(out_iq + 2 * done, s->d_ring + p, sizeof(float2) * len, hipMemcpyDeviceToHost));
                done += len;
            }
        return GSH_OK;
    }

    int gsh_convert_samples_device(int device, const void* device_items, int item_type, int inverted_spectrum, void* device_dst, uint64_t n,
        void* hip_stream)
    {
        GSH_REQUIRE(n == 0 || (device_items != nullptr && device_dst != nullptr), "null argument");
        GSH_REQUIRE(gsh::item_bytes(item_type) != 0, "unknown item type %d", item_type);
        GSH_REQUIRE((reinterpret_cast<uintptr_t>(device_dst) & 7u) == 0, "destination must be 8-byte aligned");
        int rc = gsh::use_device(device);
        if (rc != GSH_OK) return rc;
        return gsh::convert_to_complex(device_items, item_type, inverted_spectrum ? 1 : 0, static_cast<float2*>(device_dst), n,
            static_cast<hipStream_t>(hip_stream));
    }
}
