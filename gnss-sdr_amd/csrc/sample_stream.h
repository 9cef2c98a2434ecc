// Device-resident IF sample ring (C ABI gsh_stream_*, include/gnss_sdr_hip.h).  Internal view for the other
// translation units of the library.
#ifndef GSH_SAMPLE_STREAM_H
#define GSH_SAMPLE_STREAM_H
#include "gsh_internal.h"
#include <memory>
#include <mutex>
#include <vector>

namespace gsh
{
// What a live loop (gsh_trk_live_*, tracking_loop.hip) leaves in page-locked host memory per channel after every code period -- and what a push
// looks at before it overwrites the oldest samples of a ring such a loop follows.
struct LiveTail
{
    unsigned long long pos;  // first sample of the channel's next correlation window
    unsigned long long seq;  // periods completed since the handle was created (the record of period q sits in slot q % ring length)
    int active;              // the loop still advances the channel
    int exit_reason;         // why the channel's work-group left its last residency (LIVE_EXIT_*)
};
// A live handle's channels as seen by the ring they read: registered with the ring, outlives neither side (shared).
struct LiveFloor
{
    std::mutex m;
    const volatile LiveTail* tails{nullptr};  // nullptr once the owning handle is gone
    int n{0};
    // the ring is being destroyed under the handle (gsh_stream_destroy): the handle's residencies must leave and forget the ring before its memory goes.
    // Set by the handle with its registration; called by the ring WITHOUT m held (the callback unregisters, which takes m).
    void (*ring_gone)(void* owner){nullptr};
    void* owner{nullptr};
};
}  // namespace gsh

struct gsh_stream
{
    int device{0};
    hipStream_t stream{nullptr};
    unsigned long long capacity{0};    // C: samples kept
    unsigned long long max_window{0};  // M: mirror length; any window <= M is contiguous
    float2* d_ring{nullptr};           // C + M + 2 samples; absolute index i lives at i % C (and at C + i % C when i % C < M)
    void* d_raw{nullptr};              // staging for raw host items before conversion
    size_t raw_cap{0};                 // bytes
    unsigned long long next{0};        // absolute index of the next sample to be pushed
    unsigned long long origin{0};      // absolute index of the first sample ever pushed since the last seek (nothing older is resident)
    hipEvent_t pushed{nullptr};        // the history entry (push_ev) recorded last: "after the latest push's device work"; not owned
    void* d_raw2[2]{nullptr, nullptr}; // gsh_stream_push_async: two device staging buffers, used alternately
    size_t raw2_cap[2]{0, 0};
    hipEvent_t raw2_done[2]{nullptr, nullptr};  // the conversion that read staging buffer i has finished
    int raw2_next{0};
    // Ordering between the ring's writer (pushes, on the ring's stream) and its readers (banks, loops, acquisition handles, on theirs) is kept
    // by events, never by the host, and by SAMPLE RANGE so that copies and kernels really overlap:
    //   push history    every push records (end index, event); a reader that needs samples up to `need_end` waits for the OLDEST push that
    //                   covers it -- not for whatever was queued last;
    //   reader history  every launch that reads the ring records (lowest index it reads, event); a push that overwrites everything below
    //                   `next + n - capacity` waits only for the readers that still reach below that bound.
    static constexpr int HIST = 16;
    unsigned long long push_end[HIST]{};
    hipEvent_t push_ev[HIST]{};
    int push_count{0};                 // pushes recorded so far (slot = count % HIST)
    unsigned long long read_min[HIST]{};
    hipEvent_t read_ev[HIST]{};
    int read_count{0};
    // reader launches that have dropped out of the 16-entry history are not forgotten: before a slot is re-used the ring's own stream is made
    // to wait for the launch it held, and `read_fold` (recorded on the ring's stream right after) stands for all of them -- every push waits on it
    hipEvent_t read_fold{nullptr};
    bool has_fold{false};
    // gsh_stream_push_staged: page-locked host staging + device staging, four in rotation
    static constexpr int NSTAGE = 4;
    void* h_stage[NSTAGE]{};
    void* d_stage[NSTAGE]{};
    size_t stage_cap[NSTAGE]{};
    hipEvent_t stage_done[NSTAGE]{};  // the conversion that read d_stage[i] (and therefore the copy out of h_stage[i]) has finished
    int stage_next{0};
    // Live readers (gsh_trk_live_*): a kernel that stays resident cannot be ordered against pushes by events -- it learns how far the ring is
    // COMPLETE from two words in device memory that a one-thread kernel, queued on the pushing stream behind every push's copies and conversion,
    // rewrites: d_live[0] = absolute index one past the newest complete sample, d_live[1] = first index resident since the last seek.
    unsigned long long* d_live{nullptr};
    std::vector<std::shared_ptr<gsh::LiveFloor>> live_floors;  // what the live loops still read: a push never overwrites it (write_items)
    std::vector<void*> parked_device, parked_host;  // outgrown staging buffers kept until the ring goes (release_buffer, sample_stream.hip)
    std::mutex hist_mutex;  // the push history above is also read by gsh_stream_wait_copied_upto, from any thread
};

namespace gsh
{
// device address of absolute sample `index`, contiguous for n samples; GSH_ERR_INVALID when [index, index+n) is not resident
int stream_window(const gsh_stream* s, unsigned long long index, unsigned long long n, const float2** ptr);
inline unsigned long long stream_oldest(const gsh_stream* s)
{
    const unsigned long long by_capacity = s->next > s->capacity ? s->next - s->capacity : 0ull;
    return by_capacity > s->origin ? by_capacity : s->origin;
}
// a reader has queued work on `st` that reads ring samples at or above min_index (0: anything): pushes that overwrite below min_index + ... wait for it
int stream_mark_read(gsh_stream* s, unsigned long long min_index, hipStream_t st);
// make `st` wait until samples below `need_end` are in the ring (need_end = ~0ull: everything pushed so far)
int stream_wait_pushed(gsh_stream* s, unsigned long long need_end, hipStream_t st);
// queue the conversion of n raw items at d_src (device memory) into ring positions [next, next + n) on the ring's own stream, after the readers'
// fences; records `pushed` and advances `next`
int stream_write_device_items(gsh_stream* s, const void* d_src, unsigned long long n, int item_type, int conj, hipStream_t st);
// the two live words of the ring (allocated, and published for what is resident now, at the first call); nullptr + last error on failure
unsigned long long* stream_live_words(gsh_stream* s);
// lowest window start any registered live channel still has to read; ~0ull when none is active
unsigned long long stream_live_floor(gsh_stream* s);
}  // namespace gsh
#endif
