/*!
 * \file hip_tracking_runtime.h
 * \brief Channel-batching runtime behind the tracking blocks: ONE device launch advances every channel of a stream that has work.
 *
 * In gnss-sdr all channels run concurrently off one input buffer (src/core/receiver/gnss_flowgraph.cc:1227-1231), each
 * dll_pll_veml_tracking block doing one code period per general_work call on its own scheduler thread
 * (src/algorithms/tracking/gnuradio_blocks/dll_pll_veml_tracking.cc:1898-2001, "trk.cc").  The engine's device-closed loop (gsh_trk_*)
 * serves N channels per launch; a block that owned a one-channel handle would pay one launch and one synchronisation per channel and
 * period.  This runtime keeps the reference's threading model -- every block still calls push / take from its own thread -- and makes the
 * launch a shared one:
 *
 *   - one gsh_trk handle per loop configuration ("group": the channels of one signal class with identical parameters), a slot per
 *     tracking block;
 *   - the stream lives once in a device ring (Hip_Sample_Ring): push() de-duplicates by absolute sample index under one lock
 *     (Hip_Sample_Ring::push_from), so 32 channels of a stream upload it once, through page-locked staging, while the kernel runs;
 *   - take(): a block that finds no finished record for itself and no launch in flight for its group becomes the launcher: it queues ONE
 *     gsh_trk_run over the group -- every started channel advances through all the periods whose samples are resident (up to
 *     periods_per_launch) -- waits for it and files the records into per-slot queues; blocks that arrive meanwhile wait on a condition
 *     variable and usually find their records filed when they wake (leader / followers; no extra thread, and a single-threaded caller
 *     works unchanged).  The device therefore runs ahead of the slowest block by at most what the fastest one has pushed; a block takes
 *     only records whose samples it has been offered itself (consume_each may not pass its own input);
 *   - launch-ahead: the launcher, having filed a launch, queues the next one at once when at least half as many periods are already resident again,
 *     and returns without waiting for it; the device then works while the blocks consume what was filed, and the block that next runs dry ends and files
 *     the launch (GSH_TRK_LAUNCH_AHEAD=0 in the environment turns this off: one launch, waited for, at a time);
 *   - LIVE mode (the default; GSH_TRK_LIVE=0 or live = false restores the launches above): the group's loop kernel stays resident
 *     (gsh_trk_live_begin), learns of new samples from the ring itself and leaves every period's record in a ring of records in page-locked host
 *     memory; take() then is a read of that memory (gsh_trk_live_take: no lock of the runtime, no device call, no wake-up of anybody), and a
 *     block that finds its next window resident but no record yet makes sure a residency is in flight and waits for the record (microseconds).
 *     At the reference's cadence -- one code period per general_work call, trk.cc:1898-2001 -- this is what removes the ~270 us of host time
 *     around every launch.  start / stop quiesce the group's residencies first (the device side of both needs the loop state at rest);
 *   - start / stop of one channel serialise with the launches of its group and nothing else (a launch queued ahead is ended and filed first).
 *
 * Plain C++17 over the C ABI (include/gnss_sdr_hip.h); no HIP headers, no GNU Radio.  No CPU fallback.
 */
#ifndef GNSS_SDR_HIP_TRACKING_RUNTIME_H
#define GNSS_SDR_HIP_TRACKING_RUNTIME_H

#include "gnss_sdr_hip.h"
#include "hip_correlator_runtime.h"  // Hip_Sample_Ring
#include <atomic>
#include <chrono>
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

class Hip_Tracking_Runtime
{
public:
    struct Stats
    {
        uint64_t launches{0};         //!< gsh_trk_run launches
        uint64_t channel_periods{0};  //!< records filed (one per channel and code period)
        uint64_t channels_served{0};  //!< sum over launches of the channels that completed at least one period in it
        uint32_t largest_launch{0};   //!< most channel-periods filed by one launch
        uint64_t launch_ns{0};        //!< wall time inside launches (queueing + kernel + results), summed
        uint64_t begin_ns{0};         //!< ... of which queueing the launch (gsh_trk_run_begin, ring lock included)
        uint64_t ring_wait_ns{0};     //!< ... of which waiting for the ring (a push in progress)
        uint64_t file_ns{0};          //!< wall time filing the records of finished launches into the blocks' queues, summed
        uint64_t wake_ns{0};          //!< from a launch's records being filed to a block that waited for them having taken its own, summed over those blocks
        uint64_t wakes{0};            //!< ... how many such takes
        uint64_t push_ns{0};          //!< wall time the front-runner blocks spent appending samples (staging copy + queueing), summed
        uint64_t pushed_samples{0};   //!< samples appended (every sample of the stream once, however many channels read it)
        uint64_t residencies{0};      //!< live mode: residencies of the loop kernel queued (counted in `launches` as well)
        uint64_t record_wait_ns{0};   //!< live mode: wall time blocks spent waiting for a record whose window was resident, summed
        uint64_t record_waits{0};     //!< ... how many such waits
    };

    /*! ring: the device sample ring the channels read (shared by every block that holds this runtime).
        periods_per_launch: most code periods per channel one launch runs (the launch stops earlier at the newest resident sample).
        channels_per_group: slots of one gsh_trk handle (a further group is opened when they are used up). */
    Hip_Tracking_Runtime(int device, std::shared_ptr<Hip_Sample_Ring> ring, int periods_per_launch = 16, int channels_per_group = 64, bool live = true);
    bool live() const { return d_live; }
    /*! live mode watchdog: a channel whose next window has been resident for this long without the device delivering its record is given up with an error (the block
        then publishes "events" 3 and the channel goes back to acquisition) -- a residency that never reports must not leave a block calling take() for ever.
        Default 1000 ms (the device needs ~10 us per period); <role>.hip_record_timeout_ms. */
    void set_record_timeout_ms(int ms) { d_record_timeout_ns.store(static_cast<int64_t>(std::max(ms, 1)) * 1000000); }
    /*! launched mode only (live residencies keep one work-group per channel): work-groups that share every window of a channel, gsh_trk_set_split -- 1 (default)
        one, 2 .. 8 that many, 0 the engine's own choice by window length and free compute units.  Worth it for long windows on a device with room (Galileo E1's
        128 000-sample periods: 36.5 -> 16 us per period with four); the sums are those of another order of summation, equal to rounding.  A handle on which the
        engine refuses the request (high dynamics, more work-groups than the device holds at once) runs with one.  <role>.hip_work_groups_per_channel;
        call before the first block attaches. */
    void set_work_groups_per_channel(int n) { d_work_groups_per_channel = std::min(std::max(n, 0), 8); }
    int work_groups_per_channel() const { return d_work_groups_per_channel; }
    ~Hip_Tracking_Runtime();
    Hip_Tracking_Runtime(const Hip_Tracking_Runtime&) = delete;
    Hip_Tracking_Runtime& operator=(const Hip_Tracking_Runtime&) = delete;

    bool ok() const { return d_ring && d_ring->ok(); }
    int device() const { return d_device; }
    Hip_Sample_Ring* ring() const { return d_ring.get(); }

    /*! a tracking block joins with its loop configuration; returns its slot (>= 0), -1 on failure (last_error(-1)) */
    int attach(const gsh_trk_conf& conf, int max_code_length);
    void detach(int slot);
    /*! start_tracking + pull-in (trk.cc:793-866, 1949-1978) for the slot's channel: see gsh_trk_pull_in / gsh_trk_start_ex.  On return
        *samples_offset is what the block must consume to align the stream with the replica and *first_prn_length the nominal length of the
        first period (d_current_prn_length_samples after the pull-in). */
    bool start(int slot, const float* code, const float* data_code, int code_length, uint64_t nitems_read, double acq_delay_samples,
        double acq_doppler_hz, uint64_t acq_samplestamp_samples, int32_t* samples_offset, int32_t* first_prn_length);
    /*! stop_tracking (trk.cc:1113-1116): the channel no longer advances, its queued records are dropped */
    void stop(int slot);
    bool tracking(int slot) const;
    /*! true while any slot of this runtime is tracking */
    bool any_tracking() const;
    /*! samples [first_index, first_index + n) of the stream as the calling block sees them; see Hip_Sample_Ring::push_from.  Every block of the
        stream calls it in EVERY general_work, whatever its state: the ring then always holds the stream up to the front-runner's read pointer
        (the front-runner uploads, for everybody else the call is a comparison of two indices), so a channel that starts -- or starts again --
        anywhere within the ring's capacity behind the front-runner finds its samples resident.  need_resident: the caller is tracking and
        will wait (bounded) for slower siblings to close a gap in front of its samples; a block in standby passes false. */
    bool push(const std::complex<float>* samples, uint64_t first_index, uint64_t n, bool need_resident = true);
    /*! Before the calling block hands samples below `upto` back to the scheduler (consume_each): those it pushed by DMA out of the scheduler's buffer
        have left it.  push() itself does not wait for its copies any more -- what a block is offered beyond what it consumes stays valid, and the copy
        of the look-ahead runs while the blocks work through the periods in front of it. */
    bool release_input(uint64_t upto) { return d_ring->wait_copied_upto(upto); }
    /*! up to max_records finished periods of the slot's channel, oldest first, whose samples lie below limit_end (sample_counter +
        max(vector_length, prn_length_samples) <= limit_end).  Launches the group's loop when the channel has resident work and nothing is
        in flight; waits while a launch that may produce the slot's records is running.  Returns the number of records (0: the next period's
        samples are not there yet), -1 on an engine error (last_error(slot)).  A record with flags bit 1 is a loss of lock: the channel
        has stopped and the record is the last one returned. */
    int take(int slot, uint64_t limit_end, int max_records, gsh_trk_epoch* out);
    /*! absolute index of the first sample of the slot's next window (host view) */
    uint64_t next_window(int slot) const;
    std::string last_error(int slot) const;
    Stats stats() const;

private:
    struct Group
    {
        gsh_trk_conf conf{};
        int max_code_length{0};
        gsh_trk_t* trk{nullptr};
        std::vector<int> slot_of_channel;  // -1: free
        std::mutex handle_mutex;           // the C handle: one thread at a time (launch, start, stop)
        bool in_flight{false};             // guarded by d_mutex: a thread is queueing, ending or filing a launch of the group
        std::condition_variable filed;     // (with d_mutex) a launch of THIS group has been filed / the group is free again: only its own blocks wake up
        // A launch that has been queued and not yet waited for (launch-ahead): guarded by handle_mutex.  Whoever next needs the handle -- the block that finds
        // its queue empty, start(), stop() -- ends and files it first.
        bool begun{false};
        int begun_epochs{0};
        std::vector<uint64_t> begun_generation;  // Slot::generation per channel when the launch was queued
        std::vector<gsh_trk_epoch> records;
        std::vector<int32_t> done;
        // live mode
        bool live_failed{false};                  // gsh_trk_live_begin failed once: the group stays with launches (guarded by handle_mutex; read racily as a hint)
        std::atomic<int64_t> live_checked_ns{0};  // when a block last made sure a residency was in flight (steady_clock)
        std::atomic<int64_t> held_off_ns{0};      // when start / stop of one of the group's channels last held the handle (the residencies are quiesced meanwhile): the record
                                                  // watchdog does not count that as the device's silence
    };
    struct Slot
    {
        Group* group{nullptr};
        int channel{-1};
        bool used{false};
        bool tracking{false};
        bool device_active{false};   // the device may still be advancing the channel (started, and neither stopped by the host nor by a loss-of-lock record): what stop()
                                     // has to undo even when `tracking` is already false -- a take that failed, a channel given up by the watchdog
        uint64_t generation{0};      // bumped by start / stop: records of a launch begun before are not filed
        uint64_t next_window{0};
        std::deque<gsh_trk_epoch> queue;
        std::string error;
        // live mode: the block's own thread reads records without d_mutex; start / stop (other threads) exclude it through take_mutex.  The two
        // atomics mirror `tracking` / `next_window` for push(), which scans them without a lock.
        std::mutex take_mutex;
        std::atomic<bool> live_tracking{false};
        std::atomic<uint64_t> live_next_window{0};
        int64_t starved_since_ns{0};  // (take_mutex) since when the channel's next window has been resident without a record; 0: not waiting
    };
    Group* group_for(const gsh_trk_conf& conf, int max_code_length, int* channel);
    // the two halves of a launch; both with the group's handle_mutex held and d_mutex NOT held
    int take_live(Slot& S, uint64_t limit_end, int max_records, gsh_trk_epoch* out);
    void ensure_live(Group* g, bool wait_for_handle = false);  // a residency in flight (and one queued behind it) for the group, if nobody else is at it
    void quiesce_live(Group* g);                       // handle_mutex held
    uint64_t lowest_next_window_live() const;          // no lock: min over the slots' live_next_window
    int begin_launch(Group* g);                        // how far the resident samples let the group run -> gsh_trk_run_begin; returns the periods queued (0: none)
    uint32_t end_and_file(Group* g, uint64_t* most_resident);  // gsh_trk_run_end -> the blocks' queues; returns the records filed
    uint64_t lowest_next_window_locked() const;

    int d_device;
    std::shared_ptr<Hip_Sample_Ring> d_ring;
    int d_periods_per_launch;
    int d_channels_per_group;
    bool d_launch_ahead{true};
    bool d_live{true};
    int d_work_groups_per_channel{1};  // launched mode: gsh_trk_set_split of every handle opened
    bool d_push_try{true};                 // live mode: a block that finds the ring busy does not queue up behind the thread that is appending
    bool d_push_spare_slowest{true};
    int d_push_batch{2};                    // live mode: appends smaller than this many code periods wait for more (while the device has work in hand)
    int d_spin_us{40};                 // how long a block polls for a record before it starts sleeping between looks
    std::atomic<int64_t> d_record_timeout_ns{1000000000};
    int d_sleep_us{20};                // ... and how long each of those sleeps is asked to be
    int d_spin_us_single{0};           // the same two for a block that takes ONE period per call (the reference's cadence): no polling, short sleeps
    int d_sleep_us_single{25};
    int d_single_max_records{4};       // ... "one period per call" = at most this many records asked for (2 and 4 periods per call gain as much from the sleeping wait; from 8 on
                                       // the records of a call arrive over a longer stretch than a sleep and polling wins: profiles/ab/r05/dropin_wait_notes.txt)
    int d_timer_slack_ns{1000};        // the timer slack of a block thread that takes one period per call and sleeps for its record (the kernel's default, 50 us, is longer than
                                       // the whole wait); 0: left alone
    std::atomic<uint64_t> d_min_vlen{0};  // shortest code period (samples) among the loop configurations attached so far
    std::atomic<size_t> d_n_slots{0};  // slots ever created (d_slots never shrinks and is reserved up front: readers without the lock index below this)
    std::atomic<uint64_t> d_live_records{0}, d_live_residencies{0}, d_record_wait_ns{0}, d_record_waits{0};
    mutable std::mutex d_mutex;  // slots, queues, in_flight flags, stats
    std::vector<std::unique_ptr<Group>> d_groups;
    std::vector<std::unique_ptr<Slot>> d_slots;
    std::string d_error;
    Stats d_stats;
    std::chrono::steady_clock::time_point d_last_filed{};  // when the latest launch's records were filed (statistics)
    std::atomic<uint64_t> d_lowest_next_window{UINT64_MAX};  // min over the tracking slots' next windows (UINT64_MAX: none), written under d_mutex
    std::atomic<uint64_t> d_push_ns{0}, d_pushed_samples{0};
};

#endif  // GNSS_SDR_HIP_TRACKING_RUNTIME_H
