/*!
 * \file hip_correlator_runtime.h
 * \brief Receiver-side batching runtime for the MI355X correlator engine.
 *
 * In gnss-sdr every channel's tracking block runs on its own GNU Radio thread and calls its own correlator object once per
 * code period (dll_pll_veml_tracking.cc:1232-1257 from general_work, :1975-2001); all of them read the same input buffer
 * (gnss_flowgraph.cc:1227-1231).  One GPU launch per call is latency-bound (launch + synchronisation ~ tens of microseconds
 * for ~1 microsecond of work).  This runtime keeps that threading model and removes the per-call cost:
 *
 *  - Hip_Sample_Ring: the IF stream of one RF front-end, pushed to the device ONCE by whoever owns the input (a sink block
 *    next to the signal conditioner), converted there from the front-end's item type (data_type_adapter arithmetic), and
 *    addressed by absolute sample index -- the same counter the tracking block already keeps (d_sample_counter,
 *    dll_pll_veml_tracking.cc:1963-1978, 2287).
 *  - Hip_Correlator_Runtime: the channel threads rendezvous; the thread whose arrival completes the batch (every channel that is
 *    currently tracking has a job in it) issues ONE gsh_bank_correlate for the whole batch on the spot -- no wake-up sits on the
 *    critical path -- and the first arriver closes it after a bounded wait if a channel is late; every caller returns with its
 *    own taps.  Waiters spin briefly on the batch's completion flag before they block (the launch takes tens of microseconds, a
 *    futex sleep + wake costs as much).  Callers block exactly as they do inside the reference's synchronous correlator call.
 *  - Hip_Multicorrelator_Batched: the eight methods of Cpu_Multicorrelator_Real_Codes (cpu_multicorrelator_real_codes.h:40-49)
 *    on top of the runtime, plus set_input_sample_index(): the window is named by its absolute index instead of a host pointer.
 *
 * Plain C++17 over the C ABI (include/gnss_sdr_hip.h); no HIP headers, no GNU Radio.
 */
#ifndef GNSS_SDR_HIP_CORRELATOR_RUNTIME_H
#define GNSS_SDR_HIP_CORRELATOR_RUNTIME_H

#include "gnss_sdr_hip.h"
#include <atomic>
#include <chrono>
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

class Hip_Sample_Ring
{
public:
    /*! capacity_samples: how much of the stream stays resident; max_window_samples: longest correlation window (2 * vector_length
        covers dll_pll_veml_tracking's forecast, trk.cc:747-754) */
    Hip_Sample_Ring(int device, uint64_t capacity_samples, uint32_t max_window_samples);
    /*! The same stream in SEVERAL GPUs of the node (channels shard over them): one ring per device, kept identical by the engine's
        stream group (gsh_stream_group_*: every appended block enters devices[0] over PCIe ONCE and is replicated to the others over RCCL / xGMI).
        To its users this is still one ring -- one lock, one index space, one push_from that de-duplicates across the blocks of all devices --;
        whoever binds a device handle to it asks for that device's ring with handle_for().  One device: exactly the constructor above. */
    Hip_Sample_Ring(const std::vector<int>& devices, uint64_t capacity_samples, uint32_t max_window_samples);
    ~Hip_Sample_Ring();
    Hip_Sample_Ring(const Hip_Sample_Ring&) = delete;
    Hip_Sample_Ring& operator=(const Hip_Sample_Ring&) = delete;

    bool ok() const { return d_handle != nullptr; }
    /*! (a copy, taken under the error text's own lock: block threads read it while another thread's failing push writes it) */
    std::string last_error() const
    {
        std::lock_guard<std::mutex> lk(d_error_mutex);
        return d_error;
    }
    /*! append samples; returns the absolute index of the first one (UINT64_MAX on failure).  One producer thread. */
    uint64_t push(const std::complex<float>* samples, uint64_t n, bool inverted_spectrum = false);
    uint64_t push_ishort(const int16_t* iq, uint64_t n, bool inverted_spectrum = false);  //!< item_type ishort / cshort
    uint64_t push_ibyte(const int8_t* iq, uint64_t n, bool inverted_spectrum = false);    //!< item_type ibyte / cbyte
    /*! What a block that shares the ring with other channel threads calls with the samples of ITS input buffer, named by their absolute index:
        range check, de-duplication and append happen under ONE acquisition of the ring's lock, so two threads that see the same `next` cannot
        both append.  Samples already resident are skipped (another channel of the stream pushed them); the rest is appended through page-locked
        staging (gsh_stream_push_staged: `samples` may be re-used on return).  A block that runs AHEAD of the ring (first_index > next: the
        samples in between belong to slower siblings that have not pushed them yet) repositions the ring when `may_seek` says nobody needs
        what is resident, and otherwise waits up to `gap_timeout` for the siblings to close the gap (0: does not wait and pushes nothing --
        the siblings' own pushes will cover the stretch).  Returns false on a gap that stays open or on an engine error (last_error()). */
    bool push_from(uint64_t first_index, const std::complex<float>* samples, uint64_t n, bool may_seek,
        std::chrono::milliseconds gap_timeout = std::chrono::milliseconds(200), uint64_t* appended = nullptr, uint64_t* append_ns = nullptr, bool wait_copy = true,
        bool try_only = false);  //!< try_only: when another thread holds the ring (it is appending, most likely the same samples), return at once without appending
    /*! For a caller of push_from(..., wait_copy = false): returns when every sample below `end` that was pushed by DMA out of the caller's own memory has
        been read from it -- the caller may hand that part of its buffer back.  A GNU Radio block gives back only what it consumes; what it was OFFERED beyond
        that (and pushed ahead) stays valid, so its copy need not be waited for in the call that queued it. */
    bool wait_copied_upto(uint64_t end);
    /*! Page-lock [ptr, ptr + bytes) (rounded outwards to pages; parts already locked are skipped) so that push_from can hand it to the DMA
        engine without a staging copy.  A GNU Radio input buffer is the same memory for the whole run: after the first few calls every push is
        a true DMA.  false (and push_from falls back to the staging copy) when the range cannot be registered. */
    bool register_host(const void* ptr, size_t bytes);
    /*! push_from registers whatever part of the range it is handed is not page-locked yet (default off: the caller decides) */
    void set_auto_register(bool on) { d_auto_register = on; }
    /*! position an idle ring: the next pushed sample gets absolute index `next_index`, nothing older is resident */
    bool seek(uint64_t next_index);
    uint64_t capacity() const { return d_capacity; }
    uint32_t max_window() const { return d_max_window; }
    /*! the next index to be pushed, without the lock (the value wait_for() polls) */
    uint64_t next_index() const { return d_next.load(std::memory_order_acquire); }
    /*! the oldest resident index, without the lock (may lag a concurrent push by one call: use it for plausibility, not for addressing) */
    uint64_t oldest_index() const
    {
        const uint64_t next = d_next.load(std::memory_order_acquire), origin = d_origin.load(std::memory_order_acquire);
        const uint64_t by_capacity = next > d_capacity ? next - d_capacity : 0;
        return by_capacity > origin ? by_capacity : origin;
    }
    /*! the ring's lock, for a caller that must keep pushes out while it queues a launch that reads the ring's bookkeeping
        (Hip_Correlator_Runtime, Hip_Tracking_Runtime) */
    std::mutex& mutex() const { return d_mutex; }
    /*! [oldest, next) resident */
    void range(uint64_t* oldest, uint64_t* next) const;
    /*! blocks until sample index `end` has been pushed (next >= end) or the timeout expires */
    bool wait_for(uint64_t end, std::chrono::milliseconds timeout) const;
    gsh_stream_t* handle() const { return d_handle; }
    /*! the ring that lives on `device` (nullptr when the stream is not resident there) */
    gsh_stream_t* handle_for(int device) const;
    const std::vector<int>& devices() const { return d_devices; }
    int device() const { return d_device; }

private:
    uint64_t push_items(const void* items, uint64_t n, int item_type, bool inverted_spectrum);
    // the two ways complex64 items reach the ring(s) from push_from (d_mutex held): out of page-locked memory as it lies, or through a page-locked copy
    int append_pinned(const std::complex<float>* items, uint64_t n, uint64_t* first);
    int append_pageable(const std::complex<float>* items, uint64_t n, uint64_t* first);
    int seek_locked(uint64_t next_index);
    gsh_stream_group_t* d_group{nullptr};  // several devices: the rings belong to it (d_handle = the ingest device's)
    std::vector<int> d_devices;
    // group mode: page-locked staging of our own for items that are not (the group's copy to the ingest GPU reads the caller's memory asynchronously)
    static constexpr int NSTAGE = 4;
    std::vector<std::complex<float>> d_stage[NSTAGE];
    uint64_t d_stage_end[NSTAGE]{};  // absolute index behind the append that last used the buffer
    int d_stage_next{0};
    gsh_stream_t* d_handle{nullptr};
    int d_device{0};
    uint64_t d_capacity{0};
    uint32_t d_max_window{0};
    std::string d_error;
    mutable std::mutex d_error_mutex;  // d_error only (innermost: taken with or without d_mutex held, never the other way round)
    void set_error(const std::string& e)
    {
        std::lock_guard<std::mutex> lk(d_error_mutex);
        d_error = e;
    }
    void reposition_locked(uint64_t next_index);
    mutable std::mutex d_mutex;  // serialises pushes against job translation (the C handle is not thread-safe)
    mutable std::condition_variable d_pushed;
    std::atomic<uint64_t> d_next{0};  // read without the lock on the fast path of wait_for (32 channel threads ask at the same instant)
    std::atomic<uint64_t> d_origin{0};  // first index resident since the last seek
    std::atomic<uint64_t> d_copied_upto{0};  // samples below this index are known to have left the callers' buffers
    // page-locked host ranges: one entry per registration, sorted and disjoint (a DMA must lie inside ONE registration), and what the destructor releases
    uintptr_t piece_end_locked(uintptr_t at) const;
    bool register_locked(uintptr_t a, uintptr_t b);
    std::vector<std::pair<uintptr_t, uintptr_t>> d_pinned;
    std::vector<void*> d_registered;
    bool d_auto_register{false};
    friend class Hip_Correlator_Runtime;
};

class Hip_Correlator_Runtime
{
public:
    struct Stats
    {
        uint64_t batches{0};
        uint64_t jobs{0};
        uint64_t timeouts{0};  //!< batches launched because max_wait expired before every active channel arrived
        uint32_t largest_batch{0};
    };

    /*! max_wait: how long the first arriver of a batch waits for the other active channels before launching with what it has.
        spin_us: how long a waiter polls the completion flag before blocking; -1 = automatic (150 us when the host has at least
        twice as many hardware threads as max_channels, else 0: spinning on an oversubscribed host only steals the launcher's core) */
    Hip_Correlator_Runtime(Hip_Sample_Ring* ring, int max_channels, int max_code_length,
        std::chrono::microseconds max_wait = std::chrono::microseconds(200), int spin_us = -1);
    ~Hip_Correlator_Runtime();
    Hip_Correlator_Runtime(const Hip_Correlator_Runtime&) = delete;
    Hip_Correlator_Runtime& operator=(const Hip_Correlator_Runtime&) = delete;

    bool ok() const { return d_bank != nullptr; }
    const std::string& last_error() const { return d_error; }

    /*! a tracking block entering / leaving state "tracking" (start_tracking / loss of lock): returns the code slot, -1 when full */
    int register_channel();
    void unregister_channel(int channel);
    bool set_code(int channel, const float* code, int code_length);
    /*! one Carrier_wipeoff_multicorrelator_resampler call (mcorr.cc:103-144) of channel `channel`: job.sample_offset is an absolute
        sample index, job.code_slot is overwritten with the channel's slot.  Blocks until the batch it joined has run; out receives
        job.n_taps complex values.  Thread-safe; at most one call in flight per channel. */
    bool correlate(int channel, const gsh_corr_job& job, std::complex<float>* out);
    /*! the same for TWO correlators of one tracking block that run over the same window with the same NCO parameters -- the pilot's
        VE/E/P/L/VL and the single-tap data prompt of track_pilot (trk.cc:1236-1256): both jobs join the batch side by side (the bank
        computes the second inside the first, gsh_bank_set_pair_fusion) and the block spends one rendezvous per epoch instead of two. */
    bool correlate_pair(int channel, const gsh_corr_job& job, std::complex<float>* out, int channel2, const gsh_corr_job& job2, std::complex<float>* out2);
    Stats stats() const;

private:
    struct Batch
    {
        std::vector<gsh_corr_job> jobs;
        std::vector<float> out;  // n_jobs * GSH_MAX_TAPS * 2
        std::mutex m;            // guards done_cv only: the next batch's arrivals never contend with this batch's wake-ups
        std::condition_variable done_cv;
        std::atomic<int> done{0};
        bool taken{false};       // a closer has claimed it (under d_mutex)
        int status{0};
        std::string error;
    };
    bool correlate_n(int n_jobs, const int* channels, const gsh_corr_job* const* jobs, std::complex<float>* const* outs);
    void run_batch(const std::shared_ptr<Batch>& b, bool timed_out);
    std::shared_ptr<Batch> new_batch() const;
    Hip_Sample_Ring* d_ring;
    gsh_bank_t* d_bank{nullptr};
    std::string d_error;
    std::chrono::microseconds d_max_wait;
    int d_spin_us{0};
    mutable std::mutex d_mutex;
    std::condition_variable d_arrived;
    std::shared_ptr<Batch> d_current;
    std::vector<char> d_slot_used;
    int d_active{0};
    std::mutex d_bank_mutex;  // one batch on the bank at a time
    Stats d_stats;
};

class Hip_Multicorrelator_Batched
{
public:
    explicit Hip_Multicorrelator_Batched(Hip_Correlator_Runtime* runtime) : d_runtime(runtime) {}
    ~Hip_Multicorrelator_Batched() { free(); }
    Hip_Multicorrelator_Batched(const Hip_Multicorrelator_Batched&) = delete;
    Hip_Multicorrelator_Batched& operator=(const Hip_Multicorrelator_Batched&) = delete;

    // ---- the reference's eight methods (cpu_multicorrelator_real_codes.h:40-49), same argument order and meaning
    void set_high_dynamics_resampler(bool use_high_dynamics_resampler) { d_use_high_dynamics_resampler = use_high_dynamics_resampler; }
    bool init(int max_signal_length_samples, int n_correlators);
    bool set_local_code_and_taps(int code_length_chips, const float* local_code_in, float* shifts_chips);
    /*! sig_in is accepted for signature compatibility and ignored: the window is named by set_input_sample_index() */
    bool set_input_output_vectors(std::complex<float>* corr_out, const std::complex<float>* sig_in);
    bool Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float phase_rate_step_rad,
        float rem_code_phase_chips, float code_phase_step_chips, float code_phase_rate_step_chips, int signal_length_samples);
    bool Carrier_wipeoff_multicorrelator_resampler(float rem_carrier_phase_in_rad, float phase_step_rad, float rem_code_phase_chips,
        float code_phase_step_chips, float code_phase_rate_step_chips, int signal_length_samples);
    bool free();
    // ---- the one addition: absolute index (in the ring) of the first sample of the window the next call correlates
    void set_input_sample_index(uint64_t index) { d_sample_index = index; }
    /*! track_pilot (trk.cc:1246-1256): `data` is the block's second correlator (d_correlator_data_cpu), called right after this one with
        the same parameters.  Once set, this correlator's call computes both (Hip_Correlator_Runtime::correlate_pair) and leaves the
        companion's result in the companion's output vector; the companion's own call then finds its parameters already served and
        returns at once.  Any call the companion receives with other parameters runs normally.  nullptr detaches. */
    void set_companion(Hip_Multicorrelator_Batched* data) { d_companion = data; }
    const std::string& last_error() const { return d_error; }

private:
    bool run(int mode, float rem_carr, float phase_step, float phase_rate, float rem_code, float code_step, float code_rate, int n);
    Hip_Correlator_Runtime* d_runtime;
    int d_channel{-1};
    int d_max_len{0};
    int d_n_correlators{0};
    float* d_shifts{nullptr};               // borrowed, re-read every call (mcorr.cc:58)
    std::complex<float>* d_corr_out{nullptr};  // borrowed (mcorr.cc:70)
    uint64_t d_sample_index{0};
    bool d_use_high_dynamics_resampler{true};  // same default as the reference (mcorr.h:60)
    std::string d_error;
    Hip_Multicorrelator_Batched* d_companion{nullptr};
    // what the companion mechanism has already computed for THIS correlator (set by the leader, consumed by the next run())
    bool d_served{false};
    gsh_corr_job d_served_job{};
    bool ready() const { return d_channel >= 0 && d_shifts != nullptr && d_corr_out != nullptr; }
    void fill_job(gsh_corr_job* j, int mode, float rem_carr, float phase_step, float phase_rate, float rem_code, float code_step, float code_rate, int n) const;
};

#endif  // GNSS_SDR_HIP_CORRELATOR_RUNTIME_H
