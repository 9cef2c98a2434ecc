/*!
 * \file hip_acquisition_runtime.h
 * \brief Rendezvous of the acquisition blocks of one stream: channels that search the same input block share ONE dwell batch.
 *
 * In gnss-sdr every channel owns a pcps_acquisition block; each dwell computes the D Doppler-wiped forward transforms of ITS input block
 * and then the D code correlations (src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.cc:522-560, "acq.cc").  Channels that are
 * searching at the same time read the same stream (gnss_flowgraph.cc:1227-1231), so the forward half is the same work done once per
 * channel.  The engine's dwell takes n local codes per batch (gsh_acq_dwell_slots): the forward transforms once, n x D cells after them,
 * one launch and one synchronisation instead of n.  What keeps the blocks from using it on their own is that each starts buffering
 * wherever its read pointer happens to be, so no two of them hold the same samples.  Hence two rules:
 *
 *   - blocks that share a runtime cut the stream on a common grid: a dwell window starts at a multiple of the dwell length (absolute sample
 *     index); a block that becomes active skips ahead to the next grid line (less than one dwell length) before it buffers.  Which samples a
 *     search looks at is scheduler-dependent in the reference too; the stamp the block reports (Acq_samplestamp_samples) is that of ITS window
 *     as before;
 *   - a block announces the window it is buffering.  When its buffer is full it joins that window's batch; the batch runs as soon as every
 *     block that announced the window has joined, or when max_wait has passed since the first one did (a block that was deactivated in
 *     between withdraws).  Every participant holds the same samples; the first one's buffer is uploaded.
 *
 * Only the first dwell of a search with the statistics formed on chip goes through the batch (max_dwells = 1, no dump, no fine-Doppler step
 * in flight): everything else -- non-coherent accumulation, step two, cshort items -- stays on the block's own handle as before.
 * Plain C++17 over the C ABI; no HIP headers, no GNU Radio.  No CPU fallback.
 */
#ifndef GNSS_SDR_HIP_ACQUISITION_RUNTIME_H
#define GNSS_SDR_HIP_ACQUISITION_RUNTIME_H

#include "gnss_sdr_hip.h"
#include <chrono>
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

class Hip_Acquisition_Runtime
{
public:
    struct Stats
    {
        uint64_t batches{0};      //!< dwell batches launched (= sets of D forward transforms computed)
        uint64_t dwells{0};       //!< channel dwells served by them
        uint64_t timeouts{0};     //!< batches closed by max_wait before every announced channel had joined
        uint32_t largest_batch{0};
    };

    /*! conf: the dwell geometry every block of this runtime uses (fft_size, bins, statistic ...); max_prn and no_grid are set here.
        max_channels: slots (local codes) of the shared handle. */
    Hip_Acquisition_Runtime(int device, const gsh_acq_conf& conf, int max_channels, std::chrono::microseconds max_wait = std::chrono::microseconds(2000));
    ~Hip_Acquisition_Runtime();
    Hip_Acquisition_Runtime(const Hip_Acquisition_Runtime&) = delete;
    Hip_Acquisition_Runtime& operator=(const Hip_Acquisition_Runtime&) = delete;

    bool ok() const { return d_handle != nullptr; }
    const std::string& last_error() const { return d_error; }
    /*! true when `other` describes the same dwell (what decides whether a block may join this runtime) */
    bool same_geometry(const gsh_acq_conf& other) const;
    uint32_t window_length() const { return d_conf.consumed_samples; }
    /*! first grid line at or after `sample_index` */
    uint64_t next_window(uint64_t sample_index) const;

    int attach();            //!< a slot for one block, -1 when full
    void detach(int slot);
    /*! acq.cc:218-251 for the slot's satellite; the Doppler centre / FDMA bias are properties of the shared grid and must be 0 */
    bool set_local_code(int slot, const std::complex<float>* code);
    /*! the block has been activated (Channel: set_active(true)) and will announce a window at its next scheduler call: a batch that is about
        to close waits for it (bounded by max_wait) -- it may well be heading for the same window */
    void searching(int slot);
    /*! the block starts buffering the window that begins at window_start (a grid line) */
    void announce(int slot, uint64_t window_start);
    /*! the block left the search before its dwell (set_active(false), set_state(0)) */
    void withdraw(int slot);
    /*! the block's buffer holds the announced window: join its batch, wait for the batch, take this slot's result */
    bool dwell(int slot, uint64_t window_start, const std::complex<float>* window, gsh_acq_result* out);
    Stats stats() const;

private:
    struct Batch
    {
        std::vector<uint32_t> slots;
        std::vector<gsh_acq_result> results;
        const std::complex<float>* data{nullptr};
        std::chrono::steady_clock::time_point first_arrival{};
        bool taken{false}, done{false};
        int status{0};
        std::string error;
    };
    void run(const std::shared_ptr<Batch>& b, std::unique_lock<std::mutex>& lk, bool timed_out);
    int announced_for(uint64_t window_start) const;  // slots still expected in this window's batch (announced it, or activated and undecided)

    int d_device;
    gsh_acq_conf d_conf{};
    gsh_acq_t* d_handle{nullptr};
    std::string d_error;
    std::chrono::microseconds d_max_wait;
    mutable std::mutex d_mutex;
    std::condition_variable d_cv;
    std::mutex d_handle_mutex;  // the C handle: one thread at a time
    std::vector<char> d_used;
    std::vector<int64_t> d_announced;  // per slot: window start; -1 none; -2 activated, window not known yet
    std::map<uint64_t, std::shared_ptr<Batch>> d_batches;
    Stats d_stats;
};

#endif  // GNSS_SDR_HIP_ACQUISITION_RUNTIME_H
