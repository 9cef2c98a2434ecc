/*!
 * \file hip_tracking_loop.h
 * \brief Host-side class over the device-closed DLL/PLL loop (gsh_trk_* of gnss_sdr_hip.h) for ONE tracking channel.
 *
 * What it stands for in gnss-sdr: the signal-processing half of dll_pll_veml_tracking (src/algorithms/tracking/gnuradio_blocks/
 * dll_pll_veml_tracking.cc, "trk.cc") -- start_tracking (:793-1110), the pull-in alignment (:1949-1978) and the per-period
 * work of states 2 / 3 / 4 (:1980-2252: correlation, lock detectors, run_dll_pll, update_tracking_vars, symbol synchronisation) --
 * without the GNU Radio shell.  The shell (gnss_sdr_adapters/dll_pll_veml_tracking_hip.{h,cc}) owns ports, forecast and the
 * Gnss_Synchro items; this class owns the device state and the sample feed.
 *
 * Sample feed: the block's input buffer is the only place the samples live on the host, so every general_work call hands the samples
 * it can see to push(); they go to a device ring addressed by ABSOLUTE sample index (= nitems_read of the block's input).  Several
 * channels of one RF stream may share one ring (Hip_Sample_Ring): a push that finds its samples already resident is a no-op, so
 * 32 channels upload the stream once.
 *
 * No CPU fallback: every method fails (ok() false / returns false, last_error() says why) when no HIP device is present.
 */
#ifndef GNSS_SDR_HIP_TRACKING_LOOP_H
#define GNSS_SDR_HIP_TRACKING_LOOP_H

#include "gnss_sdr_hip.h"
#include "hip_correlator_runtime.h"  // Hip_Sample_Ring
#include <complex>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

class Hip_Tracking_Loop
{
public:
    /*! device: HIP device index; conf: the loop configuration (see hip_fill_trk_conf in gnss_sdr_adapters/dll_pll_conf_hip.h for the
        Dll_Pll_Conf mapping); max_code_length: floats of the longest local replica (chips x samples per chip);
        shared_ring: a ring fed by somebody else (or by the other channels of the same stream), or nullptr: the loop creates its own,
        ring_capacity_samples long (0: 8 correlation windows). */
    Hip_Tracking_Loop(int device, const gsh_trk_conf& conf, int max_code_length, std::shared_ptr<Hip_Sample_Ring> shared_ring = nullptr,
        uint64_t ring_capacity_samples = 0);
    ~Hip_Tracking_Loop();
    Hip_Tracking_Loop(const Hip_Tracking_Loop&) = delete;
    Hip_Tracking_Loop& operator=(const Hip_Tracking_Loop&) = delete;

    bool ok() const { return d_trk != nullptr; }
    const std::string& last_error() const { return d_error; }
    const gsh_trk_conf& conf() const { return d_conf; }

    /*! start_tracking + pull-in (trk.cc:793-866, 1949-1978).  nitems_read: absolute index of the first sample the block can see now;
        acq_*: what acquisition left in Gnss_Synchro.  code / data_code: the local replica(s) (code_length floats; data_code nullptr unless
        conf.track_pilot).  On return *samples_offset is what the block must consume to align the stream with the replica; the first code
        period starts at nitems_read + *samples_offset. */
    bool start(const float* code, const float* data_code, int code_length, uint64_t nitems_read, double acq_delay_samples,
        double acq_doppler_hz, uint64_t acq_samplestamp_samples, int32_t* samples_offset);
    /*! stop_tracking (trk.cc:1113-1116) */
    void stop();
    bool tracking() const { return d_tracking; }

    /*! make samples [first_index, first_index + n) resident on the device (no-op for the part that already is).  The indices must follow
        the ring's content without a gap. */
    bool push(const std::complex<float>* samples, uint64_t first_index, uint64_t n);

    /*! run up to max_periods code periods whose correlation windows are resident; records[max_periods].  Returns the number of periods
        completed (0 when the next window is not yet resident), -1 on error.  A record with flags bit 1 set is a loss of lock: the
        channel has stopped (tracking() turns false). */
    int run(int max_periods, gsh_trk_epoch* records);

    /*! absolute index of the first sample of the next correlation window */
    uint64_t next_window() const { return d_next_window; }
    Hip_Sample_Ring* ring() const { return d_ring.get(); }

private:
    bool fail(const char* what);
    gsh_trk_conf d_conf{};
    gsh_trk_t* d_trk{nullptr};
    std::shared_ptr<Hip_Sample_Ring> d_ring;
    std::string d_error;
    uint64_t d_next_window{0};
    bool d_tracking{false};
    std::vector<int32_t> d_done;
};

#endif  // GNSS_SDR_HIP_TRACKING_LOOP_H
