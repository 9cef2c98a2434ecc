/*!
 * \file pcps_acquisition_hip.h
 * \brief GNU Radio shell around Hip_Pcps_Acquisition_Core: the MI355X counterpart of gnss-sdr's pcps_acquisition block
 *        (src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.h:93-251).
 *
 * BUILT ONLY INSIDE A gnss-sdr TREE (needs GNU Radio, pmt, Gnss_Synchro, ChannelFsm; none are in this repository's
 * image).  See INTEGRATION.md for the CMake lines.  It derives from the reference's own abstract block
 * acquisition_impl_interface (acquisition_impl_interface.h:50-64), so the existing adapters' plumbing
 * (BasePcpsAcquisition-style connect/disconnect, Channel, ChannelFsm) works with it unchanged.
 * Same stream contract as the reference block: 1 input of gr_complex (lv_16sc_t for item_type = cshort), 0..1 output of Gnss_Synchro, message port
 * "events" carrying 1 (positive) / 2 (negative) unless a ChannelFsm is set (acq.cc:102-104,146,318-351).
 */
#ifndef GNSS_SDR_PCPS_ACQUISITION_HIP_H
#define GNSS_SDR_PCPS_ACQUISITION_HIP_H

#include "acquisition_impl_interface.h"
#include "channel_fsm.h"
#include "gnss_synchro.h"
#include "hip_acquisition_runtime.h"
#include "hip_pcps_acquisition_core.h"
#include <gnuradio/block.h>
#include <complex>
#include <memory>
#include <string>
#include <vector>

class pcps_acquisition_hip;
using pcps_acquisition_hip_sptr = gnss_shared_ptr<pcps_acquisition_hip>;

pcps_acquisition_hip_sptr pcps_make_acquisition_hip(const Hip_Acq_Conf& conf, int device, bool blocking_on_standby,
    std::shared_ptr<Hip_Acquisition_Runtime> runtime = nullptr);

class pcps_acquisition_hip : public acquisition_impl_interface
{
public:
    ~pcps_acquisition_hip() override;

    void set_gnss_synchro(Gnss_Synchro* p_gnss_synchro) override
    {
        gr::thread::scoped_lock lock(d_setlock);
        d_gnss_synchro = p_gnss_synchro;
    }
    void set_channel(uint32_t channel_id) override
    {
        gr::thread::scoped_lock lock(d_setlock);
        d_channel = channel_id;
    }
    void set_channel_fsm(std::weak_ptr<ChannelFsm> channel_fsm) override
    {
        gr::thread::scoped_lock lock(d_setlock);
        d_channel_fsm = std::move(channel_fsm);
    }
    void set_local_code(std::complex<float>* code) override;
    uint32_t mag() const override { return 0; }
    void set_active(bool active) override;
    void set_doppler_center(int32_t doppler_center);
    void set_threshold(float threshold);
    void set_state(int32_t state);
    void set_resampler_latency(uint32_t latency_samples)
    {
        gr::thread::scoped_lock lock(d_setlock);
        d_core.set_resampler_latency(latency_samples);
    }
    /*! false when the engine could not be created (no GPU, unsupported transform length): the adapter then reports item_size() == 0 */
    bool ok() const { return d_core.ok(); }
    //! the rendezvous this block shares with the other channels of its stream (nullptr: none)
    const std::shared_ptr<Hip_Acquisition_Runtime>& runtime() const { return d_runtime; }

    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
        gr_vector_void_star& output_items) override;

private:
    friend pcps_acquisition_hip_sptr pcps_make_acquisition_hip(const Hip_Acq_Conf& conf, int device, bool blocking_on_standby, std::shared_ptr<Hip_Acquisition_Runtime> runtime);
    pcps_acquisition_hip(const Hip_Acq_Conf& conf, int device, bool blocking_on_standby, std::shared_ptr<Hip_Acquisition_Runtime> runtime);
    void run_dwell(uint64_t sample_count, const std::shared_ptr<Hip_Acquisition_Runtime>& runtime, int slot, uint64_t window, bool shared_dwell);
    void leave_shared_window();

    // channels of one stream that search at the same time share their dwell batches (hip_acquisition_runtime.h); nullptr: every dwell on d_core's own handle
    std::shared_ptr<Hip_Acquisition_Runtime> d_runtime;
    int d_slot{-1};
    uint64_t d_window{0};       // first sample of the window being buffered (a line of the runtime's grid)
    uint32_t d_skip{0};         // samples still to pass before the window starts
    bool d_shared_dwell{false}; // the dwell of the window being buffered goes through the shared batch

    Hip_Pcps_Acquisition_Core d_core;
    std::vector<std::complex<float>> d_data_buffer;
    std::vector<std::complex<int16_t>> d_data_buffer_sc;  // item_type = cshort (acq.cc:161-164)
    bool d_cshort{false};
    std::weak_ptr<ChannelFsm> d_channel_fsm;
    Gnss_Synchro* d_gnss_synchro{nullptr};
    uint64_t d_sample_count{0};
    uint32_t d_buffer_count{0};
    uint32_t d_channel{0};
    // dump (acq.cc:354-406): one .mat file per completed search of the dumped channel
    std::string d_dump_filename;
    uint32_t d_dump_channel{0};
    bool d_dump{false};
    int64_t d_dump_number{0};
    void dump_results(const Hip_Pcps_Acquisition_Core::AcquisitionResult& result);
    int32_t d_state{0};
    bool d_active{false};
    bool d_blocking_on_standby{false};
};

#endif  // GNSS_SDR_PCPS_ACQUISITION_HIP_H
