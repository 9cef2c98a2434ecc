/*!
 * \file hip_pcps_detectors.h
 * \brief The dwell logic of the reference's other PCPS detector blocks on the MI355X engine, without the GNU Radio shell:
 *
 *   Hip_Pcps_Tong_Core         pcps_tong_acquisition_cc          (gnuradio_blocks/pcps_tong_acquisition_cc.cc, "tong.cc")
 *   Hip_Galileo_Pcps_8ms_Core  galileo_pcps_8ms_acquisition_cc   (gnuradio_blocks/galileo_pcps_8ms_acquisition_cc.cc, "8ms.cc")
 *   Hip_Pcps_Quicksync_Core    pcps_quicksync_acquisition_cc     (gnuradio_blocks/pcps_quicksync_acquisition_cc.cc, "qs.cc")
 *   Hip_Pcps_Fine_Doppler_Core pcps_acquisition_fine_doppler_cc  (gnuradio_blocks/pcps_acquisition_fine_doppler_cc.cc, "fd.cc")
 *   Hip_Galileo_E5a_Noncoherent_Iq_Core  galileo_e5a_noncoherentIQ_acquisition_caf_cc  (gnuradio_blocks/galileo_e5a_noncoherent_iq_acquisition_caf_cc.cc, "e5a.cc")
 *
 * Each class keeps the reference block's member names and its general_work state numbering: init() is state 0, work() is
 * one pass of state 1 over one input vector and returns the next state (1 = keep going, 2 = positive, 3 = negative).
 * Only the counters live on the host; wipe-off, transforms, |.|^2, power normalisation, grid accumulation, arg-max and the
 * input-power estimate run on the GPU through the C ABI (gsh_acq_*).  No CPU fallback.
 */
#ifndef GNSS_SDR_HIP_PCPS_DETECTORS_H
#define GNSS_SDR_HIP_PCPS_DETECTORS_H

#include "gnss_sdr_hip.h"
#include "hip_pcps_acquisition_core.h"
#include <complex>
#include <cstdint>
#include <string>
#include <vector>

/*! ThresholdComputeQuickSync::calculate_threshold (adapters/base_pcps_acquisition_custom.cc:118-140) */
float hip_threshold_compute_quicksync(float pfa, uint32_t code_length, uint32_t folding_factor, int32_t doppler_max, int32_t doppler_step);

/*! ThresholdComputeDoppler::calculate_threshold (adapters/base_pcps_acquisition_custom.cc:89-112) */
float hip_threshold_compute_doppler(float pfa, uint32_t vector_length, int32_t doppler_max, int32_t doppler_step);

/*! what both blocks write into their Gnss_Synchro (tong.cc:259-264, 8ms.cc:254-259) */
struct Hip_Detector_Result
{
    double Acq_delay_samples{0.0};
    double Acq_doppler_hz{0.0};
    uint64_t Acq_samplestamp_samples{0ULL};
    uint32_t Acq_doppler_step{0U};
    uint32_t index_time{0U};
    uint32_t index_doppler{0U};
};

class Hip_Pcps_Tong_Core
{
public:
    Hip_Pcps_Tong_Core(const Hip_Acq_Conf& conf, uint32_t tong_init_val, uint32_t tong_max_val, uint32_t tong_max_dwells, int device = 0);
    ~Hip_Pcps_Tong_Core();
    Hip_Pcps_Tong_Core(const Hip_Pcps_Tong_Core&) = delete;
    Hip_Pcps_Tong_Core& operator=(const Hip_Pcps_Tong_Core&) = delete;

    bool ok() const { return d_handle != nullptr; }
    const std::string& last_error() const { return d_error; }

    void set_local_code(const std::complex<float>* code);  //!< tong.cc:136-144
    void init();                                           //!< state 0, tong.cc:162-184
    /*! state 1, tong.cc:187-301; sample_counter is d_sample_counter after the block's increment (:200).  Returns d_state, -1 on error */
    int work(uint64_t sample_counter, const std::complex<float>* in);

    const Hip_Detector_Result& result() const { return d_result; }
    uint32_t num_doppler_bins() const { return d_num_doppler_bins; }
    uint32_t fft_size() const { return d_fft_size; }
    uint32_t dwell_count() const { return d_dwell_count; }
    uint32_t tong_count() const { return d_tong_count; }
    float mag() const { return d_mag; }
    float input_power() const { return d_input_power; }
    float test_statistics() const { return d_test_statistics; }
    int state() const { return d_state; }

private:
    Hip_Acq_Conf d_acq_params;
    gsh_acq* d_handle{nullptr};
    std::string d_error;
    Hip_Detector_Result d_result;
    float d_mag{0.0F}, d_input_power{0.0F}, d_test_statistics{0.0F};
    int d_state{0};
    uint32_t d_dwell_count{0}, d_tong_init_val, d_tong_max_val, d_tong_max_dwells, d_tong_count;
    uint32_t d_fft_size{0}, d_num_doppler_bins{0};
};

class Hip_Galileo_Pcps_8ms_Core
{
public:
    explicit Hip_Galileo_Pcps_8ms_Core(const Hip_Acq_Conf& conf, int device = 0);
    ~Hip_Galileo_Pcps_8ms_Core();
    Hip_Galileo_Pcps_8ms_Core(const Hip_Galileo_Pcps_8ms_Core&) = delete;
    Hip_Galileo_Pcps_8ms_Core& operator=(const Hip_Galileo_Pcps_8ms_Core&) = delete;

    bool ok() const { return d_handle != nullptr; }
    const std::string& last_error() const { return d_error; }

    void set_local_code(const std::complex<float>* code);  //!< 8ms.cc:103-131: code A as given, code B with the second period inverted
    void init();                                           //!< state 0, 8ms.cc:147-160
    int work(uint64_t sample_counter, const std::complex<float>* in);  //!< state 1, 8ms.cc:163-287

    const Hip_Detector_Result& result() const { return d_result; }
    uint32_t num_doppler_bins() const { return d_num_doppler_bins; }
    uint32_t fft_size() const { return d_fft_size; }
    int winning_code() const { return d_winning_code; }  //!< 0 = A, 1 = B (not exposed by the reference; for tests)
    float mag() const { return d_mag; }
    float input_power() const { return d_input_power; }
    float test_statistics() const { return d_test_statistics; }
    int state() const { return d_state; }

private:
    Hip_Acq_Conf d_acq_params;
    gsh_acq* d_handle{nullptr};
    std::string d_error;
    Hip_Detector_Result d_result;
    std::vector<float> d_peak_a, d_peak_b;
    std::vector<uint32_t> d_index_a, d_index_b;
    std::vector<std::complex<float>> d_code_b;
    float d_mag{0.0F}, d_input_power{0.0F}, d_test_statistics{0.0F};
    int d_state{0}, d_winning_code{0};
    uint32_t d_well_count{0};
    uint32_t d_fft_size{0}, d_num_doppler_bins{0};
};

/*!
 * \brief The search of pcps_cccwsr_acquisition_cc (coherent channel combining with sign recovery,
 * pcps_cccwsr_acquisition_cc.cc:137-373, cited as cccwsr.cc) on the HIP engine.
 * The block forms data_corr + j pilot_corr and data_corr - j pilot_corr per cell (cccwsr.cc:235-244); the correlation is linear
 * in the local code, so these are the correlations with (data - j pilot) and (data + j pilot): slot 0 and slot 1 of one dwell.
 */
class Hip_Pcps_Cccwsr_Core
{
public:
    explicit Hip_Pcps_Cccwsr_Core(const Hip_Acq_Conf& conf, int device = 0);
    ~Hip_Pcps_Cccwsr_Core();
    Hip_Pcps_Cccwsr_Core(const Hip_Pcps_Cccwsr_Core&) = delete;
    Hip_Pcps_Cccwsr_Core& operator=(const Hip_Pcps_Cccwsr_Core&) = delete;

    bool ok() const { return d_handle != nullptr; }
    const std::string& last_error() const { return d_error; }

    void set_local_code(const std::complex<float>* code_data, const std::complex<float>* code_pilot);  //!< cccwsr.cc:116-134
    void init();                                                                                       //!< state 0, cccwsr.cc:152-164
    int work(uint64_t sample_counter, const std::complex<float>* in);                                  //!< state 1, cccwsr.cc:166-306

    const Hip_Detector_Result& result() const { return d_result; }
    uint32_t num_doppler_bins() const { return d_num_doppler_bins; }
    uint32_t fft_size() const { return d_fft_size; }
    int winning_branch() const { return d_winning_branch; }  //!< 0 = data + j pilot, 1 = data - j pilot (not exposed by the reference; for tests)
    float mag() const { return d_mag; }
    float input_power() const { return d_input_power; }
    float test_statistics() const { return d_test_statistics; }
    int state() const { return d_state; }

private:
    Hip_Acq_Conf d_acq_params;
    gsh_acq* d_handle{nullptr};
    std::string d_error;
    Hip_Detector_Result d_result;
    std::vector<float> d_peak_plus, d_peak_minus;
    std::vector<uint32_t> d_index_plus, d_index_minus;
    std::vector<std::complex<float>> d_code_combined;
    float d_mag{0.0F}, d_input_power{0.0F}, d_test_statistics{0.0F};
    int d_state{0}, d_winning_branch{0};
    uint32_t d_well_count{0};
    uint32_t d_fft_size{0}, d_num_doppler_bins{0};
};

class Hip_Pcps_Quicksync_Core
{
public:
    /*! code_length = Acq_Conf::code_length (samples per code period), as pcps_quicksync_make_acquisition_cc reads it (qs.cc:52) */
    Hip_Pcps_Quicksync_Core(const Hip_Acq_Conf& conf, uint32_t code_length, uint32_t folding_factor, uint32_t max_dwells, int device = 0);
    ~Hip_Pcps_Quicksync_Core();
    Hip_Pcps_Quicksync_Core(const Hip_Pcps_Quicksync_Core&) = delete;
    Hip_Pcps_Quicksync_Core& operator=(const Hip_Pcps_Quicksync_Core&) = delete;

    bool ok() const { return d_handle != nullptr; }
    const std::string& last_error() const { return d_error; }

    void set_local_code(const std::complex<float>* code);  //!< qs.cc:133-156: keeps the unfolded code, hands the folded one to the engine
    void init();                                           //!< state 0, qs.cc:180-192
    /*! state 1, qs.cc:195-395 over folding_factor * code_length samples */
    int work(uint64_t sample_counter, const std::complex<float>* in);

    const Hip_Detector_Result& result() const { return d_result; }
    uint32_t num_doppler_bins() const { return d_num_doppler_bins; }
    uint32_t fft_size() const { return d_fft_size; }
    uint32_t input_length() const { return d_samples_per_code * d_folding_factor; }
    const std::vector<float>& corr_output_f() const { return d_corr_output_f; }  //!< |.|^2 of the alias correlations (qs.cc:332)
    float mag() const { return d_mag; }
    float input_power() const { return d_input_power; }
    float test_statistics() const { return d_test_statistics; }
    int state() const { return d_state; }

private:
    Hip_Acq_Conf d_acq_params;
    gsh_acq* d_handle{nullptr};
    std::string d_error;
    Hip_Detector_Result d_result;
    std::vector<std::complex<float>> d_code, d_code_folded, d_accumulator;
    std::vector<float> d_peak, d_corr_output_f;
    std::vector<uint32_t> d_index, d_possible_delay;
    float d_mag{0.0F}, d_input_power{0.0F}, d_test_statistics{0.0F};
    int d_state{0};
    uint32_t d_samples_per_code, d_folding_factor, d_max_dwells, d_well_count{0};
    uint32_t d_fft_size{0}, d_num_doppler_bins{0};
};

/*!
 * GPS_L1_CA_PCPS_Acquisition_Fine_Doppler.  Members follow the block: compute_and_accumulate_grid (state 1, fd.cc:266-305),
 * compute_CAF (state 2, :182-251), estimate_Doppler (state 3, :316-389).
 *
 * consistent_grid: the block as written wipes bin i off at doppler_step * i - doppler_step (:170) but reports i * doppler_step -
 * doppler_max (:243), so with doppler_max != doppler_step the reported grid Doppler is not the one that was searched and the fine
 * estimate fails its 1 kHz plausibility check (:376).  false (default) reproduces the file to the letter; true searches the grid
 * that :243 reports.
 */
class Hip_Pcps_Fine_Doppler_Core
{
public:
    explicit Hip_Pcps_Fine_Doppler_Core(const Hip_Acq_Conf& conf, bool consistent_grid = false, int device = 0);
    ~Hip_Pcps_Fine_Doppler_Core();
    Hip_Pcps_Fine_Doppler_Core(const Hip_Pcps_Fine_Doppler_Core&) = delete;
    Hip_Pcps_Fine_Doppler_Core& operator=(const Hip_Pcps_Fine_Doppler_Core&) = delete;

    bool ok() const { return d_handle != nullptr; }
    const std::string& last_error() const { return d_error; }

    void set_local_code(const std::complex<float>* code);  //!< fd.cc:130-136; the replica is also kept for estimate_Doppler (:330-345)
    void reset_grid();                                     //!< state 0, fd.cc:149-160, 438-448
    /*! state 1: one block of d_fft_size samples into the grid and into the 10 ms buffer; returns the next state (1 or 2) */
    int compute_and_accumulate_grid(const std::complex<float>* in);
    /*! state 2: d_test_statistics and the Gnss_Synchro fields (:238-249); returns 3 (above threshold) or 5 */
    int compute_CAF(uint64_t sample_counter);
    /*! state 3: append up to 10 ms of further samples (:473-483); returns how many were taken */
    uint32_t buffer_more(const std::complex<float>* in, uint32_t n_items);
    bool buffer_full() const { return d_10_ms_buffer.size() >= 10U * d_fft_size; }
    /*! state 3 -> 4: the zero-padded fine transform; false when it could not run (the grid Doppler then stands) */
    bool estimate_Doppler();

    const Hip_Detector_Result& result() const { return d_result; }
    int num_doppler_points() const { return d_num_doppler_points; }
    uint32_t fft_size() const { return d_fft_size; }
    float test_statistics() const { return d_test_statistics; }
    float fine_doppler_hz() const { return d_fine_doppler_hz; }
    int well_count() const { return d_well_count; }

private:
    Hip_Acq_Conf d_acq_params;
    gsh_acq* d_handle{nullptr};
    std::string d_error;
    Hip_Detector_Result d_result;
    std::vector<std::complex<float>> d_code, d_10_ms_buffer;
    float d_test_statistics{0.0F}, d_fine_doppler_hz{0.0F}, d_last_statistic{0.0F};
    uint32_t d_last_index_time{0}, d_last_index_doppler{0};
    int d_device{0}, d_num_doppler_points{0}, d_well_count{0};
    uint32_t d_fft_size{0};
};

/*!
 * \brief The search of galileo_e5a_noncoherentIQ_acquisition_caf_cc (e5a.cc: set_local_code :162-222, state 2 :300-665) on the HIP engine.
 * The block's local codes -- data "A" (1,1,1), pilot "A", and for coherent times above one code period the "B" combinations with the
 * first period inverted -- sit in up to four slots of ONE dwell over shared forward transforms; per Doppler bin the A / B choice of
 * each component, the addition of the two kept magnitude rows and the arg-max of the sum run on the device
 * (gsh_acq_noncoherent_pair_peaks); the bin loop, the statistic, the CAF filter across bins (:546-631) and the dwell counting are here,
 * in the block's own float / double mix.  Reference behaviour kept as written: the B codes overwrite only the first code period of the
 * FFT input buffer (with both components code I-B is therefore [-I, Q, Q]), and Q-B is ranked by the I-B row (:393).
 * States as in the block: init() = state 0, work() = one pass of state 2 over a full block; returns 1 (another dwell), 3 (positive), 4 (negative).
 */
class Hip_Galileo_E5a_Noncoherent_Iq_Core
{
public:
    Hip_Galileo_E5a_Noncoherent_Iq_Core(const Hip_Acq_Conf& conf, bool both_signal_components, int CAF_window_hz, int Zero_padding, int device = 0);
    ~Hip_Galileo_E5a_Noncoherent_Iq_Core();
    Hip_Galileo_E5a_Noncoherent_Iq_Core(const Hip_Galileo_E5a_Noncoherent_Iq_Core&) = delete;
    Hip_Galileo_E5a_Noncoherent_Iq_Core& operator=(const Hip_Galileo_E5a_Noncoherent_Iq_Core&) = delete;

    bool ok() const { return d_handle != nullptr; }
    const std::string& last_error() const { return d_error; }

    void set_local_code(const std::complex<float>* codeI, const std::complex<float>* codeQ);  //!< e5a.cc:162-222
    void init();                                                                               //!< state 0, e5a.cc:262-274
    int work(uint64_t sample_counter, const std::complex<float>* in);                          //!< state 2, e5a.cc:300-665

    const Hip_Detector_Result& result() const { return d_result; }
    uint32_t num_doppler_bins() const { return d_num_doppler_bins; }
    uint32_t fft_size() const { return d_fft_size; }
    uint32_t well_count() const { return d_well_count; }
    float mag() const { return d_mag; }
    float input_power() const { return d_input_power; }
    float test_statistics() const { return d_test_statistics; }
    int state() const { return d_state; }
    const std::vector<float>& CAF_vector() const { return d_CAF_vector; }

private:
    void caf_filter();
    Hip_Acq_Conf d_acq_params;
    gsh_acq* d_handle{nullptr};
    std::string d_error;
    Hip_Detector_Result d_result;
    std::vector<gsh_acq_pair_peak> d_pair;
    std::vector<std::complex<float>> d_inbuf;  //!< d_fft_if's input buffer: what the previous code left there stays (e5a.cc:187-222)
    std::vector<float> d_CAF_vector, d_CAF_vector_I, d_CAF_vector_Q;
    float d_mag{0.0F}, d_input_power{0.0F}, d_test_statistics{0.0F};
    int d_state{0};
    int d_CAF_window_hz{0};
    int32_t d_slot_IA{0}, d_slot_QA{-1}, d_slot_IB{-1}, d_slot_QB{-1};
    uint32_t d_well_count{0}, d_sampled_ms{1};
    uint32_t d_fft_size{0}, d_num_doppler_bins{0};
    bool d_both_signal_components{false};
};

#endif
