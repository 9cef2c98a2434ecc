/*!
 * \file hip_pcps_acquisition_core.h
 * \brief The arithmetic and dwell logic of gnss-sdr's pcps_acquisition block
 *        (src/algorithms/acquisition/gnuradio_blocks/pcps_acquisition.{h,cc}) on an MI355X, without the GNU Radio shell.
 *
 * Method and member names follow the reference block so that the GNU Radio wrapper in
 * gnss_sdr_adapters/pcps_acquisition_hip.cc is a line-for-line shell around this class:
 *   set_local_code        acq.cc:218-251      set_doppler_center  acq.cc:737-746
 *   acquisition_core      acq.cc:648-728      update_synchro      acq.cc:580-602
 *   compute_threshold     acq.cc:52-56
 * Everything numeric goes through the C ABI (include/gnss_sdr_hip.h, gsh_acq_*).  No CPU fallback.
 * make_2_steps (acq.cc:294-301, 605-632) and item_type = cshort (acq.cc:653-656) are handled like the reference does.
 */
#ifndef GNSS_SDR_HIP_PCPS_ACQUISITION_CORE_H
#define GNSS_SDR_HIP_PCPS_ACQUISITION_CORE_H

#include "gnss_sdr_hip.h"
#include <cmath>
#include <complex>
#include <cstdint>
#include <string>
#include <vector>


/*! The members of Acq_Conf (src/algorithms/acquisition/libs/acq_conf.h:33-87) the arithmetic depends on, same names. */
struct Hip_Acq_Conf
{
    int64_t fs_in{4000000LL};
    int64_t resampled_fs{0LL};
    float samples_per_ms{0.0F};
    float threshold{0.0F};
    float pfa{0.0F};
    float samples_per_code{0.0F};
    float resampler_ratio{1.0F};
    uint32_t sampled_ms{1U};
    uint32_t ms_per_code{1U};
    uint32_t samples_per_chip{2U};
    uint32_t chips_per_second{1023000U};
    uint32_t max_dwells{1U};
    uint32_t resampler_latency_samples{0U};
    int32_t doppler_max{5000};
    int32_t doppler_step{500};
    float doppler_step2{125.0F};          //!< acq_conf.h:50, key second_doppler_step
    float pfa2{0.0F};                     //!< acq_conf.h:45, key pfa_second_step
    uint32_t num_doppler_bins_step2{4U};  //!< acq_conf.h:62, key second_nbins
    bool make_2_steps{false};             //!< acq_conf.h:74, key make_two_steps
    bool cshort{false};                   //!< item_type == "cshort" (acq_conf.cc:33-36): acquisition_core takes lv_16sc_t samples
    bool bit_transition_flag{false};
    bool use_CFAR_algorithm_flag{true};
    bool use_automatic_resampler{false};
    bool dump{false};  //!< Acq_Conf::dump: the grid must stay readable (read_grid)
    std::string dump_filename;  //!< Acq_Conf::dump_filename (the adapter has already put the dump directory in front, as the reference's constructor does)
    uint32_t dump_channel{0U};  //!< Acq_Conf::dump_channel: only this channel's searches are dumped (acq.cc:555, 720)

    /*! acq_conf.cc:119-124 */
    void SetDerivedParams()
    {
        if (resampled_fs == 0) resampled_fs = fs_in;
        samples_per_ms = static_cast<float>(resampled_fs) * 0.001F;
        samples_per_chip = static_cast<unsigned int>(std::ceil(static_cast<float>(resampled_fs) / static_cast<float>(chips_per_second)));
        samples_per_code = samples_per_ms * static_cast<float>(ms_per_code);
    }
};

class Hip_Pcps_Acquisition_Core
{
public:
    /*! pcps_acquisition::AcquisitionResult (pcps_acquisition.h) */
    struct AcquisitionResult
    {
        uint64_t sample_count{0};
        float test_statistics{0.0F};
        int32_t doppler{0};
        uint32_t index_time{0};
        bool positive_acq{false};
        bool step_two{false};  //!< the dwell that produced this result ran the narrow grid (d_step_two at acq.cc:598)
        bool search_complete{false};  //!< acq.cc:717: the point at which the reference dumps the search and clears its dwell counter
        uint32_t num_dwells{0};       //!< d_num_noncoherent_integrations_counter at that point (dump variable num_dwells)
    };

    enum Outcome
    {
        ACQ_CONTINUE = 0,  //!< below threshold, more non-coherent dwells allowed (acq.cc:694-698: d_state = 1)
        ACQ_POSITIVE = 1,  //!< send_positive_acquisition, "events" message 1 (acq.cc:318-341)
        ACQ_NEGATIVE = 2,  //!< send_negative_acquisition, "events" message 2 (acq.cc:344-351)
        ACQ_ERROR = -1
    };

    /*! num_doppler_bins_override = 0 keeps the reference's ceil(2*doppler_max/doppler_step) (acq.cc:113) */
    explicit Hip_Pcps_Acquisition_Core(const Hip_Acq_Conf& conf, int device = 0, uint32_t num_doppler_bins_override = 0);
    ~Hip_Pcps_Acquisition_Core();
    Hip_Pcps_Acquisition_Core(const Hip_Pcps_Acquisition_Core&) = delete;
    Hip_Pcps_Acquisition_Core& operator=(const Hip_Pcps_Acquisition_Core&) = delete;

    bool ok() const { return d_handle != nullptr; }
    const std::string& last_error() const { return d_error; }

    void set_local_code(const std::complex<float>* code);
    void set_doppler_center(int32_t doppler_center);
    /*! d_doppler_bias of is_fdma() (acq.cc:252-272): the adapter passes DFRQ1_GLO (or DFRQ2_GLO) * GLONASS_PRN.at(prn), 0 for CDMA signals */
    void set_doppler_bias(int32_t doppler_bias);
    void set_threshold(float threshold) { d_threshold = threshold; }
    void set_resampler_latency(uint32_t latency_samples) { d_acq_parameters.resampler_latency_samples = latency_samples; }  //!< acq.h:168-172
    float get_threshold() const { return d_step_two ? d_threshold_step_two : d_threshold; }  //!< acq.cc:731-734
    void reset()
    {
        d_num_noncoherent_integrations_counter = 0;
        d_step_two = false;
    }
    bool step_two() const { return d_step_two; }

    /*! one dwell over d_consumed_samples samples; the caller does the buffering of acq.cc:790-815.  With make_2_steps a
        threshold crossing in step one returns ACQ_CONTINUE and arms step two for the next block (acq.cc:609-624). */
    Outcome acquisition_core(uint64_t sample_count, const std::complex<float>* data, AcquisitionResult* result);
    /*! The first dwell of a search whose |.|^2 statistics came out of a batch shared with other channels (Hip_Acquisition_Runtime): everything
        acquisition_core does around the dwell -- counter, thresholds, the two-step state machine (acq.cc:668, 686-727) -- with `r` in place of
        this handle's own result.  dwell_ok = false: the shared dwell failed (treated like a failed private dwell). */
    Outcome acquisition_core_shared(uint64_t sample_count, bool dwell_ok, const gsh_acq_result& r, AcquisitionResult* result);
    /*! true when the NEXT dwell may go through a shared batch: the first dwell of step one of a single-dwell search over gr_complex items with the
        statistics formed on chip and the Doppler grid centred where every channel's is (no dump, no set_doppler_center / FDMA offset) */
    bool next_dwell_is_shareable() const
    {
        return d_handle != nullptr && !d_step_two && d_num_noncoherent_integrations_counter == 0 && d_acq_parameters.max_dwells <= 1U && !d_acq_parameters.cshort &&
               !d_acq_parameters.dump && d_doppler_center == 0 && d_doppler_bias == 0;
    }
    /*! the dwell geometry this core gave the engine (what a shared runtime must agree with) */
    const gsh_acq_conf& engine_conf() const { return d_engine_conf; }
    /*! the same for item_type = cshort: `data` holds d_consumed_samples interleaved int16 I,Q pairs (acq.cc:653-656) */
    Outcome acquisition_core(uint64_t sample_count, const std::complex<int16_t>* data, AcquisitionResult* result);

    /*! acq.cc:580-602; Synchro is gnss-sdr's Gnss_Synchro (gnss_synchro.h:38-82) or anything with the same members */
    template <typename Synchro>
    void update_synchro(const AcquisitionResult& result, Synchro* s) const
    {
        s->Acq_delay_samples = static_cast<double>(std::fmod(static_cast<float>(result.index_time), d_acq_parameters.samples_per_code));
        s->Acq_doppler_hz = static_cast<double>(result.doppler);
        if (d_acq_parameters.use_automatic_resampler)
            {
                s->Acq_delay_samples = (s->Acq_delay_samples * d_acq_parameters.resampler_ratio) - static_cast<double>(d_acq_parameters.resampler_latency_samples);
                s->Acq_samplestamp_samples = static_cast<uint64_t>(std::rint(static_cast<double>(result.sample_count) * d_acq_parameters.resampler_ratio));
                s->fs = d_acq_parameters.resampled_fs;
            }
        else
            {
                s->Acq_samplestamp_samples = result.sample_count;
                s->fs = d_acq_parameters.fs_in;
            }
        if (result.step_two) s->Acq_doppler_step = d_acq_parameters.doppler_step2;  // acq.cc:598-601
    }

    /*! dump support (acq.cc:555-558): D rows of d_effective_fft_size floats.  In a make_two_steps search the device's narrow grid takes the place of the wide grid's
     *  first rows (as d_magnitude_grid is reused, acq.cc:526), so a dumping block keeps host copies the way the reference keeps d_grid and d_narrow_grid (acq.cc:555-558):
     *  the wide grid after every step-one dwell, the narrow one after every step-two dwell (keep_dump_grids, called by the acquisition_core entry points). */
    bool read_grid(float* grid);
    /*! acq_grid_narrow of dump_results (acq.cc:392-400): num_doppler_bins_step2 rows of d_effective_fft_size floats -- zeros until a step-two dwell has run, then
     *  whatever the last one left, across searches, as the reference's d_narrow_grid */
    bool read_narrow_grid(float* grid) const;
    bool make_two_steps() const { return d_acq_parameters.make_2_steps; }
    uint32_t num_doppler_bins_step2() const { return d_acq_parameters.num_doppler_bins_step2; }
    float doppler_step2() const { return d_acq_parameters.doppler_step2; }
    float doppler_center_step_two() const { return d_doppler_center_step_two; }

    uint32_t consumed_samples() const { return d_consumed_samples; }
    uint32_t fft_size() const { return d_fft_size; }
    uint32_t effective_fft_size() const { return d_effective_fft_size; }
    uint32_t num_doppler_bins() const { return d_num_doppler_bins; }
    float input_power() const { return d_input_power; }
    int32_t doppler_max() const { return d_acq_parameters.doppler_max; }
    int32_t doppler_step() const { return d_acq_parameters.doppler_step; }

    static float compute_threshold(float pfa, uint32_t effective_fft_size, uint32_t num_doppler_bins, uint32_t max_dwells);

private:
    Hip_Acq_Conf d_acq_parameters;
    gsh_acq_conf d_engine_conf{};
    int32_t d_doppler_bias{0};
    gsh_acq* d_handle{nullptr};
    std::string d_error;
    uint32_t d_consumed_samples{0};
    uint32_t d_fft_size{0};
    uint32_t d_effective_fft_size{0};
    uint32_t d_num_doppler_bins{0};
    uint32_t d_num_noncoherent_integrations_counter{0};
    int32_t d_doppler_center{0};
    float d_threshold{0.0F};
    float d_threshold_step_two{0.0F};
    float d_input_power{0.0F};
    float d_doppler_center_step_two{0.0F};
    bool d_step_two{false};
    std::vector<float> d_dump_grid, d_dump_narrow_grid;  // host copies for dump_results of a make_two_steps search (see read_grid)
    bool d_have_dump_grid{false};
    void keep_dump_grids(bool dwell_was_step_two);
    std::vector<std::complex<float>> d_cshort_scratch;

    Outcome core_after_dwell(uint64_t sample_count, bool dwell_ok, const void* gsh_result, AcquisitionResult* result);
};

#endif  // GNSS_SDR_HIP_PCPS_ACQUISITION_CORE_H
