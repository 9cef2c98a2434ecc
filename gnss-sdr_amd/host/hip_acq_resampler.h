/*!
 * \file hip_acq_resampler.h
 * \brief Host-side design of the acquisition resampler GNSSFlowgraph puts in front of a channel's acquisition when
 *        GNSS-SDR.use_acquisition_resampler is set (src/core/receiver/gnss_flowgraph.cc:1116-1211): the decimation factor, the
 *        low-pass taps and the latency handed to AcquisitionInterface::set_resampler_latency.  The filtering itself runs on the GPU
 *        (gsh_fir_*, csrc/fir_filter.hip: gr::filter::fir_filter_ccf with decimation).
 *
 * The taps come from gr::filter::firdes::low_pass (GNU Radio gr-filter, not vendored in the reference tree): a Hamming-windowed
 * sinc whose length follows from the transition width, normalised to the requested DC gain.  That published algorithm is restated
 * here (parity unpinned: no GNU Radio in this image; the taps are compared with scipy.signal.firwin in tests/test_resampler.py).
 */
#ifndef GNSS_SDR_HIP_ACQ_RESAMPLER_H
#define GNSS_SDR_HIP_ACQ_RESAMPLER_H

#include <cmath>
#include <cstdint>
#include <vector>

/*! gr::filter::firdes::low_pass(gain, sampling_freq, cutoff_freq, transition_width) with the default WIN_HAMMING window */
inline std::vector<float> hip_firdes_low_pass(double gain, double sampling_freq, double cutoff_freq, double transition_width)
{
    // firdes::compute_ntaps: attenuation of the Hamming window is taken as 53 dB, the count is made odd
    int ntaps = static_cast<int>(53.0 * sampling_freq / (22.0 * transition_width));
    if ((ntaps & 1) == 0) ntaps++;
    std::vector<float> taps(static_cast<size_t>(ntaps));
    std::vector<float> w(static_cast<size_t>(ntaps));
    const double pi = 3.14159265358979323846;
    for (int n = 0; n < ntaps; n++) w[n] = static_cast<float>(0.54 - 0.46 * std::cos((2.0 * pi * n) / (ntaps - 1)));
    const int M = (ntaps - 1) / 2;
    const double fwT0 = 2.0 * pi * cutoff_freq / sampling_freq;
    for (int n = -M; n <= M; n++)
        {
            if (n == 0)
                taps[n + M] = static_cast<float>(fwT0 / pi * w[n + M]);
            else
                taps[n + M] = static_cast<float>(std::sin(n * fwT0) / (n * pi) * w[n + M]);
        }
    // unity (times gain) response at DC
    double fmax = taps[0 + M];
    for (int n = 1; n <= M; n++) fmax += 2 * taps[n + M];
    const double g = gain / fmax;
    for (int i = 0; i < ntaps; i++) taps[i] = static_cast<float>(taps[i] * g);
    return taps;
}

struct Hip_Acq_Resampler_Design
{
    int decimation{1};              //!< 1: "Disabled acquisition resampler because the input sampling frequency is too low"
    double acq_fs_decimated{0.0};   //!< the rate the acquisition block is configured with (Acq_Conf::resampled_fs)
    std::vector<float> taps;        //!< empty when decimation == 1
    uint32_t resampler_latency{0};  //!< (taps.size() - 1) / 2, gnss_flowgraph.cc:1207
};

/*! gnss_flowgraph.cc:1165-1211 for internal_fs_sps = fs and the signal's *_OPT_ACQ_FS_SPS = acq_fs */
inline Hip_Acq_Resampler_Design hip_design_acq_resampler(uint32_t fs, double acq_fs)
{
    Hip_Acq_Resampler_Design d;
    d.acq_fs_decimated = static_cast<double>(fs);
    if (!(acq_fs < fs)) return d;
    const double resampler_ratio = static_cast<double>(fs) / acq_fs;
    int decimation = static_cast<int>(std::floor(resampler_ratio));
    while (fs % decimation > 0) decimation--;
    if (decimation <= 1) return d;
    d.decimation = decimation;
    d.acq_fs_decimated = static_cast<double>(fs) / static_cast<double>(decimation);
    d.taps = hip_firdes_low_pass(1.0, fs, d.acq_fs_decimated / 2.1, d.acq_fs_decimated / 2);
    d.resampler_latency = static_cast<uint32_t>((d.taps.size() - 1) / 2);
    return d;
}

#endif
