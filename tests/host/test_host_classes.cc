// GPU test of the C++ host side (gnss-sdr_amd/host/): drives Hip_Multicorrelator_Real_Codes exactly as
// dll_pll_veml_tracking does (trk.cc:652-675 init, :1030 set_local_code_and_taps, :1236-1243 correlate) and
// Hip_Pcps_Acquisition_Core as pcps_acquisition does, and checks both against the oracle (oracle/gnss_oracle.h).
// Built by __graft_entry__.build(); run by tests/test_host_classes_gpu.py.  Prints "HOST CLASSES OK" on success.
#include "gnss_oracle.h"
#include "hip_multicorrelator_real_codes.h"
#include "hip_pcps_acquisition_core.h"
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

namespace
{
int fails = 0;
#define EXPECT(cond, ...)                      \
    do                                         \
        {                                      \
            if (!(cond))                       \
                {                              \
                    std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
                    std::printf(__VA_ARGS__);  \
                    std::printf("\n");         \
                    fails++;                   \
                }                              \
        }                                      \
    while (0)

struct Synchro  // the members of Gnss_Synchro (gnss_synchro.h:46-82) that update_synchro writes
{
    double Acq_delay_samples{0.0};
    double Acq_doppler_hz{0.0};
    uint64_t Acq_samplestamp_samples{0};
    int64_t fs{0};
    uint32_t Acq_doppler_step{0};
};

std::vector<std::complex<float>> make_signal(int n, double fs, int prn, double doppler, double code_phase_chips, double amp, unsigned seed)
{
    std::mt19937 gen(seed);
    std::normal_distribution<float> g(0.0F, 1.0F);
    std::vector<float> code(1023);
    oracle_gps_l1_ca_code_gen_float(code.data(), prn, 0);
    std::vector<std::complex<float>> x(n);
    const double f_code = 1.023e6 * (1.0 + doppler / 1575.42e6);
    for (int i = 0; i < n; i++)
        {
            const long chip = static_cast<long>(std::floor(i * f_code / fs + code_phase_chips)) % 1023;
            const double ph = 2.0 * M_PI * doppler / fs * i;
            x[i] = std::complex<float>(g(gen), g(gen)) + std::complex<float>(static_cast<float>(amp * code[chip] * std::cos(ph)), static_cast<float>(amp * code[chip] * std::sin(ph)));
        }
    return x;
}
}  // namespace

int main()
{
    // ---------------------------------------------------------------- tracking correlator, trk.cc call pattern
    {
        const int vector_length = 25000, n_taps = 3;
        const double fs = 25e6, doppler = 2345.0;
        auto in = make_signal(2 * vector_length, fs, 9, doppler, 0.0, 0.05, 7);
        std::vector<float> ca(1023);
        oracle_gps_l1_ca_code_gen_float(ca.data(), 9, 0);
        std::vector<float> shifts = {-0.5F, 0.0F, 0.5F};
        std::vector<std::complex<float>> outs(n_taps);
        Hip_Multicorrelator_Real_Codes mc;
        EXPECT(mc.init(2 * vector_length, n_taps), "init: %s", mc.last_error().c_str());
        mc.set_high_dynamics_resampler(false);
        EXPECT(mc.set_local_code_and_taps(1023, ca.data(), shifts.data()), "set_local_code_and_taps: %s", mc.last_error().c_str());
        double rem_carr = 0.3, rem_code = 0.0;
        const float phase_step = static_cast<float>(2.0 * M_PI * doppler / fs);
        const float code_step = static_cast<float>(1.023e6 * (1.0 + doppler / 1575.42e6) / fs);
        for (int epoch = 0; epoch < 3; epoch++)
            {
                const std::complex<float>* win = in.data() + epoch * 7;  // odd and even offsets
                EXPECT(mc.set_input_output_vectors(outs.data(), win), "set_input_output_vectors");
                EXPECT(mc.Carrier_wipeoff_multicorrelator_resampler(static_cast<float>(rem_carr), phase_step, 0.0F, static_cast<float>(rem_code), code_step, 0.0F, vector_length),
                    "correlate: %s", mc.last_error().c_str());
                double truth[6], sabs = 0.0;
                oracle_mcorr_f64(ca.data(), 1023, shifts.data(), n_taps, reinterpret_cast<const float*>(win), vector_length, static_cast<float>(rem_carr), phase_step, 0.0F,
                    static_cast<float>(rem_code), code_step, 0.0F, 0, truth, &sabs);
                for (int t = 0; t < n_taps; t++)
                    {
                        const double err = std::hypot(outs[t].real() - truth[2 * t], outs[t].imag() - truth[2 * t + 1]) / sabs;
                        EXPECT(err < 1e-6, "epoch %d tap %d: scale error %.3e", epoch, t, err);
                    }
                if (epoch == 1)
                    {
                        shifts[0] = -0.15F;  // taps are borrowed and mutated in place, trk.cc:2132-2146
                        shifts[2] = 0.15F;
                    }
                rem_carr = std::fmod(rem_carr + phase_step * 7, 2.0 * M_PI);
            }
        EXPECT(mc.free(), "free");
        // use before init must fail loudly, not compute garbage
        Hip_Multicorrelator_Real_Codes cold;
        EXPECT(!cold.Carrier_wipeoff_multicorrelator_resampler(0.F, 0.F, 0.F, 0.F, 0.1F, 0.F, 100), "uninitialised correlate must fail");
    }
    // ---------------------------------------------------------------- acquisition, pcps_acquisition call pattern
    {
        Hip_Acq_Conf conf;
        conf.fs_in = 4000000;
        conf.doppler_max = 10000;
        conf.doppler_step = 250;
        conf.pfa = 0.001F;
        conf.max_dwells = 1;
        conf.SetDerivedParams();
        Hip_Pcps_Acquisition_Core acq(conf, 0);
        EXPECT(acq.ok(), "acq create: %s", acq.last_error().c_str());
        EXPECT(acq.consumed_samples() == 4000 && acq.fft_size() == 4000 && acq.num_doppler_bins() == 80, "sizes %u %u %u", acq.consumed_samples(), acq.fft_size(), acq.num_doppler_bins());
        std::vector<float> code_iq(2 * 4000);
        oracle_gps_l1_ca_code_gen_complex_sampled(code_iq.data(), 10, 4000000, 0);
        acq.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
        const double amp = std::sqrt(std::pow(10.0, 4.7) * 2.0 / 4e6);
        auto x = make_signal(4000, 4e6, 10, 750.0, 1023.0 - 600.0, amp, 2013);
        Hip_Pcps_Acquisition_Core::AcquisitionResult res;
        const auto out = acq.acquisition_core(123456, x.data(), &res);
        EXPECT(out == Hip_Pcps_Acquisition_Core::ACQ_POSITIVE, "outcome %d stat %f thr %f", out, res.test_statistics, acq.get_threshold());
        Synchro syn;
        acq.update_synchro(res, &syn);
        EXPECT(std::fabs(600.0 - syn.Acq_delay_samples * 1023.0 / 4000.0) < 0.5, "delay %f", syn.Acq_delay_samples);
        EXPECT(std::fabs(syn.Acq_doppler_hz - 750.0) < 2.0 / 3e-3, "doppler %f", syn.Acq_doppler_hz);
        EXPECT(syn.Acq_samplestamp_samples == 123456 && syn.fs == 4000000, "stamp");
        // a PRN that is not there: negative after max_dwells
        oracle_gps_l1_ca_code_gen_complex_sampled(code_iq.data(), 11, 4000000, 0);
        acq.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
        EXPECT(acq.acquisition_core(0, x.data(), &res) == Hip_Pcps_Acquisition_Core::ACQ_NEGATIVE, "absent PRN must be rejected (stat %f)", res.test_statistics);
    }
    // ---------------------------------------------------------------- make_two_steps (acq.cc:605-632) and cshort input (acq.cc:653-656)
    {
        Hip_Acq_Conf conf;
        conf.fs_in = 4000000;
        conf.doppler_max = 5000;
        conf.doppler_step = 500;
        conf.pfa = 0.001F;
        conf.pfa2 = 0.001F;
        conf.max_dwells = 2;
        conf.make_2_steps = true;
        conf.num_doppler_bins_step2 = 4;
        conf.doppler_step2 = 125.0F;
        conf.SetDerivedParams();
        Hip_Pcps_Acquisition_Core acq(conf, 0);
        EXPECT(acq.ok(), "two-step acq create: %s", acq.last_error().c_str());
        std::vector<float> code_iq(2 * 4000);
        oracle_gps_l1_ca_code_gen_complex_sampled(code_iq.data(), 17, 4000000, 0);
        acq.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
        const double amp = std::sqrt(std::pow(10.0, 4.8) * 2.0 / 4e6);
        auto x = make_signal(8000, 4e6, 17, 1310.0, 200.0, amp, 99);
        Hip_Pcps_Acquisition_Core::AcquisitionResult res;
        // step one crosses the threshold: no message yet, step two armed (acq.cc:617-624)
        auto out = acq.acquisition_core(4000, x.data(), &res);
        EXPECT(out == Hip_Pcps_Acquisition_Core::ACQ_CONTINUE && acq.step_two() && !res.positive_acq, "step one outcome %d step_two %d", out, acq.step_two());
        EXPECT(res.doppler == 1500 || res.doppler == 1000, "coarse doppler %d", res.doppler);
        const float thr1 = Hip_Pcps_Acquisition_Core::compute_threshold(0.001F, 4000, 20, 2);
        const float thr2 = Hip_Pcps_Acquisition_Core::compute_threshold(0.001F, 4000, 4, 2);
        EXPECT(acq.get_threshold() == thr2 && thr2 < thr1, "step-two threshold %f (step one %f)", acq.get_threshold(), thr1);
        // step two on the next block: positive, Doppler within one fine bin, Acq_doppler_step reported (acq.cc:598-601)
        out = acq.acquisition_core(8000, x.data() + 4000, &res);
        EXPECT(out == Hip_Pcps_Acquisition_Core::ACQ_POSITIVE && res.positive_acq && res.step_two && !acq.step_two(), "step two outcome %d", out);
        EXPECT(std::abs(res.doppler - 1310) <= 125, "fine doppler %d", res.doppler);
        Synchro syn;
        acq.update_synchro(res, &syn);
        EXPECT(syn.Acq_doppler_step == 125 && syn.Acq_samplestamp_samples == 8000, "synchro after step two: step %u stamp %llu", syn.Acq_doppler_step, (unsigned long long)syn.Acq_samplestamp_samples);
        // cshort input gives the same decision as the float path over the converted samples
        Hip_Acq_Conf c16 = conf;
        c16.make_2_steps = false;
        c16.max_dwells = 1;
        c16.cshort = true;
        Hip_Pcps_Acquisition_Core acq16(c16, 0), acqf(c16, 0);
        acq16.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
        acqf.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
        std::vector<std::complex<int16_t>> x16(4000);
        std::vector<std::complex<float>> xf(4000);
        for (int i = 0; i < 4000; i++)
            {
                x16[i] = std::complex<int16_t>(static_cast<int16_t>(std::lrint(x[i].real() * 300.0F)), static_cast<int16_t>(std::lrint(x[i].imag() * 300.0F)));
                xf[i] = std::complex<float>(x16[i].real(), x16[i].imag());
            }
        Hip_Pcps_Acquisition_Core::AcquisitionResult r16, rf;
        const auto o16 = acq16.acquisition_core(1, x16.data(), &r16);
        const auto of = acqf.acquisition_core(1, xf.data(), &rf);
        EXPECT(o16 == of && o16 == Hip_Pcps_Acquisition_Core::ACQ_POSITIVE, "cshort outcome %d vs %d", o16, of);
        EXPECT(r16.index_time == rf.index_time && r16.doppler == rf.doppler && r16.test_statistics == rf.test_statistics, "cshort result differs from the float path");
    }
    if (fails == 0) std::printf("HOST CLASSES OK\n");
    return fails == 0 ? 0 : 1;
}
