// GPU test of the C++ host side (gnss-sdr_amd/host/): drives Hip_Multicorrelator_Real_Codes exactly as
// dll_pll_veml_tracking does (trk.cc:652-675 init, :1030 set_local_code_and_taps, :1236-1243 correlate) and
// Hip_Pcps_Acquisition_Core as pcps_acquisition does, and checks both against the oracle (oracle/gnss_oracle.h).
// Built by __graft_entry__.build(); run by tests/test_host_classes_gpu.py.  Prints "HOST CLASSES OK" on success.
#include "gnss_oracle.h"
#include "hip_multicorrelator_real_codes.h"
#include "hip_multicorrelator_16sc.h"
#include "hip_pcps_acquisition_core.h"
#include "hip_pcps_detectors.h"
#include "hip_acq_resampler.h"
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
        mc.update_local_code(vector_length, 0.25F, code_step);  // mcorr.h:46: compiles and is harmless (the replicas are never materialised)
        EXPECT(mc.last_error().empty(), "update_local_code: %s", mc.last_error().c_str());
        EXPECT(mc.free(), "free");
        // use before init must fail loudly, not compute garbage
        Hip_Multicorrelator_Real_Codes cold;
        EXPECT(!cold.Carrier_wipeoff_multicorrelator_resampler(0.F, 0.F, 0.F, 0.F, 0.1F, 0.F, 100), "uninitialised correlate must fail");
    }
    // ---------------------------------------------------------------- the 16-bit correlator (Cpu_Multicorrelator_16sc's call pattern), bit for bit
    {
        const int n = 25000, n_taps = 3;
        std::mt19937 gen(16);
        std::uniform_int_distribution<int> amp(-60, 60);
        std::vector<std::complex<int16_t>> in(n + 40), code(1023), outs(n_taps);
        for (auto& v : in) v = std::complex<int16_t>(static_cast<int16_t>(amp(gen)), static_cast<int16_t>(amp(gen)));
        std::vector<float> ca(1023);
        oracle_gps_l1_ca_code_gen_float(ca.data(), 5, 0);
        for (int i = 0; i < 1023; i++) code[i] = std::complex<int16_t>(static_cast<int16_t>(ca[i]), 0);
        std::vector<float> shifts = {-0.5F, 0.0F, 0.5F};
        Hip_Multicorrelator_16sc mc;
        EXPECT(mc.init(2 * n, n_taps), "16sc init: %s", mc.last_error().c_str());
        EXPECT(mc.set_local_code_and_taps(1023, code.data(), shifts.data()), "16sc set_local_code_and_taps: %s", mc.last_error().c_str());
        for (int epoch = 0; epoch < 3; epoch++)
            {
                const std::complex<int16_t>* win = in.data() + epoch * 13;
                const float rem_carr = 0.4F + 1.1F * epoch, phase_step = 0.0123F * (epoch - 1), rem_code = 0.37F * epoch, code_step = 0.04092F;
                EXPECT(mc.set_input_output_vectors(outs.data(), win), "16sc set_input_output_vectors");
                EXPECT(mc.Carrier_wipeoff_multicorrelator_resampler(rem_carr, phase_step, rem_code, code_step, n), "16sc correlate: %s", mc.last_error().c_str());
                int16_t want[6];
                oracle_mcorr16(reinterpret_cast<const int16_t*>(code.data()), 1023, shifts.data(), n_taps, reinterpret_cast<const int16_t*>(win), n, rem_carr, phase_step, rem_code,
                    code_step, want);
                for (int t = 0; t < n_taps; t++)
                    EXPECT(outs[t].real() == want[2 * t] && outs[t].imag() == want[2 * t + 1], "16sc epoch %d tap %d: (%d, %d) against the oracle's (%d, %d)", epoch, t,
                        outs[t].real(), outs[t].imag(), want[2 * t], want[2 * t + 1]);
                if (epoch == 0) shifts[2] = 0.25F;  // borrowed taps, changed in place
            }
        mc.update_local_code(n, 0.25F, 0.04F);
        EXPECT(mc.last_error().empty(), "16sc update_local_code: %s", mc.last_error().c_str());
        EXPECT(mc.free(), "16sc free");
        Hip_Multicorrelator_16sc cold;
        outs[0] = std::complex<int16_t>(7, 7);
        cold.set_input_output_vectors(outs.data(), in.data());
        EXPECT(!cold.Carrier_wipeoff_multicorrelator_resampler(0.F, 0.F, 0.F, 0.1F, 100), "16sc: an uninitialised correlate must fail");
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
    // ---------------------------------------------------------------- Tong detector, pcps_tong_acquisition_cc call pattern
    {
        // gps_l1_ca_pcps_tong_acquisition_gsoc2013_test.cc:199-258: PRN 10, 750 Hz, 600 chips, 44 dB-Hz, threshold 0.00108, init 1, max 8
        const double amp = std::sqrt(std::pow(10.0, 4.4) * 2.0 / 4e6);
        auto x = make_signal(12 * 4000, 4e6, 10, 750.0, 1023.0 - 600.0, amp, 2013);
        Hip_Acq_Conf conf;
        conf.fs_in = 4000000;
        conf.doppler_max = 10000;
        conf.doppler_step = 250;
        conf.threshold = 0.00108F;
        conf.SetDerivedParams();
        Hip_Pcps_Tong_Core tong(conf, 1, 8, 9, 0);
        EXPECT(tong.ok(), "tong create: %s", tong.last_error().c_str());
        EXPECT(tong.num_doppler_bins() == 81 && tong.fft_size() == 4000, "tong sizes %u %u", tong.num_doppler_bins(), tong.fft_size());
        std::vector<float> code_iq(2 * 4000);
        oracle_gps_l1_ca_code_gen_complex_sampled(code_iq.data(), 10, 4000000, 0);
        tong.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
        tong.init();
        int st = 1, k = 0;
        float last = 0.0F;
        while (st == 1 && k < 12)
            {
                st = tong.work(4000ULL * (k + 1), x.data() + 4000 * k);
                EXPECT(st >= 1, "tong work: %s", tong.last_error().c_str());
                EXPECT(tong.tong_count() == static_cast<uint32_t>(k + 2), "tong counter %u after dwell %d", tong.tong_count(), k + 1);
                EXPECT(tong.mag() > last && tong.mag() > 0.00108F * (k + 1), "accumulated statistic %g after dwell %d", tong.mag(), k + 1);
                last = tong.mag();
                k++;
            }
        EXPECT(st == 2 && k == 7, "tong ends in state %d after %d dwells", st, k);
        EXPECT(std::abs(600.0 - tong.result().Acq_delay_samples * 1023.0 / 4000.0) < 0.5, "tong delay %f", tong.result().Acq_delay_samples);
        EXPECT(std::abs(tong.result().Acq_doppler_hz - 750.0) < 2.0 / 3e-3, "tong doppler %f", tong.result().Acq_doppler_hz);
        EXPECT(tong.result().Acq_samplestamp_samples == 28000ULL && tong.result().Acq_doppler_step == 250U, "tong synchro fields");
        // noise only: the first miss takes the counter 1 -> 0 (tong.cc:288-294)
        auto noise = make_signal(4000, 4e6, 10, 0.0, 0.0, 0.0, 5);
        Hip_Acq_Conf cn = conf;
        cn.threshold = 0.004F;
        Hip_Pcps_Tong_Core tn(cn, 1, 8, 9, 0);
        tn.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
        tn.init();
        EXPECT(tn.work(4000, noise.data()) == 3 && tn.tong_count() == 0, "tong noise-only: state %d count %u", tn.state(), tn.tong_count());
        EXPECT(std::abs(hip_threshold_compute_doppler(0.1F, 4000, 10000, 250) - 0.0037F) < 2e-4F, "ThresholdComputeDoppler %g", hip_threshold_compute_doppler(0.1F, 4000, 10000, 250));
    }
    // ---------------------------------------------------------------- 8 ms detector logic, galileo_pcps_8ms_acquisition_cc call pattern
    {
        // the class is code-agnostic; two C/A periods stand in for the two E1 primary-code periods (the E1 case runs in
        // tests/test_pcps_detectors_gpu.py): a sign flip between the periods must select code B, none code A
        const double amp = std::sqrt(std::pow(10.0, 4.4) * 2.0 / 4e6);
        Hip_Acq_Conf conf;
        conf.fs_in = 4000000;
        conf.sampled_ms = 2;
        conf.ms_per_code = 1;
        conf.doppler_max = 5000;
        conf.doppler_step = 250;
        conf.max_dwells = 1;
        conf.SetDerivedParams();
        conf.threshold = hip_threshold_compute_doppler(0.01F, 8000, 5000, 250);
        std::vector<float> one(2 * 4000), code_iq(2 * 8000);
        oracle_gps_l1_ca_code_gen_complex_sampled(one.data(), 7, 4000000, 0);
        for (int i = 0; i < 8000; i++) code_iq[i] = one[i], code_iq[8000 + i] = one[i];
        for (int flip = 0; flip < 2; flip++)
            {
                // the block starts exactly on a code period (delay 0) so that the symbol boundary sits mid-block
                auto x = make_signal(8000, 4e6, 7, -1250.0, 0.0, amp, 31 + flip);
                if (flip)
                    {
                        // subtract twice the signal part of the second period: x = noise + s  ->  noise - s
                        std::vector<float> ca(1023);
                        oracle_gps_l1_ca_code_gen_float(ca.data(), 7, 0);
                        const double f_code = 1.023e6 * (1.0 - 1250.0 / 1575.42e6);
                        for (int i = 4000; i < 8000; i++)
                            {
                                const long chip = static_cast<long>(std::floor(i * f_code / 4e6)) % 1023;
                                const double ph = 2.0 * M_PI * -1250.0 / 4e6 * i;
                                x[i] -= 2.0F * std::complex<float>(static_cast<float>(amp * ca[chip] * std::cos(ph)), static_cast<float>(amp * ca[chip] * std::sin(ph)));
                            }
                    }
                Hip_Galileo_Pcps_8ms_Core e8(conf, 0);
                EXPECT(e8.ok() && e8.num_doppler_bins() == 41 && e8.fft_size() == 8000, "8ms create: %s", e8.last_error().c_str());
                e8.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
                e8.init();
                const int st = e8.work(8000, x.data());
                EXPECT(st == 2, "8ms state %d (%s), statistic %g threshold %g", st, e8.last_error().c_str(), e8.test_statistics(), conf.threshold);
                EXPECT(e8.winning_code() == flip, "8ms picked code %d with flip %d", e8.winning_code(), flip);
                EXPECT(e8.result().Acq_delay_samples < 2.0 || e8.result().Acq_delay_samples > 3998.0, "8ms delay %f", e8.result().Acq_delay_samples);
                EXPECT(std::abs(e8.result().Acq_doppler_hz + 1250.0) <= 250.0, "8ms doppler %f", e8.result().Acq_doppler_hz);
                EXPECT(e8.input_power() > 1.9F && e8.input_power() < 2.2F, "8ms input power %g", e8.input_power());
            }
        auto noise = make_signal(8000, 4e6, 7, 0.0, 0.0, 0.0, 77);
        Hip_Galileo_Pcps_8ms_Core e8(conf, 0);
        e8.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
        e8.init();
        EXPECT(e8.work(8000, noise.data()) == 3, "8ms noise-only state %d statistic %g", e8.state(), e8.test_statistics());
    }
    // ---------------------------------------------------------------- CCCWSR detector, pcps_cccwsr_acquisition_cc call pattern
    {
        // the class is code-agnostic: C/A PRN 7 stands in for the data code and PRN 9 for the pilot code (the E1B / E1C case of
        // galileo_e1_pcps_cccwsr_ambiguous_acquisition_gsoc2013_test.cc runs in tests/test_pcps_detectors_gpu.py)
        const double amp = std::sqrt(std::pow(10.0, 4.4) * 2.0 / 4e6);
        Hip_Acq_Conf conf;
        conf.fs_in = 4000000;
        conf.sampled_ms = 2;
        conf.ms_per_code = 1;
        conf.doppler_max = 5000;
        conf.doppler_step = 250;
        conf.max_dwells = 1;
        conf.SetDerivedParams();
        // the threshold is configuration-given for this block: its adapter passes ThresholdComputeBasic
        // (galileo_e1_pcps_cccwsr_ambiguous_acquisition.cc:45), pfa is not used.  The single-code Pfa formula would be wrong here anyway:
        // the combined local codes (data -+ j pilot) carry twice the energy of one code and there are two branches, so the maximum of a
        // noise-only grid sits near 2 ln(2 * 8000 * 41) / 8000 = 0.0034 (0.00323 measured for seed 77), twice the single-code value.
        // The coherent branch of the 44 dB-Hz signal gives about 4 A^2 / 2 = 0.025.
        conf.threshold = 0.008F;
        std::vector<float> one(2 * 4000), data_iq(2 * 8000), pilot_iq(2 * 8000);
        oracle_gps_l1_ca_code_gen_complex_sampled(one.data(), 7, 4000000, 0);
        for (int i = 0; i < 8000; i++) data_iq[i] = one[i], data_iq[8000 + i] = one[i];
        oracle_gps_l1_ca_code_gen_complex_sampled(one.data(), 9, 4000000, 0);
        for (int i = 0; i < 8000; i++) pilot_iq[i] = one[i], pilot_iq[8000 + i] = one[i];
        const auto* cd = reinterpret_cast<const std::complex<float>*>(data_iq.data());
        const auto* cp = reinterpret_cast<const std::complex<float>*>(pilot_iq.data());
        // same seed, amplitude 0: the same noise draws (make_signal draws per sample whatever amp is) -> the bare pilot component
        auto pilot_part = make_signal(8000, 4e6, 9, -1250.0, 0.0, amp, 41);
        const auto pilot_noise = make_signal(8000, 4e6, 9, -1250.0, 0.0, 0.0, 41);
        for (int i = 0; i < 8000; i++) pilot_part[i] -= pilot_noise[i];
        // mode 0/1: pilot in quadrature (+90 deg) with sign +1 / -1: data + j pilot sees (1 - s), data - j pilot sees (1 + s) -> branch 1 / 0
        // mode 2: pilot in phase, sign -1 (the E1 composite): equal expected peaks in both branches, only the published values are checked
        for (int mode = 0; mode < 3; mode++)
            {
                auto x = make_signal(8000, 4e6, 7, -1250.0, 0.0, amp, 51 + mode);
                const std::complex<float> w = mode == 0 ? std::complex<float>(0, 1) : (mode == 1 ? std::complex<float>(0, -1) : std::complex<float>(-1, 0));
                for (int i = 0; i < 8000; i++) x[i] += w * pilot_part[i];
                Hip_Pcps_Cccwsr_Core cw(conf, 0);
                EXPECT(cw.ok() && cw.num_doppler_bins() == 41 && cw.fft_size() == 8000, "cccwsr create: %s", cw.last_error().c_str());
                cw.set_local_code(cd, cp);
                cw.init();
                const int st = cw.work(8000, x.data());
                EXPECT(st == 2, "cccwsr mode %d state %d (%s), statistic %g threshold %g", mode, st, cw.last_error().c_str(), cw.test_statistics(), conf.threshold);
                if (mode < 2) EXPECT(cw.winning_branch() == (mode == 0 ? 1 : 0), "cccwsr mode %d picked branch %d", mode, cw.winning_branch());
                EXPECT(cw.result().Acq_delay_samples < 2.0 || cw.result().Acq_delay_samples > 3998.0, "cccwsr mode %d delay %f", mode, cw.result().Acq_delay_samples);
                EXPECT(std::abs(cw.result().Acq_doppler_hz + 1250.0) <= 250.0, "cccwsr mode %d doppler %f", mode, cw.result().Acq_doppler_hz);
                EXPECT(cw.result().Acq_samplestamp_samples == 8000, "cccwsr sample stamp");
                EXPECT(cw.input_power() > 1.9F && cw.input_power() < 2.3F, "cccwsr input power %g", cw.input_power());
            }
        // noise only: negative after max_dwells; then the running maximum (cccwsr.cc:160: d_mag is cleared in state 0 only) over two dwells
        auto noise = make_signal(8000, 4e6, 7, 0.0, 0.0, 0.0, 77);
        {
            Hip_Pcps_Cccwsr_Core cw(conf, 0);
            cw.set_local_code(cd, cp);
            cw.init();
            EXPECT(cw.work(8000, noise.data()) == 3, "cccwsr noise-only state %d statistic %g", cw.state(), cw.test_statistics());
        }
        {
            Hip_Acq_Conf conf2 = conf;
            conf2.max_dwells = 2;
            conf2.threshold = 1e9F;
            auto x = make_signal(8000, 4e6, 7, -1250.0, 0.0, amp, 53);
            for (int i = 0; i < 8000; i++) x[i] -= pilot_part[i];
            Hip_Pcps_Cccwsr_Core cw(conf2, 0);
            cw.set_local_code(cd, cp);
            cw.init();
            EXPECT(cw.work(8000, x.data()) == 1, "cccwsr dwell 1 of 2 state %d", cw.state());
            const float mag1 = cw.mag();
            const double delay1 = cw.result().Acq_delay_samples;
            EXPECT(cw.work(16000, noise.data()) == 3, "cccwsr dwell 2 of 2 state %d", cw.state());
            EXPECT(cw.mag() == mag1 && cw.result().Acq_delay_samples == delay1 && cw.result().Acq_samplestamp_samples == 8000,
                "cccwsr running maximum: mag %g -> %g, stamp %llu", mag1, cw.mag(), static_cast<unsigned long long>(cw.result().Acq_samplestamp_samples));
            EXPECT(cw.test_statistics() == mag1 / cw.input_power(), "cccwsr statistic uses the last dwell's power");
        }
    }
    // ---------------------------------------------------------------- QuickSync detector, pcps_quicksync_acquisition_cc call pattern
    {
        // gps_l1_ca_pcps_quicksync_acquisition_gsoc2014_test.cc:222-279: fs 8 Msps, 4 ms, PRN 10, 750 Hz, 600 chips, folding factor 4
        const double amp = std::sqrt(std::pow(10.0, 4.4) * 2.0 / 8e6);
        auto x = make_signal(32000, 8e6, 10, 750.0, 1023.0 - 600.0, amp, 2014);
        Hip_Acq_Conf conf;
        conf.fs_in = 8000000;
        conf.doppler_max = 10000;
        conf.doppler_step = 250;
        conf.threshold = 0.7F;
        conf.SetDerivedParams();
        Hip_Pcps_Quicksync_Core qs(conf, 8000, 4, 1, 0);
        EXPECT(qs.ok() && qs.fft_size() == 2000 && qs.num_doppler_bins() == 81 && qs.input_length() == 32000, "quicksync create: %s", qs.last_error().c_str());
        std::vector<float> code_iq(2 * 8000);
        oracle_gps_l1_ca_code_gen_complex_sampled(code_iq.data(), 10, 8000000, 0);
        qs.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
        qs.init();
        const int st = qs.work(32000, x.data());
        EXPECT(st == 2, "quicksync state %d (%s) statistic %g", st, qs.last_error().c_str(), qs.test_statistics());
        EXPECT(std::abs(600.0 - qs.result().Acq_delay_samples * 1023.0 / 8000.0) < 0.5, "quicksync delay %f", qs.result().Acq_delay_samples);
        EXPECT(std::abs(qs.result().Acq_doppler_hz - 750.0) < 2.0 / (3 * 4e-3), "quicksync doppler %f", qs.result().Acq_doppler_hz);
        EXPECT(qs.result().Acq_delay_samples == qs.result().index_time + 2.0 * 2000, "alias: delay %f folded %u", qs.result().Acq_delay_samples, qs.result().index_time);
        float top = 0.0F, second = 0.0F;
        for (float v : qs.corr_output_f())
            {
                if (v > top)
                    {
                        second = top;
                        top = v;
                    }
                else if (v > second) second = v;
            }
        EXPECT(top > 4.0F * second, "alias correlations not separated: %g vs %g", top, second);
        EXPECT(std::abs(hip_threshold_compute_quicksync(0.1F, 8000, 4, 10000, 250) - 0.0071F) < 3e-4F, "ThresholdComputeQuickSync %g", hip_threshold_compute_quicksync(0.1F, 8000, 4, 10000, 250));
        Hip_Pcps_Quicksync_Core bad(conf, 8000, 101, 1, 0);
        EXPECT(!bad.ok(), "folding factor 101 must be refused");
    }
    // ---------------------------------------------------------------- acquisition resampler design, gnss_flowgraph.cc:1165-1211
    {
        const auto d = hip_design_acq_resampler(16000000U, 2e6);
        EXPECT(d.decimation == 8 && d.acq_fs_decimated == 2e6 && d.taps.size() == 39 && d.resampler_latency == 19, "design: D %d fs %g taps %zu latency %u",
            d.decimation, d.acq_fs_decimated, d.taps.size(), d.resampler_latency);
        double sum = 0.0;
        for (float t : d.taps) sum += t;
        EXPECT(std::abs(sum - 1.0) < 1e-6 && d.taps[19] > d.taps[18] && d.taps[0] == d.taps[38], "low-pass taps: sum %g", sum);
        // centre tap of a Hamming-windowed sinc with cutoff fs_dec / 2.1 at fs = 16 Msps (scipy.signal.firwin(39, 2e6/2.1, fs=16e6)[19])
        EXPECT(std::abs(d.taps[19] - 0.118619F) < 1e-5F, "centre tap %g", d.taps[19]);
        const auto d2 = hip_design_acq_resampler(25000000U, 2e6);
        EXPECT(d2.decimation == 10 && d2.acq_fs_decimated == 2.5e6, "25 Msps: D %d", d2.decimation);
        const auto d3 = hip_design_acq_resampler(3000000U, 2e6);
        EXPECT(d3.decimation == 1 && d3.taps.empty() && d3.resampler_latency == 0, "3 Msps: resampler disabled");
    }
    // ---------------------------------------------------------------- fine-Doppler acquisition, pcps_acquisition_fine_doppler_cc call pattern
    {
        const double amp = std::sqrt(std::pow(10.0, 4.7) * 2.0 / 4e6);
        auto x = make_signal(12 * 4000, 4e6, 10, 1580.0, 1023.0 - 600.0, amp, 77);
        Hip_Acq_Conf conf;
        conf.fs_in = 4000000;
        conf.doppler_max = 5000;
        conf.doppler_step = 500;
        conf.max_dwells = 2;
        conf.threshold = 2.5F;
        conf.SetDerivedParams();
        std::vector<float> code_iq(2 * 4000);
        oracle_gps_l1_ca_code_gen_complex_sampled(code_iq.data(), 10, 4000000, 0);
        for (int consistent = 0; consistent < 2; consistent++)
            {
                Hip_Pcps_Fine_Doppler_Core fd(conf, consistent != 0, 0);
                EXPECT(fd.ok() && fd.num_doppler_points() == 20 && fd.fft_size() == 4000, "fine doppler create: %s", fd.last_error().c_str());
                fd.set_local_code(reinterpret_cast<const std::complex<float>*>(code_iq.data()));
                fd.reset_grid();
                EXPECT(fd.compute_and_accumulate_grid(x.data()) == 1, "first dwell");
                EXPECT(fd.compute_and_accumulate_grid(x.data() + 4000) == 2, "second dwell");
                EXPECT(fd.compute_CAF(8000) == 3 && fd.test_statistics() > 5.0F, "CAF: statistic %g", fd.test_statistics());
                EXPECT(std::abs(600.0 - fd.result().Acq_delay_samples * 1023.0 / 4000.0) < 0.5, "fine doppler delay %f", fd.result().Acq_delay_samples);
                EXPECT(fd.result().Acq_doppler_hz == (consistent ? 1500.0 : -3000.0), "grid doppler %f (consistent %d)", fd.result().Acq_doppler_hz, consistent);
                EXPECT(fd.buffer_more(x.data() + 8000, 40000) == 32000 && fd.buffer_full(), "10 ms buffer");
                EXPECT(fd.estimate_Doppler(), "estimate_Doppler: %s", fd.last_error().c_str());
                EXPECT(std::abs(fd.fine_doppler_hz() - 1580.0F) <= 12.5F, "fine doppler %f", fd.fine_doppler_hz());
                EXPECT(fd.result().Acq_doppler_hz == (consistent ? static_cast<double>(fd.fine_doppler_hz()) : -3000.0), "reported doppler %f (consistent %d)",
                    fd.result().Acq_doppler_hz, consistent);
            }
    }
    if (fails == 0) std::printf("HOST CLASSES OK\n");
    return fails == 0 ? 0 : 1;
}
