// GPU test + timing of the receiver-side batching runtime (gnss-sdr_amd/host/hip_correlator_runtime.{h,cc}).
// Threading model of the reference: one thread per channel, each calling its correlator once per code period
// (cpu_multicorrelator_real_codes_test.cc:137-158 runs the correlator from N std::threads the same way; in the receiver the
// callers are the tracking blocks' GNU Radio threads).  Here a producer thread pushes an 8-bit front-end stream into the
// device ring in 5 ms blocks while 32 channel threads correlate their own windows through Hip_Multicorrelator_Batched; every
// result is checked against the float64 oracle, and the same work is timed through the synchronous drop-in class
// Hip_Multicorrelator_Real_Codes (one launch + synchronisation per call) for comparison.
// Prints "RUNTIME OK" and one "RUNTIME_STATS {json}" line.  Built by __graft_entry__.build(); run by tests/test_runtime_gpu.py.
#include "gnss_oracle.h"
#include "hip_correlator_runtime.h"
#include "hip_multicorrelator_real_codes.h"
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace
{
std::atomic<int> fails{0};
#define EXPECT(cond, ...)                                            \
    do                                                               \
        {                                                            \
            if (!(cond))                                             \
                {                                                    \
                    if (fails.fetch_add(1) < 20)                     \
                        {                                            \
                            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
                            std::printf(__VA_ARGS__);                \
                            std::printf("\n");                       \
                        }                                            \
                }                                                    \
        }                                                            \
    while (0)

struct Chan
{
    int prn;
    double doppler;
    uint64_t offset;
    float rem_carr, phase_step, rem_code, code_step;
    std::vector<float> code;
    std::vector<std::complex<float>> results;  // epochs * 3
    std::vector<std::complex<float>> data_results;  // epochs (pilot mode)
};
}  // namespace

int main(int argc, char** argv)
{
    const int C = argc > 1 ? std::atoi(argv[1]) : 32;
    const int E = argc > 2 ? std::atoi(argv[2]) : 200;
    // "mixed": odd channels run the high-dynamics resampler + rotator (trk.cc:675), so every batch holds two correlator flavours and the
    // runtime has to split it into one launch per flavour
    const bool mixed = argc > 3 && std::string(argv[3]) == "mixed";
    // "pilot": every channel thread drives TWO correlators per epoch like a track_pilot block (trk.cc:1236-1256): the VE/E/P/L/VL-style one and,
    // right after it with the same parameters, a single-tap one with another code; the first has the second as its companion
    // (Hip_Multicorrelator_Batched::set_companion).  "pilot_nocompanion": the same without the companion link, for comparison.
    const bool pilot_plain = argc > 3 && std::string(argv[3]) == "pilot_nocompanion";
    const bool pilot = pilot_plain || (argc > 3 && std::string(argv[3]) == "pilot");
    const int N = 25000, T = 3;
    const double fs = 25e6;
    const uint64_t total = static_cast<uint64_t>(E + 2) * N;
    // ---- an 8-bit front-end stream: noise + 4 C/A signals, quantised (what File_Signal_Source item_type=ibyte delivers)
    std::vector<int8_t> raw(2 * total);
    std::vector<std::complex<float>> xf(total);
    {
        std::mt19937 gen(12345);
        std::normal_distribution<float> g(0.0F, 1.0F);
        std::vector<std::vector<float>> codes(4, std::vector<float>(1023));
        const double dop[4] = {1200.0, -3300.0, 450.0, 4100.0};
        const double cph[4] = {10.5, 300.25, 777.0, 1000.75};
        for (int s = 0; s < 4; s++) oracle_gps_l1_ca_code_gen_float(codes[s].data(), s + 1, 0);
        for (uint64_t i = 0; i < total; i++)
            {
                double re = g(gen), im = g(gen);
                for (int s = 0; s < 4; s++)
                    {
                        const long chip = static_cast<long>(std::floor(i * 1.023e6 * (1.0 + dop[s] / 1575.42e6) / fs + cph[s])) % 1023;
                        const double ph = std::fmod(2.0 * M_PI * dop[s] / fs * static_cast<double>(i), 2.0 * M_PI);
                        re += 0.08 * codes[s][chip] * std::cos(ph);
                        im += 0.08 * codes[s][chip] * std::sin(ph);
                    }
                const int qi = std::max(-127, std::min(127, static_cast<int>(std::lrint(re * 30.0))));
                const int qq = std::max(-127, std::min(127, static_cast<int>(std::lrint(im * 30.0))));
                raw[2 * i] = static_cast<int8_t>(qi);
                raw[2 * i + 1] = static_cast<int8_t>(qq);
                xf[i] = std::complex<float>(static_cast<float>(qi), static_cast<float>(qq));  // ibyte_to_complex: plain cast
            }
    }
    std::vector<Chan> ch(C);
    {
        std::mt19937 gen(777);
        std::uniform_real_distribution<double> u(0.0, 1.0);
        for (int c = 0; c < C; c++)
            {
                ch[c].prn = c % 32 + 1;
                ch[c].doppler = -5000.0 + 10000.0 * u(gen);
                ch[c].offset = static_cast<uint64_t>(u(gen) * N);
                ch[c].rem_carr = static_cast<float>(2.0 * M_PI * u(gen));
                ch[c].phase_step = static_cast<float>(2.0 * M_PI * ch[c].doppler / fs);
                ch[c].rem_code = static_cast<float>(u(gen));
                ch[c].code_step = static_cast<float>(1.023e6 * (1.0 + ch[c].doppler / 1575.42e6) / fs);
                ch[c].code.resize(1023);
                oracle_gps_l1_ca_code_gen_float(ch[c].code.data(), ch[c].prn, 0);
                ch[c].results.assign(static_cast<size_t>(E) * T, {0.0F, 0.0F});
                ch[c].data_results.assign(static_cast<size_t>(E), {0.0F, 0.0F});
            }
    }
    const float shifts_init[3] = {-0.5F, 0.0F, 0.5F};

    // ---------------------------------------------------------------- batched runtime
    double batched_s = 0.0;
    Hip_Correlator_Runtime::Stats st;
    {
        const uint64_t block = 5 * N;  // 5 ms per push
        Hip_Sample_Ring ring(0, 40ull * N, 2 * N);
        EXPECT(ring.ok(), "ring: %s", ring.last_error().c_str());
        Hip_Correlator_Runtime rt(&ring, pilot ? 2 * C : C, 1023, std::chrono::microseconds(300));
        EXPECT(rt.ok(), "runtime: %s", rt.last_error().c_str());
        if (fails.load()) return 1;
        std::vector<std::atomic<uint64_t>> consumed(C);
        for (auto& a : consumed) a.store(0);
        std::vector<std::unique_ptr<Hip_Multicorrelator_Batched>> mc(C);
        std::vector<std::vector<float>> shifts(C, std::vector<float>(shifts_init, shifts_init + 3));
        for (int c = 0; c < C; c++)
            {
                mc[c] = std::make_unique<Hip_Multicorrelator_Batched>(&rt);
                EXPECT(mc[c]->init(2 * N, T), "init: %s", mc[c]->last_error().c_str());
                mc[c]->set_high_dynamics_resampler(mixed && (c & 1));
                EXPECT(mc[c]->set_local_code_and_taps(1023, ch[c].code.data(), shifts[c].data()), "set_local_code_and_taps: %s", mc[c]->last_error().c_str());
            }
        std::vector<std::unique_ptr<Hip_Multicorrelator_Batched>> md(pilot ? C : 0);
        std::vector<std::vector<float>> data_code(pilot ? C : 0, std::vector<float>(1023));
        std::vector<std::vector<float>> data_shift(pilot ? C : 0, std::vector<float>(1, 0.0F));
        for (int c = 0; pilot && c < C; c++)
            {
                oracle_gps_l1_ca_code_gen_float(data_code[c].data(), (c + 7) % 32 + 1, 0);
                md[c] = std::make_unique<Hip_Multicorrelator_Batched>(&rt);
                EXPECT(md[c]->init(2 * N, 1), "data init: %s", md[c]->last_error().c_str());
                md[c]->set_high_dynamics_resampler(false);
                EXPECT(md[c]->set_local_code_and_taps(1023, data_code[c].data(), data_shift[c].data()), "data code: %s", md[c]->last_error().c_str());
                mc[c]->set_high_dynamics_resampler(false);
                if (!pilot_plain) mc[c]->set_companion(md[c].get());
            }
        const auto t0 = std::chrono::steady_clock::now();
        std::thread producer([&] {
            uint64_t pushed = 0;
            while (pushed < total)
                {
                    // flow control: never overwrite samples a channel has not consumed yet (ring keeps 40 ms)
                    uint64_t slowest = UINT64_MAX;
                    for (auto& a : consumed) slowest = std::min<uint64_t>(slowest, a.load(std::memory_order_acquire));
                    const uint64_t n = std::min<uint64_t>(block, total - pushed);
                    if (pushed + n > slowest + 38ull * N)
                        {
                            std::this_thread::yield();
                            continue;
                        }
                    const uint64_t first = ring.push_ibyte(raw.data() + 2 * pushed, n);
                    EXPECT(first == pushed, "push returned %llu, expected %llu (%s)", (unsigned long long)first, (unsigned long long)pushed, ring.last_error().c_str());
                    if (first != pushed) return;
                    pushed += n;
                }
        });
        std::vector<std::thread> workers;
        for (int c = 0; c < C; c++)
            workers.emplace_back([&, c] {
                std::complex<float> outs[T];
                std::complex<float> dout[1] = {{0.0F, 0.0F}};
                mc[c]->set_input_output_vectors(outs, nullptr);
                if (pilot) md[c]->set_input_output_vectors(dout, nullptr);  // once, as start_tracking does (trk.cc:668-669)
                for (int e = 0; e < E; e++)
                    {
                        const uint64_t w0 = ch[c].offset + static_cast<uint64_t>(e) * N;
                        if (!ring.wait_for(w0 + N, std::chrono::milliseconds(20000)))
                            {
                                EXPECT(false, "channel %d epoch %d: samples never arrived", c, e);
                                break;
                            }
                        mc[c]->set_input_sample_index(w0);
                        if (pilot)
                            {
                                md[c]->set_input_sample_index(w0);
                                dout[0] = {-1.0F, -1.0F};
                            }
                        const bool hd = mixed && (c & 1);
                        const bool ok = mc[c]->Carrier_wipeoff_multicorrelator_resampler(ch[c].rem_carr, ch[c].phase_step, hd ? 2.0e-9F : 0.0F, ch[c].rem_code, ch[c].code_step,
                            hd ? 1.0e-9F : 0.0F, N);
                        EXPECT(ok, "channel %d epoch %d: %s", c, e, mc[c]->last_error().c_str());
                        if (!ok) break;
                        for (int t = 0; t < T; t++) ch[c].results[static_cast<size_t>(e) * T + t] = outs[t];
                        if (pilot)
                            {
                                // trk.cc:1246-1256: the data correlator, same window, same parameters, right after the pilot's
                                const bool okd = md[c]->Carrier_wipeoff_multicorrelator_resampler(ch[c].rem_carr, ch[c].phase_step, 0.0F, ch[c].rem_code, ch[c].code_step, 0.0F, N);
                                EXPECT(okd, "data correlator channel %d epoch %d: %s", c, e, md[c]->last_error().c_str());
                                ch[c].data_results[static_cast<size_t>(e)] = dout[0];
                            }
                        consumed[c].store(w0 + N, std::memory_order_release);
                    }
                consumed[c].store(UINT64_MAX, std::memory_order_release);
                if (pilot) md[c]->free();
                mc[c]->free();  // leaves the rendezvous: the remaining channels stop waiting for this one
            });
        for (auto& w : workers) w.join();
        producer.join();
        batched_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        st = rt.stats();
    }
    // ---- every result against the float64 oracle on the converted samples
    double worst = 0.0;
    for (int c = 0; c < C; c++)
        for (int e = 0; e < E; e += (e < 4 ? 1 : 7))
            {
                const uint64_t w0 = ch[c].offset + static_cast<uint64_t>(e) * N;
                double truth[2 * T], sabs = 0.0;
                const bool hd = mixed && (c & 1);
                oracle_mcorr_f64(ch[c].code.data(), 1023, shifts_init, T, reinterpret_cast<const float*>(xf.data() + w0), N, ch[c].rem_carr, ch[c].phase_step, hd ? 2.0e-9F : 0.0F,
                    ch[c].rem_code, ch[c].code_step, hd ? 1.0e-9F : 0.0F, hd ? 1 : 0, truth, &sabs);
                for (int t = 0; t < T; t++)
                    {
                        const auto& r = ch[c].results[static_cast<size_t>(e) * T + t];
                        const double err = std::hypot(r.real() - truth[2 * t], r.imag() - truth[2 * t + 1]) / sabs;
                        worst = std::max(worst, err);
                        EXPECT(err < 1e-6, "channel %d epoch %d tap %d: scale error %.3e", c, e, t, err);
                    }
            }
    if (pilot)
        {
            const float zero_shift[1] = {0.0F};
            for (int c = 0; c < C; c++)
                for (int e = 0; e < E; e += (e < 4 ? 1 : 7))
                    {
                        const uint64_t w0 = ch[c].offset + static_cast<uint64_t>(e) * N;
                        std::vector<float> dc(1023);
                        oracle_gps_l1_ca_code_gen_float(dc.data(), (c + 7) % 32 + 1, 0);
                        double truth[2], sabs = 0.0;
                        oracle_mcorr_f64(dc.data(), 1023, zero_shift, 1, reinterpret_cast<const float*>(xf.data() + w0), N, ch[c].rem_carr, ch[c].phase_step, 0.0F, ch[c].rem_code,
                            ch[c].code_step, 0.0F, 0, truth, &sabs);
                        const auto& r = ch[c].data_results[static_cast<size_t>(e)];
                        const double err = std::hypot(r.real() - truth[0], r.imag() - truth[1]) / sabs;
                        worst = std::max(worst, err);
                        EXPECT(err < 1e-6, "data correlator channel %d epoch %d: scale error %.3e (got %g %g)", c, e, err, r.real(), r.imag());
                    }
            EXPECT(st.jobs == 2ull * C * E, "runtime served %llu jobs, expected %d", (unsigned long long)st.jobs, 2 * C * E);
            if (!pilot_plain) EXPECT(st.batches <= static_cast<uint64_t>(E) * 3 / 2 + 8, "companion mode: %llu rendezvous for %d epochs", (unsigned long long)st.batches, E);
        }
    else
        EXPECT(st.jobs == static_cast<uint64_t>(C) * E, "runtime served %llu jobs, expected %d", (unsigned long long)st.jobs, C * E);

    // ---------------------------------------------------------------- the synchronous drop-in class, same work, same threads
    double dropin_s = 0.0;
    if (!mixed && !pilot)
    {
        const int E2 = std::min(E, 50);
        std::vector<std::unique_ptr<Hip_Multicorrelator_Real_Codes>> mc(C);
        std::vector<std::vector<float>> shifts(C, std::vector<float>(shifts_init, shifts_init + 3));
        for (int c = 0; c < C; c++)
            {
                mc[c] = std::make_unique<Hip_Multicorrelator_Real_Codes>(0);
                EXPECT(mc[c]->init(2 * N, T), "drop-in init: %s", mc[c]->last_error().c_str());
                mc[c]->set_high_dynamics_resampler(false);
                EXPECT(mc[c]->set_local_code_and_taps(1023, ch[c].code.data(), shifts[c].data()), "drop-in code");
            }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> workers;
        for (int c = 0; c < C; c++)
            workers.emplace_back([&, c] {
                std::complex<float> outs[T];
                for (int e = 0; e < E2; e++)
                    {
                        const uint64_t w0 = ch[c].offset + static_cast<uint64_t>(e) * N;
                        mc[c]->set_input_output_vectors(outs, xf.data() + w0);
                        const bool ok = mc[c]->Carrier_wipeoff_multicorrelator_resampler(ch[c].rem_carr, ch[c].phase_step, 0.0F, ch[c].rem_code, ch[c].code_step, 0.0F, N);
                        EXPECT(ok, "drop-in channel %d epoch %d: %s", c, e, mc[c]->last_error().c_str());
                        if (!ok) break;
                        if (e < 3)
                            for (int t = 0; t < T; t++)
                                {
                                    const auto& r = ch[c].results[static_cast<size_t>(e) * T + t];
                                    const double d = std::abs(outs[t] - r) / (std::abs(r) + 1e-3);
                                    EXPECT(d < 2e-4, "drop-in vs batched channel %d epoch %d tap %d: %.3e", c, e, t, d);
                                }
                    }
            });
        for (auto& w : workers) w.join();
        dropin_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * (static_cast<double>(E) / E2);
    }
    std::printf("RUNTIME_STATS {\"channels\": %d, \"epochs\": %d, \"samples_per_epoch\": %d, \"batched_channel_epochs_per_s\": %.1f, "
                "\"dropin_channel_epochs_per_s\": %.1f, \"batches\": %llu, \"avg_batch\": %.2f, \"largest_batch\": %u, \"timeouts\": %llu, "
                "\"real_time_factor_batched\": %.2f, \"worst_scale_error\": %.3e}\n",
        C, E, N, C * static_cast<double>(E) / batched_s, dropin_s > 0.0 ? C * static_cast<double>(E) / dropin_s : 0.0, (unsigned long long)st.batches,
        st.batches ? static_cast<double>(st.jobs) / st.batches : 0.0, st.largest_batch, (unsigned long long)st.timeouts, E * 1e-3 / batched_s, worst);
    if (fails.load() == 0) std::printf("RUNTIME OK\n");
    return fails.load() == 0 ? 0 : 1;
}
