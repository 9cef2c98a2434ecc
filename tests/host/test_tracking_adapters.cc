// End-to-end test of the TrackingInterface adapters (gnss-sdr_amd/host/gnss_sdr_adapters/dll_pll_tracking_hip.{h,cc},
// dll_pll_veml_tracking_hip.{h,cc}, dll_pll_conf_hip.{h,cc}) and of Hip_Tracking_Runtime, compiled against the reference's OWN headers
// (tracking_interface.h, dll_pll_conf.h, gnss_synchro.h, in_memory_configuration.h, the signal constant headers, the replica
// generators) and tests/host/mock_gnuradio/ for the GNU Radio runtime.
//
// The CHECKER is the reference itself: oracle/_ref/libgnsssdr_ref_trk.so (the reference's adapters + dll_pll_veml_tracking.cc + libs
// compiled from /root/reference, C driver reftrk_* in oracle/ref_trk_api.cc).  Both chains are built from the same configuration
// properties the way GNSSBlockFactory builds them (constructor(configuration, role, in_streams, out_streams)), wired the way Channel does
// (set_channel, set_gnss_synchro, start_tracking after an acquisition has filled Gnss_Synchro), and driven through general_work over the
// same synthetic IF stream, call for call.
//
//   test_tracking_adapters conf    CPU only: Dll_Pll_Conf -> gsh_trk_conf (hip_fill_trk_conf) equals what the reference block's constructor
//                                  derives, field by field, for every supported signal; replicas equal the block's
//   test_tracking_adapters         on the GPU box: the above + trajectories (prints "TRACKING ADAPTERS OK")
#include "Beidou_B1I.h"
#include "GLONASS_L1_L2_CA.h"
#include "GPS_L1_CA.h"
#include "GPS_L2C.h"
#include "Galileo_E1.h"
#include "Galileo_E5b.h"
#include "qzss.h"
#include "dll_pll_conf_hip.h"
#include "dll_pll_tracking_hip.h"
#include "galileo_e1_signal_replica.h"
#include "gnss_synchro.h"
#include "gps_sdr_signal_replica.h"
#include "in_memory_configuration.h"
#include <algorithm>
#include <any>
#include <array>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <fstream>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

// ---- the reference chain's C driver (oracle/ref_trk_api.cc)
extern "C" {
struct reftrk_output
{
    double fs, prompt_i, prompt_q, cn0_db_hz, carrier_doppler_hz, carrier_phase_rads, code_phase_samples;
    uint64_t tracking_sample_counter;
    int32_t flag_valid_symbol_output, correlation_length_ms, flag_pll_180_deg_phase_locked, prn;
    int32_t state, current_prn_length_samples, n_correlator_taps, cn0_estimation_counter, carrier_lock_fail_counter, code_lock_fail_counter;
    double code_freq_chips, rem_code_phase_samples, rem_code_phase_chips, acc_carrier_phase_rad, carrier_lock_test, carr_phase_error_hz,
        carr_freq_error_hz, carr_error_filt_hz, code_error_chips, code_error_filt_chips, carrier_phase_step_rad, code_phase_step_chips,
        carrier_phase_rate_step_rad, code_phase_rate_step_chips, current_correlation_time_s;
    float rem_carr_phase_rad;
    float corr[10];
    float prompt_data[2];
    float accu[10];
    float p_data_accu[2];
    int32_t n_events;
    int32_t events[16];
    uint64_t tow_at_current_symbol_ms;
};
struct reftrk_conf_out
{
    double fs_in, carrier_lock_th, signal_carrier_freq, code_period, code_chip_rate, bs_dominance_ratio;
    float pll_bw_hz, dll_bw_hz, fll_bw_hz, pll_bw_narrow_hz, dll_bw_narrow_hz, early_late_space_chips, very_early_late_space_chips,
        early_late_space_narrow_chips, very_early_late_space_narrow_chips, slope, spc, y_intercept, cn0_smoother_alpha,
        carrier_lock_test_smoother_alpha, bs_min_prompt_mag;
    uint32_t pull_in_time_s, bit_synchronization_time_limit_s, vector_length, smoother_length;
    int32_t pll_filter_order, dll_filter_order, fll_filter_order, extend_correlation_symbols, cn0_samples, cn0_smoother_samples,
        carrier_lock_test_smoother_samples, cn0_min, max_code_lock_fail, max_carrier_lock_fail, bs_stable_best_required, bs_min_events_for_lock;
    int32_t enable_fll_pull_in, enable_fll_steady_state, track_pilot, carrier_aiding, high_dyn, bs_use_phase_dot_detector;
    int32_t code_length_chips, code_samples_per_chip, symbols_per_bit, secondary, veml, cloop, use_histogram_bit_sync, interchange_iq,
        secondary_code_length, data_secondary_code_length, correlation_length_ms, n_correlator_taps, enable_doppler_correction;
    char secondary_code[256], data_secondary_code[256];
    char system, signal[3];
};
void* reftrk_create(const char* implementation, const char* role, const char* const* keys, const char* const* values, int n_props);
void* reftrk_create_block(char system, const char* signal, uint32_t vector_length, const char* role, const char* const* keys, const char* const* values, int n_props);
void reftrk_get_live(void* h, int32_t* extend_correlation_symbols, double* cfo_frequency_hz);
void reftrk_destroy(void* h);
void reftrk_set_acquisition(void* h, char system, const char* signal, uint32_t prn, double acq_delay_samples, double acq_doppler_hz, uint64_t acq_samplestamp_samples);
void reftrk_start_tracking(void* h);
int reftrk_general_work(void* h, const float* iq, int n_items, int* consumed, reftrk_output* out);
void reftrk_get_conf(void* h, reftrk_conf_out* c);
int reftrk_get_codes(void* h, float* tracking_code, float* data_code, int capacity);
void reftrk_set_channel(void* h, uint32_t channel);
void reftrk_stop_tracking(void* h);
void reftrk_deliver_tow(void* h, const char* signal, int32_t channel, uint32_t tow, uint64_t sample_stamp, int32_t wn, uint32_t prn);
void reftrk_add_input_timetag(void* h, uint64_t offset, double rx_time, int week, int tow_ms, double tow_ms_fraction);
int reftrk_output_timetags(void* h, uint64_t* offsets, int* week, int* tow_ms, double* tow_ms_fraction, double* rx_time, int capacity);
// present only in the *_fake build (tests/host/fake_gsh_engine.cc): the CPU stand-in for the device half of the C ABI wants the whole stream up front
void fake_gsh_set_reference_stream(const float* iq, uint64_t n) __attribute__((weak));
uint64_t fake_gsh_push_mismatches(void) __attribute__((weak));
int fake_gsh_concurrent_handle_entries(void) __attribute__((weak));
}

namespace
{
int fails = 0;
#define EXPECT(cond, ...)                                            \
    do                                                               \
        {                                                            \
            if (!(cond))                                             \
                {                                                    \
                    std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
                    std::printf(__VA_ARGS__);                        \
                    std::printf("\n");                               \
                    fails++;                                         \
                }                                                    \
        }                                                            \
    while (0)

typedef std::map<std::string, std::string> Props;

std::shared_ptr<InMemoryConfiguration> make_config(const Props& p)
{
    auto c = std::make_shared<InMemoryConfiguration>();
    for (const auto& kv : p) c->set_property(kv.first, kv.second);
    return c;
}

void* make_ref(const std::string& impl, const std::string& role, const Props& p)
{
    std::vector<const char*> k, v;
    for (const auto& kv : p)
        {
            k.push_back(kv.first.c_str());
            v.push_back(kv.second.c_str());
        }
    return reftrk_create(impl.c_str(), role.c_str(), k.data(), v.data(), static_cast<int>(k.size()));
}

// ---- Dll_Pll_Conf exactly as the reference adapters leave it, without building a GPU block: the HIP adapters' own constructors do the
// per-signal part; here the same through a tiny subclass that never creates the device block
struct ConfProbe
{
    Dll_Pll_Conf p;
};

void compare_conf(const char* name, const gsh_trk_conf& c, const Hip_Trk_Signal& sig, const reftrk_conf_out& r)
{
#define SAME(field, ref) EXPECT(static_cast<double>(c.field) == static_cast<double>(ref), "%s: %s %g vs reference %g", name, #field, static_cast<double>(c.field), static_cast<double>(ref))
    SAME(fs_in, r.fs_in);
    SAME(code_chip_rate, r.code_chip_rate);
    SAME(signal_carrier_freq, r.signal_carrier_freq);
    SAME(code_length_chips, r.code_length_chips);
    SAME(code_samples_per_chip, r.code_samples_per_chip);
    SAME(vector_length, r.vector_length);
    SAME(veml, r.veml);
    SAME(track_pilot, r.track_pilot);
    SAME(early_late_space_chips, r.early_late_space_chips);
    SAME(very_early_late_space_chips, r.very_early_late_space_chips);
    SAME(pll_bw_hz, r.pll_bw_hz);
    SAME(dll_bw_hz, r.dll_bw_hz);
    SAME(fll_bw_hz, r.fll_bw_hz);
    SAME(pll_filter_order, r.pll_filter_order);
    SAME(dll_filter_order, r.dll_filter_order);
    SAME(enable_fll_pull_in, r.enable_fll_pull_in);
    SAME(enable_fll_steady_state, r.enable_fll_steady_state);
    SAME(carrier_aiding, r.carrier_aiding);
    SAME(cloop, r.cloop);
    SAME(pull_in_time_s, r.pull_in_time_s);
    SAME(spc, r.spc);
    SAME(slope, r.slope);
    SAME(y_intercept, r.y_intercept);
    SAME(cn0_samples, r.cn0_samples);
    SAME(cn0_min, r.cn0_min);
    SAME(max_code_lock_fail, r.max_code_lock_fail);
    SAME(max_carrier_lock_fail, r.max_carrier_lock_fail);
    SAME(cn0_smoother_samples, r.cn0_smoother_samples);
    SAME(carrier_lock_test_smoother_samples, r.carrier_lock_test_smoother_samples);
    SAME(cn0_smoother_alpha, r.cn0_smoother_alpha);
    SAME(carrier_lock_test_smoother_alpha, r.carrier_lock_test_smoother_alpha);
    SAME(carrier_lock_th, r.carrier_lock_th);
    SAME(symbols_per_bit, r.symbols_per_bit);
    SAME(has_secondary, r.secondary);
    SAME(secondary_code_length, r.secondary_code_length);
    SAME(data_secondary_code_length, r.data_secondary_code_length);
    SAME(extend_correlation_symbols, r.extend_correlation_symbols);
    SAME(pll_bw_narrow_hz, r.pll_bw_narrow_hz);
    SAME(dll_bw_narrow_hz, r.dll_bw_narrow_hz);
    SAME(early_late_space_narrow_chips, r.early_late_space_narrow_chips);
    SAME(very_early_late_space_narrow_chips, r.very_early_late_space_narrow_chips);
    SAME(bs_min_events_for_lock, r.bs_min_events_for_lock);
    SAME(bs_stable_best_required, r.bs_stable_best_required);
    SAME(bs_use_phase_dot_detector, r.bs_use_phase_dot_detector);
    SAME(bs_min_prompt_mag, r.bs_min_prompt_mag);
    SAME(bs_dominance_ratio, r.bs_dominance_ratio);
    SAME(high_dyn, r.high_dyn);
    SAME(smoother_length, r.smoother_length);
    SAME(bit_synchronization_time_limit_s, r.bit_synchronization_time_limit_s);
    SAME(enable_doppler_correction, r.enable_doppler_correction);
#undef SAME
    EXPECT(c.enable_bit_sync_time_limit == 1, "%s: the state-2 fail-safe is not switched on", name);
    EXPECT(sig.correlation_length_ms == r.correlation_length_ms, "%s: correlation_length_ms %d vs %d", name, sig.correlation_length_ms, r.correlation_length_ms);
    EXPECT(sig.interchange_iq == (r.interchange_iq != 0), "%s: interchange_iq", name);
    if (!sig.per_prn_secondary)
        EXPECT(std::string(reinterpret_cast<const char*>(c.secondary_code), c.secondary_code_length) == std::string(r.secondary_code), "%s: secondary code", name);
    EXPECT(std::string(reinterpret_cast<const char*>(c.data_secondary_code), c.data_secondary_code_length) == std::string(r.data_secondary_code),
        "%s: data secondary code", name);
    // (the histogram bit synchroniser is configured at the end of start_tracking, trk.cc:1077: compared there)
}

// the Dll_Pll_Conf the HIP adapters end up with, without needing a GPU: same code path as DllPllTrackingHip's constructors up to
// create_tracking_block() -- obtained from a real adapter object when a GPU is present, re-derived here otherwise
struct SignalCase
{
    const char* name;
    const char* ref_impl;
    char system;
    const char* signal;
    double chip_rate, code_length;
    Props props;
    uint32_t prn{7};
    bool e6_adapter{false};  // the reference's Galileo E6 adapter as written: signal tag "5X" (galileo_e6_dll_pll_tracking.cc:62-64)
    bool block_only{false};  // no reference adapter selects this branch of the block: the block is built directly (reftrk_create_block)
};

Dll_Pll_Conf adapter_conf(const SignalCase& sc, const std::string& role)
{
    // what DllPllTrackingHip(configuration, role, ...) + the signal constructor compute (dll_pll_tracking_hip.cc); kept in step by the GPU
    // run below, which compares this very struct with adapter.config_params()
    auto cfg = make_config(sc.props);
    Dll_Pll_Conf p;
    p.SetFromConfiguration(cfg.get(), role);
    p.system = sc.system;
    std::memcpy(p.signal, sc.signal, 3);
    p.vector_length = static_cast<uint32_t>(static_cast<int>(std::round(p.fs_in / (sc.chip_rate / sc.code_length))));
    if (p.extend_correlation_symbols < 1) p.extend_correlation_symbols = 1;
    if (std::string(sc.signal) == "1C")
        {
            p.extend_correlation_symbols = std::min(p.extend_correlation_symbols, 20);
            p.track_pilot = false;
        }
    if (std::string(sc.signal) == "1B" && !p.track_pilot) p.extend_correlation_symbols = 1;
    if (std::string(sc.signal) == "L5" && !p.track_pilot) p.extend_correlation_symbols = std::min(p.extend_correlation_symbols, 10);
    if (std::string(sc.signal) == "5X" && !p.track_pilot) p.extend_correlation_symbols = std::min(p.extend_correlation_symbols, sc.e6_adapter ? 1 : 20);
    const std::string sg(sc.signal);
    if (sg == "2S")
        {
            p.extend_correlation_symbols = 1;
            p.track_pilot = false;
        }
    if (sg == "7X" && !p.track_pilot) p.extend_correlation_symbols = std::min(p.extend_correlation_symbols, 4);
    if (sg == "E6" && !p.track_pilot) p.extend_correlation_symbols = 1;
    if (sg == "B1" || sg == "B3") p.extend_correlation_symbols = std::min(p.extend_correlation_symbols, 20);
    if (sg == "B1") p.track_pilot = false;
    if (sg == "1G" || sg == "2G")
        {
            p.extend_correlation_symbols = std::min(p.extend_correlation_symbols, 10);
            p.track_pilot = false;
        }
    if (sg == "J1")
        {
            p.extend_correlation_symbols = std::min(p.extend_correlation_symbols, 20);
            p.track_pilot = false;
        }
    if (sg == "J5" && !p.track_pilot) p.extend_correlation_symbols = std::min(p.extend_correlation_symbols, 10);
    return p;
}

std::vector<SignalCase> signal_cases()
{
    const std::string R = "Tracking";
    auto base = [&](long fs) { return Props{{"GNSS-SDR.internal_fs_sps", std::to_string(fs)}, {R + ".hip_device", "0"}, {R + ".hip_register_input_buffer", "false"}}; };
    std::vector<SignalCase> v;
    {
        Props p = base(4000000);
        p[R + ".pll_bw_hz"] = "35.0";
        p[R + ".dll_bw_hz"] = "2.0";
        p[R + ".early_late_space_chips"] = "0.5";
        v.push_back({"GPS L1 C/A", "GPS_L1_CA_DLL_PLL_Tracking", 'G', "1C", 1.023e6, 1023.0, p});
        p[R + ".extend_correlation_symbols"] = "10";
        p[R + ".enable_fll_pull_in"] = "true";
        p[R + ".fll_bw_hz"] = "10.0";
        p[R + ".pll_filter_order"] = "2";
        p[R + ".dll_filter_order"] = "1";
        p[R + ".carrier_aiding"] = "false";
        v.push_back({"GPS L1 C/A (extended, FLL pull-in, orders 2/1)", "GPS_L1_CA_DLL_PLL_Tracking", 'G', "1C", 1.023e6, 1023.0, p});
    }
    {
        Props p = base(4000000);
        p[R + ".track_pilot"] = "true";
        p[R + ".early_late_space_chips"] = "0.15";
        p[R + ".very_early_late_space_chips"] = "0.6";
        p[R + ".pll_bw_hz"] = "15.0";
        p[R + ".dll_bw_hz"] = "0.75";
        v.push_back({"Galileo E1 pilot", "Galileo_E1_DLL_PLL_VEML_Tracking", 'E', "1B", 1.023e6, 4092.0, p});
        p[R + ".track_pilot"] = "false";
        p[R + ".early_late_space_chips"] = "0.5";
        v.push_back({"Galileo E1 data", "Galileo_E1_DLL_PLL_VEML_Tracking", 'E', "1B", 1.023e6, 4092.0, p});
    }
    {
        Props p = base(25000000);
        p[R + ".track_pilot"] = "true";
        v.push_back({"GPS L5 pilot", "GPS_L5_DLL_PLL_Tracking", 'G', "L5", 10.23e6, 10230.0, p});
        p[R + ".track_pilot"] = "false";
        p[R + ".high_dyn"] = "true";
        p[R + ".smoother_length"] = "12";
        v.push_back({"GPS L5 data, high dynamics", "GPS_L5_DLL_PLL_Tracking", 'G', "L5", 10.23e6, 10230.0, p});
    }
    // ---- the rest of the dll_pll_veml_tracking family (round 2)
    {
        Props p = base(4000000);
        p[R + ".extend_correlation_symbols"] = "3";  // not allowed on L2C: the adapter sets it back to 1
        v.push_back({"GPS L2C(M)", "GPS_L2_M_DLL_PLL_Tracking", 'G', "2S", 511.5e3, 10230.0, p});
    }
    {
        Props p = base(25000000);
        p[R + ".track_pilot"] = "true";
        p[R + ".extend_correlation_symbols"] = "4";
        v.push_back({"Galileo E5a pilot", "Galileo_E5a_DLL_PLL_Tracking", 'E', "5X", 10.23e6, 10230.0, p});
        v.push_back({"Galileo E5b pilot", "Galileo_E5b_DLL_PLL_Tracking", 'E', "7X", 10.23e6, 10230.0, p});
        p[R + ".track_pilot"] = "false";
        p[R + ".extend_correlation_symbols"] = "8";
        v.push_back({"Galileo E5b data", "Galileo_E5b_DLL_PLL_Tracking", 'E', "7X", 10.23e6, 10230.0, p});
        SignalCase e6{"Galileo E6 adapter as the reference has it (tag 5X)", "Galileo_E6_DLL_PLL_Tracking", 'E', "5X", 5.115e6, 5115.0, p};
        e6.e6_adapter = true;
        v.push_back(e6);
        SignalCase e6b{"Galileo E6 data, the block's own E6 branch", "", 'E', "E6", 5.115e6, 5115.0, p};
        e6b.block_only = true;
        v.push_back(e6b);
        p[R + ".track_pilot"] = "true";
        SignalCase e6c{"Galileo E6 pilot, the block's own E6 branch", "", 'E', "E6", 5.115e6, 5115.0, p};
        e6c.block_only = true;
        v.push_back(e6c);
    }
    {
        Props p = base(8000000);
        p[R + ".extend_correlation_symbols"] = "5";
        v.push_back({"BeiDou B1I MEO", "BEIDOU_B1I_DLL_PLL_Tracking", 'C', "B1", 2.046e6, 2046.0, p});
        SignalCase geo{"BeiDou B1I GEO (PRN 3)", "BEIDOU_B1I_DLL_PLL_Tracking", 'C', "B1", 2.046e6, 2046.0, p};
        geo.prn = 3;
        v.push_back(geo);
    }
    {
        Props p = base(25000000);
        v.push_back({"BeiDou B3I MEO", "BEIDOU_B3I_DLL_PLL_Tracking", 'C', "B3", 10.23e6, 10230.0, p});
        SignalCase geo{"BeiDou B3I GEO (PRN 59)", "BEIDOU_B3I_DLL_PLL_Tracking", 'C', "B3", 10.23e6, 10230.0, p};
        geo.prn = 59;
        v.push_back(geo);
    }
    {
        Props p = base(6625000);
        p[R + ".extend_correlation_symbols"] = "20";  // limited to 10
        v.push_back({"GLONASS L1 C/A (slot 7)", "GLONASS_L1_CA_DLL_PLL_Tracking", 'R', "1G", 511e3, 511.0, p});
        SignalCase l2{"GLONASS L2 C/A (slot 2)", "GLONASS_L2_CA_DLL_PLL_Tracking", 'R', "2G", 511e3, 511.0, p};
        l2.prn = 2;
        v.push_back(l2);
    }
    {
        Props p = base(4000000);
        SignalCase j1{"QZSS L1 C/A", "QZSS_L1_CA_DLL_PLL_Tracking", 'J', "J1", 1.023e6, 1023.0, p};
        j1.prn = 193;
        v.push_back(j1);
    }
    {
        Props p = base(25000000);
        p[R + ".track_pilot"] = "true";
        SignalCase j5{"QZSS L5 pilot", "QZSS_L5_DLL_PLL_Tracking", 'J', "J5", 10.23e6, 10230.0, p};
        j5.prn = 194;
        v.push_back(j5);
        p[R + ".track_pilot"] = "false";
        SignalCase j5d{"QZSS L5 data", "QZSS_L5_DLL_PLL_Tracking", 'J', "J5", 10.23e6, 10230.0, p};
        j5d.prn = 194;
        v.push_back(j5d);
    }
    return v;
}

void test_conf_mapping()
{
    for (const auto& sc : signal_cases())
        {
            const Dll_Pll_Conf p = adapter_conf(sc, "Tracking");
            void* ref = nullptr;
            if (sc.block_only)
                {
                    std::vector<const char*> k, v;
                    for (const auto& kv : sc.props)
                        {
                            k.push_back(kv.first.c_str());
                            v.push_back(kv.second.c_str());
                        }
                    // the adapter-level clamps of adapter_conf() are applied by handing the block the finished extend_correlation_symbols
                    Props pp = sc.props;
                    pp["Tracking.extend_correlation_symbols"] = std::to_string(p.extend_correlation_symbols);
                    k.clear();
                    v.clear();
                    for (const auto& kv : pp)
                        {
                            k.push_back(kv.first.c_str());
                            v.push_back(kv.second.c_str());
                        }
                    ref = reftrk_create_block(sc.system, sc.signal, p.vector_length, "Tracking", k.data(), v.data(), static_cast<int>(k.size()));
                }
            else
                ref = make_ref(sc.ref_impl, "Tracking", sc.props);
            EXPECT(ref != nullptr, "%s: reference chain could not be built", sc.name);
            if (ref == nullptr) continue;
            reftrk_conf_out r{};
            reftrk_get_conf(ref, &r);
            gsh_trk_conf c{};
            Hip_Trk_Signal sig;
            std::string why;
            const bool ok = hip_fill_trk_conf(p, &c, &sig, &why);
            EXPECT(ok, "%s: hip_fill_trk_conf: %s", sc.name, why.c_str());
            if (ok)
                {
                    compare_conf(sc.name, c, sig, r);
                    // local replicas: ours (hip_make_tracking_codes) vs what the block generated in start_tracking
                    reftrk_set_acquisition(ref, sc.system, sc.signal, sc.prn, 0.0, 0.0, 0);
                    reftrk_start_tracking(ref);
                    const int n = static_cast<int>(c.code_length_chips * c.code_samples_per_chip);
                    std::vector<float> rc(n), rd(n), code, data;
                    EXPECT(reftrk_get_codes(ref, rc.data(), rd.data(), n) == n, "%s: reference code length", sc.name);
                    const char sigc[3] = {sc.signal[0], sc.signal[1], '\0'};
                    EXPECT(hip_make_tracking_codes(sig, &c, sc.prn, sigc, &code, &data, &why), "%s: hip_make_tracking_codes: %s", sc.name, why.c_str());
                    EXPECT(code.size() == rc.size() && std::equal(code.begin(), code.end(), rc.begin()), "%s: tracking replica differs from the block's", sc.name);
                    if (c.track_pilot) EXPECT(data.size() == rd.size() && std::equal(data.begin(), data.end(), rd.begin()), "%s: data replica differs", sc.name);
                    // what start_tracking changes per satellite (BeiDou GEO, Glonass FDMA channel, per-PRN secondary codes): the block's members now
                    reftrk_get_conf(ref, &r);
                    int32_t ext_live = 0;
                    double cfo = 0.0;
                    reftrk_get_live(ref, &ext_live, &cfo);
                    EXPECT(c.symbols_per_bit == r.symbols_per_bit && c.has_secondary == r.secondary, "%s: after start_tracking symbols_per_bit %d / %d, secondary %d / %d", sc.name,
                        c.symbols_per_bit, r.symbols_per_bit, c.has_secondary, r.secondary);
                    // (the driver's record carries the first 255 characters of the string; the Glonass preamble pattern has 300)
                    EXPECT(std::string(reinterpret_cast<const char*>(c.secondary_code), std::min(c.secondary_code_length, 255)) == std::string(r.secondary_code) &&
                               c.secondary_code_length == r.secondary_code_length,
                        "%s: secondary code after start_tracking", sc.name);
                    EXPECT(c.data_secondary_code_length == r.data_secondary_code_length, "%s: data secondary code length after start_tracking %d / %d", sc.name,
                        c.data_secondary_code_length, r.data_secondary_code_length);
                    EXPECT(c.extend_correlation_symbols == ext_live, "%s: extend_correlation_symbols after start_tracking %d vs the block's %d", sc.name, c.extend_correlation_symbols, ext_live);
                    EXPECT(c.cfo_frequency_hz == cfo, "%s: cfo_frequency_hz %g vs the block's %g", sc.name, c.cfo_frequency_hz, cfo);
                    EXPECT(c.use_histogram_bit_sync == r.use_histogram_bit_sync, "%s: use_histogram_bit_sync %d vs the block's %d", sc.name, c.use_histogram_bit_sync, r.use_histogram_bit_sync);
                }
            reftrk_destroy(ref);
        }
    // unsupported signal / item type are refused with a reason
    {
        Dll_Pll_Conf p;
        p.system = 'G';
        std::memcpy(p.signal, "9Z", 3);
        gsh_trk_conf c{};
        Hip_Trk_Signal sig;
        std::string why;
        EXPECT(!hip_fill_trk_conf(p, &c, &sig, &why) && !why.empty(), "an unknown signal tag must be refused");
        p.system = 'G';
        std::memcpy(p.signal, "1C", 3);
        p.item_type = "cshort";
        EXPECT(!hip_fill_trk_conf(p, &c, &sig, &why), "cshort tracking items must be refused (the reference adapters refuse them too)");
    }
}

// ---- synthetic streams --------------------------------------------------------------------------------------------------------
// code: +-1 replica at `spc` samples per chip (as the tracking block holds it), starting at sample 0; symbol k multiplies code period k
std::vector<std::complex<float>> synth(const std::vector<float>& code, const std::vector<float>* code2, double w2, double chip_rate, int spc, double fs,
    double fd, double f_carrier, size_t n, float amp, const std::vector<int8_t>& symbols, const std::vector<int8_t>* symbols2, unsigned seed)
{
    std::mt19937 gen(seed);
    std::normal_distribution<float> g(0.0F, 1.0F);
    std::vector<std::complex<float>> x(n);
    const double rate = chip_rate * (1.0 + fd / f_carrier) / fs * spc;  // code samples per input sample
    const size_t L = code.size();
    for (size_t i = 0; i < n; i++)
        {
            const double pos = rate * static_cast<double>(i);
            const auto k = static_cast<long long>(std::floor(pos));
            const size_t idx = static_cast<size_t>(k % static_cast<long long>(L));
            const size_t period = static_cast<size_t>(k / static_cast<long long>(L));
            const float s1 = symbols.empty() ? 1.0F : static_cast<float>(symbols[std::min(period, symbols.size() - 1)]);
            float v = code[idx] * s1;
            if (code2 != nullptr)
                {
                    const float s2 = (symbols2 == nullptr || symbols2->empty()) ? 1.0F : static_cast<float>((*symbols2)[period % symbols2->size()]);
                    v = static_cast<float>((v + w2 * (*code2)[idx] * s2) / std::sqrt(2.0));
                }
            const double ph = std::fmod(2.0 * M_PI * fd / fs * static_cast<double>(i), 2.0 * M_PI);
            x[i] = std::complex<float>(g(gen), g(gen)) + amp * v * std::complex<float>(static_cast<float>(std::cos(ph)), static_cast<float>(std::sin(ph)));
        }
    return x;
}

float amp_for_cn0(double cn0_dbhz, double fs) { return static_cast<float>(std::sqrt(std::pow(10.0, cn0_dbhz / 10.0) * 2.0 / fs)); }

struct Period
{
    int consumed{0};
    int produced{0};
    Gnss_Synchro item{};
    long event{0};
    int32_t state{0};
};

// one scheduler call on our block
Period hip_call(gr::block& blk, const std::vector<std::complex<float>>& x, size_t& pos, int available)
{
    Period p;
    const int avail = static_cast<int>(std::min<size_t>(static_cast<size_t>(available), x.size() - pos));
    gr_vector_int nin{avail};
    gr_vector_const_void_star ins{static_cast<const void*>(x.data() + pos)};
    std::vector<Gnss_Synchro> outbuf(4);
    gr_vector_void_star outs{static_cast<void*>(outbuf.data())};
    blk.consumed_last = 0;
    const size_t ev0 = blk.published.size();
    p.produced = blk.general_work(1, nin, ins, outs);
    blk.mock_advance(p.produced);
    p.consumed = blk.consumed_last;
    pos += static_cast<size_t>(p.consumed);
    if (p.produced > 0) p.item = outbuf[0];
    if (blk.published.size() > ev0) p.event = pmt::to_long(blk.published.back().second);
    return p;
}

struct RefPeriod
{
    int consumed{0};
    int produced{0};
    reftrk_output o{};
};

// The throughput leg replays a stream that is periodic in g_cycle_len samples (0: every other case -- the stream is what the vector holds): the vector then holds
// one cycle plus a mirror of its beginning, so that what a call is offered is contiguous wherever it starts, and positions keep counting up.
size_t g_cycle_len = 0;
inline size_t stream_offset(size_t pos) { return g_cycle_len != 0 ? pos % g_cycle_len : pos; }

RefPeriod ref_call(void* ref, const std::vector<std::complex<float>>& x, size_t& pos, int available)
{
    RefPeriod p;
    const int avail = static_cast<int>(std::min<size_t>(static_cast<size_t>(available), x.size() - stream_offset(pos)));
    p.produced = reftrk_general_work(ref, reinterpret_cast<const float*>(x.data() + stream_offset(pos)), avail, &p.consumed, &p.o);
    pos += static_cast<size_t>(p.consumed);
    return p;
}

// drive both chains over the same stream from the same hand-over; returns the number of leading periods with identical window positions
struct TrajectoryStats
{
    int periods{0};
    int same_windows{0};      // leading periods whose consumed counts (= window positions) are identical
    int symbols_ref{0}, symbols_hip{0}, symbols_matched{0};
    double worst_prompt_rel{0.0};
    double final_doppler_ref{0.0}, final_doppler_hip{0.0};
    double final_cn0_ref{0.0}, final_cn0_hip{0.0};
    long hip_event{0};
    int ref_events{0};
    int first_symbol_period_ref{-1}, first_symbol_period_hip{-1};
    int loss_period_ref{-1}, loss_period_hip{-1};
    bool loss_item_hip{false}, loss_item_ref{false};
};

TrajectoryStats run_pair(DllPllTrackingHip& hip, void* ref, Gnss_Synchro& syn, const std::vector<std::complex<float>>& x, int vector_length, int n_periods,
    char system, const char* signal, uint32_t prn, double acq_delay, double acq_doppler, uint64_t acq_stamp)
{
    TrajectoryStats st;
    auto blk = std::dynamic_pointer_cast<gr::block>(hip.get_left_block());
    if (fake_gsh_set_reference_stream != nullptr) fake_gsh_set_reference_stream(reinterpret_cast<const float*>(x.data()), x.size());
    syn = Gnss_Synchro{};
    syn.System = system;
    std::memcpy(syn.Signal, signal, 3);
    syn.PRN = prn;
    syn.Acq_delay_samples = acq_delay;
    syn.Acq_doppler_hz = acq_doppler;
    syn.Acq_samplestamp_samples = acq_stamp;
    hip.set_channel(5);
    hip.set_gnss_synchro(&syn);
    reftrk_set_acquisition(ref, system, signal, prn, acq_delay, acq_doppler, acq_stamp);
    size_t ph = 0, pr = 0;
    const int avail = 2 * vector_length;
    // standby: both consume what they are offered
    Period a = hip_call(*blk, x, ph, avail);
    RefPeriod b = ref_call(ref, x, pr, avail);
    EXPECT(a.consumed == avail && b.consumed == avail && a.produced == 0 && b.produced == 0, "standby: consumed %d / %d", a.consumed, b.consumed);
    hip.start_tracking();
    reftrk_start_tracking(ref);
    // pull-in
    a = hip_call(*blk, x, ph, avail);
    b = ref_call(ref, x, pr, avail);
    EXPECT(a.consumed == b.consumed && a.produced == 0 && b.produced == 0, "pull-in: consumed %d vs reference %d", a.consumed, b.consumed);
    bool same = (a.consumed == b.consumed);
    std::vector<int> sym_ref, sym_hip;
    std::vector<double> pi_ref, pi_hip;
    for (int k = 0; k < n_periods; k++)
        {
            if (ph + static_cast<size_t>(avail) > x.size() || pr + static_cast<size_t>(avail) > x.size()) break;
            a = hip_call(*blk, x, ph, avail);
            b = ref_call(ref, x, pr, avail);
            st.periods++;
            if (std::getenv("TRK_DEBUG") != nullptr && k < 120)
                std::printf("  k %d ref state %d P %.0f %.0f consumed %d | hip state %d consumed %d produced %d/%d\n", k, b.o.state, b.o.corr[2], b.o.corr[3], b.consumed, a.state,
                    a.consumed, b.produced, a.produced);
            if (same && a.consumed == b.consumed)
                st.same_windows++;
            else
                same = false;
            if (b.produced > 0 && b.o.flag_valid_symbol_output != 0)
                {
                    sym_ref.push_back(k);
                    pi_ref.push_back(b.o.prompt_i);
                    if (st.first_symbol_period_ref < 0) st.first_symbol_period_ref = k;
                    st.final_doppler_ref = b.o.carrier_doppler_hz;
                    st.final_cn0_ref = b.o.cn0_db_hz;
                }
            if (a.produced > 0 && a.event != 3)
                {
                    sym_hip.push_back(k);
                    pi_hip.push_back(a.item.Prompt_I);
                    if (st.first_symbol_period_hip < 0) st.first_symbol_period_hip = k;
                    st.final_doppler_hip = a.item.Carrier_Doppler_hz;
                    st.final_cn0_hip = a.item.CN0_dB_hz;
                    EXPECT(a.item.Flag_valid_symbol_output && a.item.PRN == prn && a.item.System == system, "symbol item header");
                    if (same && b.produced > 0)
                        {
                            EXPECT(a.item.Tracking_sample_counter == b.o.tracking_sample_counter, "period %d: Tracking_sample_counter %llu vs %llu", k,
                                static_cast<unsigned long long>(a.item.Tracking_sample_counter), static_cast<unsigned long long>(b.o.tracking_sample_counter));
                            EXPECT(a.item.correlation_length_ms == b.o.correlation_length_ms, "correlation_length_ms");
                            EXPECT(a.item.Flag_PLL_180_deg_phase_locked == (b.o.flag_pll_180_deg_phase_locked != 0), "period %d: PLL 180 flag", k);
                        }
                }
            if (a.event != 0) st.hip_event = a.event;
            if (a.event == 3 && st.loss_period_hip < 0)
                {
                    st.loss_period_hip = k;
                    st.loss_item_hip = (a.produced == 1 && !a.item.Flag_valid_symbol_output);
                }
            if (b.o.n_events > st.ref_events && b.o.events[b.o.n_events - 1] == 3 && st.loss_period_ref < 0)
                {
                    st.loss_period_ref = k;
                    st.loss_item_ref = (b.produced == 1 && b.o.flag_valid_symbol_output == 0);
                }
            st.ref_events = b.o.n_events;
            if (a.event == 3 || b.o.state == 0) break;
        }
    st.symbols_ref = static_cast<int>(sym_ref.size());
    st.symbols_hip = static_cast<int>(sym_hip.size());
    for (size_t i = 0, j = 0; i < sym_ref.size() && j < sym_hip.size();)
        {
            if (sym_ref[i] == sym_hip[j])
                {
                    st.symbols_matched++;
                    const double rel = std::fabs(pi_ref[i] - pi_hip[j]) / std::max(1.0, std::fabs(pi_ref[i]));
                    st.worst_prompt_rel = std::max(st.worst_prompt_rel, rel);
                    i++;
                    j++;
                }
            else if (sym_ref[i] < sym_hip[j])
                i++;
            else
                j++;
        }
    return st;
}

std::vector<int8_t> random_symbols(size_t n, unsigned seed)
{
    std::mt19937 gen(seed);
    std::vector<int8_t> s(n);
    for (auto& v : s) v = (gen() & 1U) ? 1 : -1;
    return s;
}

void test_gps_l1_trajectory()
{
    const long fs = 4000000;
    const int n = 4000;
    const uint32_t prn = 9;
    const double fd = 1234.0;
    const std::string R = "Tracking";
    Props p{{"GNSS-SDR.internal_fs_sps", std::to_string(fs)}, {R + ".pll_bw_hz", "35.0"}, {R + ".dll_bw_hz", "2.0"}, {R + ".early_late_space_chips", "0.5"},
        {R + ".pull_in_time_s", "0"}, {R + ".hip_device", "0"}, {R + ".hip_register_input_buffer", "false"}};
    auto cfg = make_config(p);
    GpsL1CaDllPllTrackingHip hip(cfg.get(), R, 1, 1);
    EXPECT(hip.implementation() == "GPS_L1_CA_DLL_PLL_Tracking_HIP" && hip.role() == R, "names");
    EXPECT(hip.item_size() == sizeof(gr_complex), "item_size %zu", hip.item_size());
    if (hip.item_size() == 0) return;
    void* ref = make_ref("GPS_L1_CA_DLL_PLL_Tracking", R, p);
    EXPECT(ref != nullptr, "reference chain");
    // the adapter's Dll_Pll_Conf is what adapter_conf() re-derives (keeps the CPU-only conf test honest)
    {
        SignalCase sc{"GPS L1 C/A", "GPS_L1_CA_DLL_PLL_Tracking", 'G', "1C", 1.023e6, 1023.0, p};
        const Dll_Pll_Conf q = adapter_conf(sc, R);
        const Dll_Pll_Conf& a = hip.tracking_parameters();
        EXPECT(q.vector_length == a.vector_length && q.track_pilot == a.track_pilot && q.extend_correlation_symbols == a.extend_correlation_symbols &&
                   q.system == a.system && std::string(q.signal) == std::string(a.signal),
            "adapter_conf() out of step with the adapter");
        reftrk_conf_out r{};
        reftrk_get_conf(ref, &r);
        Hip_Trk_Signal sig;
        sig.correlation_length_ms = r.correlation_length_ms;
        sig.interchange_iq = r.interchange_iq != 0;
        compare_conf("GPS L1 C/A (adapter)", hip.trk_conf(), sig, r);
    }
    // navigation bits: 30 random, the TLM preamble 10001011, 60 random; one bit = 20 code periods; bits start at code period 7
    std::vector<int8_t> bits = random_symbols(30, 4);
    for (char ch : std::string("10001011")) bits.push_back(ch == '1' ? 1 : -1);
    const std::vector<int8_t> tail = random_symbols(60, 5);
    bits.insert(bits.end(), tail.begin(), tail.end());
    const int n_periods = 1000 + 20 * static_cast<int>(bits.size()) / 2 + 500;
    std::vector<int8_t> symbols(static_cast<size_t>(n_periods) + 40, 1);
    for (size_t k = 0; k < symbols.size(); k++)
        {
            const long b = (static_cast<long>(k) - 7) / 20;
            if (static_cast<long>(k) >= 7 && b < static_cast<long>(bits.size())) symbols[k] = bits[static_cast<size_t>(b)];
        }
    std::vector<float> code(1023);
    gps_l1_ca_code_gen_float(code, static_cast<int32_t>(prn), 0);
    const auto x = synth(code, nullptr, 0.0, 1.023e6, 1, fs, fd, 1575.42e6, static_cast<size_t>(n_periods + 12) * n, amp_for_cn0(47.0, fs), symbols, nullptr, 11);
    Gnss_Synchro syn;
    const TrajectoryStats st = run_pair(hip, ref, syn, x, n, n_periods, 'G', "1C", prn, 0.0, fd - 15.0, n);
    std::printf("GPS L1: %d periods, %d with identical windows, symbols ref/hip/matched %d/%d/%d, first symbol at period %d / %d, worst |dPrompt_I| rel %.2e, "
                "Doppler %.2f / %.2f Hz, C/N0 %.2f / %.2f dB-Hz\n",
        st.periods, st.same_windows, st.symbols_ref, st.symbols_hip, st.symbols_matched, st.first_symbol_period_ref, st.first_symbol_period_hip, st.worst_prompt_rel,
        st.final_doppler_ref, st.final_doppler_hip, st.final_cn0_ref, st.final_cn0_hip);
    EXPECT(st.periods >= n_periods - 2, "ran %d of %d periods", st.periods, n_periods);
    EXPECT(st.same_windows >= st.periods * 9 / 10, "window positions identical for only %d of %d periods", st.same_windows, st.periods);
    EXPECT(st.symbols_ref >= 10 && st.symbols_hip == st.symbols_ref && st.symbols_matched == st.symbols_ref, "symbol timing: ref %d hip %d matched %d", st.symbols_ref,
        st.symbols_hip, st.symbols_matched);
    EXPECT(st.first_symbol_period_hip == st.first_symbol_period_ref, "state-4 hand-over at period %d vs reference %d", st.first_symbol_period_hip, st.first_symbol_period_ref);
    EXPECT(st.worst_prompt_rel < 2e-2, "Prompt_I of the symbols differs by %.3e", st.worst_prompt_rel);
    EXPECT(std::fabs(st.final_doppler_hip - st.final_doppler_ref) < 1.0 && std::fabs(st.final_doppler_hip - fd) < 5.0, "Doppler %.2f vs %.2f", st.final_doppler_hip,
        st.final_doppler_ref);
    EXPECT(std::fabs(st.final_cn0_hip - st.final_cn0_ref) < 0.5, "C/N0 %.2f vs %.2f", st.final_cn0_hip, st.final_cn0_ref);
    EXPECT(st.hip_event == 0 && st.ref_events == 0, "no loss of lock expected (hip event %ld, reference events %d)", st.hip_event, st.ref_events);

    // ---- telemetry fault message -> forced loss of lock (trk.cc:757-769): "events" 3 and an item with Flag_valid_symbol_output = false
    {
        auto blk = std::dynamic_pointer_cast<gr::block>(hip.get_left_block());
        blk->deliver("telemetry_to_trk", pmt::make_any(std::any(1)));
        size_t pos = static_cast<size_t>(blk->nitems_read(0));
        const Period a = hip_call(*blk, x, pos, 2 * n);
        EXPECT(a.event == 3 && a.produced == 1 && !a.item.Flag_valid_symbol_output, "telemetry fault: event %ld produced %d", a.event, a.produced);
        const Period b = hip_call(*blk, x, pos, 2 * n);  // back in standby
        EXPECT(b.produced == 0 && b.consumed == 2 * n, "standby after loss of lock: consumed %d", b.consumed);
    }
    reftrk_destroy(ref);
}

void test_galileo_e1_pilot_trajectory()
{
    const long fs = 4000000;
    const int n = 16000;
    const uint32_t prn = 11;
    const double fd = -2200.0;
    const std::string R = "Tracking";
    Props p{{"GNSS-SDR.internal_fs_sps", std::to_string(fs)}, {R + ".pll_bw_hz", "15.0"}, {R + ".dll_bw_hz", "0.75"}, {R + ".early_late_space_chips", "0.15"},
        {R + ".very_early_late_space_chips", "0.6"}, {R + ".track_pilot", "true"}, {R + ".pull_in_time_s", "0"}, {R + ".hip_device", "0"}, {R + ".hip_register_input_buffer", "false"}};
    auto cfg = make_config(p);
    GalileoE1DllPllVemlTrackingHip hip(cfg.get(), R, 1, 1);
    EXPECT(hip.implementation() == "Galileo_E1_DLL_PLL_VEML_Tracking_HIP", "name");
    EXPECT(hip.item_size() == sizeof(gr_complex), "item_size %zu", hip.item_size());
    if (hip.item_size() == 0) return;
    void* ref = make_ref("Galileo_E1_DLL_PLL_VEML_Tracking", R, p);
    EXPECT(ref != nullptr, "reference chain");
    std::vector<float> e1b(8184), e1c(8184);
    const std::array<char, 3> sb{{'1', 'B', '\0'}}, sc{{'1', 'C', '\0'}};
    galileo_e1_code_gen_sinboc11_float(e1b, sb, prn);
    galileo_e1_code_gen_sinboc11_float(e1c, sc, prn);
    const int n_periods = 420;
    const std::vector<int8_t> data = random_symbols(static_cast<size_t>(n_periods) + 20, 8);
    std::vector<int8_t> sec;
    for (char ch : std::string(GALILEO_E1_C_SECONDARY_CODE)) sec.push_back(ch == '0' ? 1 : -1);
    // E1 OS: (e1b * data - e1c * secondary) / sqrt(2)
    const auto x = synth(e1b, &e1c, -1.0, 1.023e6, 2, fs, fd, 1575.42e6, static_cast<size_t>(n_periods + 6) * n, amp_for_cn0(47.0, fs), data, &sec, 13);
    Gnss_Synchro syn;
    const TrajectoryStats st = run_pair(hip, ref, syn, x, n, n_periods, 'E', "1B", prn, 0.0, fd + 10.0, n);
    std::printf("Galileo E1: %d periods, %d with identical windows, symbols ref/hip/matched %d/%d/%d, first symbol at period %d / %d, worst |dPrompt_I| rel %.2e, "
                "Doppler %.2f / %.2f Hz, C/N0 %.2f / %.2f dB-Hz\n",
        st.periods, st.same_windows, st.symbols_ref, st.symbols_hip, st.symbols_matched, st.first_symbol_period_ref, st.first_symbol_period_hip, st.worst_prompt_rel,
        st.final_doppler_ref, st.final_doppler_hip, st.final_cn0_ref, st.final_cn0_hip);
    EXPECT(st.same_windows >= st.periods * 9 / 10, "window positions identical for only %d of %d periods", st.same_windows, st.periods);
    EXPECT(st.symbols_ref >= 50 && st.symbols_hip == st.symbols_ref && st.symbols_matched == st.symbols_ref, "symbol timing: ref %d hip %d matched %d", st.symbols_ref,
        st.symbols_hip, st.symbols_matched);
    EXPECT(st.first_symbol_period_hip == st.first_symbol_period_ref, "secondary-code lock at period %d vs reference %d", st.first_symbol_period_hip, st.first_symbol_period_ref);
    EXPECT(st.worst_prompt_rel < 2e-2, "Prompt_I of the symbols differs by %.3e", st.worst_prompt_rel);
    EXPECT(std::fabs(st.final_doppler_hip - st.final_doppler_ref) < 1.0, "Doppler %.2f vs %.2f", st.final_doppler_hip, st.final_doppler_ref);
    reftrk_destroy(ref);
}

// ---- the other signals of the family: adapter + block + device loop against the reference chain on one synthetic satellite each.  The stream carries the
// signal's secondary code (where it has one) times random telemetry symbols, so that both chains go through their synchronisation state and publish symbols.
template <typename Adapter>
void signal_trajectory(const char* name, const char* ref_impl, const char* hip_impl, char system, const char* signal, uint32_t prn, long fs, double chip_rate,
    double f_carrier, Props extra, int n_periods, const std::string& secondary, int symbols_per_bit, double cfo_hz, bool expect_symbols, unsigned seed,
    const std::string& preamble_bits = std::string(), double doppler_tol_hz = 5.0, int preamble_symbols_at = -1)
{
    const std::string R = "Tracking";
    Props p{{"GNSS-SDR.internal_fs_sps", std::to_string(fs)}, {R + ".pll_bw_hz", "35.0"}, {R + ".dll_bw_hz", "2.0"}, {R + ".early_late_space_chips", "0.5"},
        {R + ".pull_in_time_s", "0"}, {R + ".hip_device", "0"}, {R + ".hip_register_input_buffer", "false"}};
    for (const auto& kv : extra) p[kv.first] = kv.second;
    auto cfg = make_config(p);
    Adapter hip(cfg.get(), R, 1, 1);
    EXPECT(hip.implementation() == hip_impl, "%s: implementation name %s", name, hip.implementation().c_str());
    EXPECT(hip.item_size() == sizeof(gr_complex), "%s: unusable adapter", name);
    if (hip.item_size() == 0) return;
    void* ref = make_ref(ref_impl, R, p);
    EXPECT(ref != nullptr, "%s: reference chain", name);
    if (ref == nullptr) return;
    const gsh_trk_conf& c0 = hip.trk_conf();
    const int n = static_cast<int>(c0.vector_length);
    // the replica as the block generates it
    gsh_trk_conf c = c0;
    Hip_Trk_Signal sig;
    std::string why;
    {
        gsh_trk_conf tmp{};
        EXPECT(hip_fill_trk_conf(hip.tracking_parameters(), &tmp, &sig, &why), "%s: %s", name, why.c_str());
    }
    std::vector<float> code, data;
    const char sigc[3] = {signal[0], signal[1], '\0'};
    EXPECT(hip_make_tracking_codes(sig, &c, prn, sigc, &code, &data, &why), "%s: %s", name, why.c_str());
    // symbols: secondary code chip x telemetry bit
    std::vector<int8_t> bits = random_symbols(static_cast<size_t>(n_periods / std::max(1, symbols_per_bit) + 4), seed);
    if (!preamble_bits.empty() && bits.size() > 12 + preamble_bits.size())  // a telemetry preamble for the chains that synchronise on it
        for (size_t i = 0; i < preamble_bits.size(); i++) bits[12 + i] = preamble_bits[i] == '1' ? 1 : -1;
    std::vector<int8_t> symbols(static_cast<size_t>(n_periods) + 40, 1);
    for (size_t k = 0; k < symbols.size(); k++)
        {
            int8_t v = bits[std::min(bits.size() - 1, k / static_cast<size_t>(std::max(1, symbols_per_bit)))];
            if (!secondary.empty()) v = static_cast<int8_t>(v * (secondary[k % secondary.size()] == '0' ? 1 : -1));
            symbols[k] = v;
        }
    if (preamble_symbols_at >= 0)  // the chain's own preamble, symbol by symbol (GLONASS: 30 time-mark bits of 10 symbols), twice
        for (int rep = 0; rep < 2; rep++)
            for (int i = 0; i < c.secondary_code_length; i++)
                {
                    const size_t k = static_cast<size_t>(preamble_symbols_at + rep * (c.secondary_code_length + 170) + i);
                    if (k < symbols.size()) symbols[k] = c.secondary_code[i] == '1' ? 1 : -1;
                }
    const double fd = 777.0;
    // (the FDMA channel offset is part of the received carrier; code Doppler follows the satellite's Doppler only)
    auto x = synth(code, nullptr, 0.0, chip_rate, static_cast<int>(c.code_samples_per_chip), static_cast<double>(fs), fd, f_carrier, static_cast<size_t>(n_periods + 12) * n,
        amp_for_cn0(48.0, static_cast<double>(fs)), symbols, nullptr, seed + 1);
    if (cfo_hz != 0.0)
        for (size_t i = 0; i < x.size(); i++)
            {
                const double ph = std::fmod(2.0 * M_PI * cfo_hz / static_cast<double>(fs) * static_cast<double>(i), 2.0 * M_PI);
                x[i] *= std::complex<float>(static_cast<float>(std::cos(ph)), static_cast<float>(std::sin(ph)));
            }
    Gnss_Synchro syn;
    const TrajectoryStats st = run_pair(hip, ref, syn, x, n, n_periods, system, signal, prn, 0.0, fd - 12.0, static_cast<uint64_t>(n));
    std::printf("%s: %d periods, %d with identical windows, symbols ref/hip/matched %d/%d/%d, first symbol at period %d / %d, worst |dPrompt_I| rel %.2e, Doppler %.2f / %.2f Hz, "
                "C/N0 %.2f / %.2f dB-Hz\n",
        name, st.periods, st.same_windows, st.symbols_ref, st.symbols_hip, st.symbols_matched, st.first_symbol_period_ref, st.first_symbol_period_hip, st.worst_prompt_rel,
        st.final_doppler_ref, st.final_doppler_hip, st.final_cn0_ref, st.final_cn0_hip);
    EXPECT(st.periods >= n_periods - 2, "%s: ran %d of %d periods", name, st.periods, n_periods);
    EXPECT(st.same_windows >= st.periods * 9 / 10, "%s: window positions identical for only %d of %d periods", name, st.same_windows, st.periods);
    EXPECT(st.symbols_hip == st.symbols_ref && st.symbols_matched == st.symbols_ref, "%s: symbol timing: ref %d hip %d matched %d", name, st.symbols_ref, st.symbols_hip, st.symbols_matched);
    EXPECT(st.first_symbol_period_hip == st.first_symbol_period_ref, "%s: first symbol at period %d vs reference %d", name, st.first_symbol_period_hip, st.first_symbol_period_ref);
    if (expect_symbols)
        {
            EXPECT(st.symbols_ref >= 10, "%s: the reference chain published only %d symbols", name, st.symbols_ref);
            EXPECT(st.worst_prompt_rel < 2e-2, "%s: Prompt_I of the symbols differs by %.3e", name, st.worst_prompt_rel);
            EXPECT(std::fabs(st.final_doppler_hip - st.final_doppler_ref) < 1.0 && std::fabs(st.final_doppler_hip - fd) < doppler_tol_hz, "%s: Doppler %.2f vs %.2f", name, st.final_doppler_hip,
                st.final_doppler_ref);
            EXPECT(std::fabs(st.final_cn0_hip - st.final_cn0_ref) < 0.5, "%s: C/N0 %.2f vs %.2f", name, st.final_cn0_hip, st.final_cn0_ref);
        }
    EXPECT(st.hip_event == 0 && st.ref_events == 0, "%s: no loss of lock expected (hip event %ld, reference events %d)", name, st.hip_event, st.ref_events);
    reftrk_destroy(ref);
}

void test_other_signal_trajectories()
{
    // The block stays in its pull-in state until one whole second of samples has passed, whatever pull_in_time_s says: the elapsed time is an integer
    // division by fs_in (trk.cc:1912), so the 1 ms signals need more than 1000 periods before the secondary code / bit search starts.
    // BeiDou B1I, MEO satellite: 20-chip NH code on every bit (trk.cc:412-430, 958-971)
    signal_trajectory<BeidouB1iDllPllTrackingHip>("BeiDou B1I", "BEIDOU_B1I_DLL_PLL_Tracking", "BEIDOU_B1I_DLL_PLL_Tracking_HIP", 'C', "B1", 8, 8000000, BEIDOU_B1I_CODE_RATE_CPS,
        BEIDOU_B1I_FREQ_HZ, {}, 1400, BEIDOU_B1I_SECONDARY_CODE_STR, 20, 0.0, true, 21);
    // QZSS L1 C/A: GPS-like, histogram bit synchronisation then the preamble
    signal_trajectory<QzssL1DllPllTrackingHip>("QZSS L1 C/A", "QZSS_L1_CA_DLL_PLL_Tracking", "QZSS_L1_CA_DLL_PLL_Tracking_HIP", 'J', "J1", 193, 4000000, QZSS_L1_CHIP_RATE, QZSS_L1_FREQ_HZ, {},
        2500, "", 20, 0.0, true, 22, "10001011");
    // GLONASS L1 C/A, slot 7 (frequency channel +5: 2.8125 MHz above the band centre, in a 6.625 Msps stream): the loop carries the channel offset in its NCO
    // (trk.cc:996-1003); bit synchronisation is the correlation with the 300-symbol GNAV time mark (no histogram for GLONASS), put into the stream
    // after the one-second pull-in
    signal_trajectory<GlonassL1CaDllPllTrackingHip>("GLONASS L1 C/A", "GLONASS_L1_CA_DLL_PLL_Tracking", "GLONASS_L1_CA_DLL_PLL_Tracking_HIP", 'R', "1G", 7, 6625000, GLONASS_L1_CA_CODE_RATE_CPS,
        GLONASS_L1_CA_FREQ_HZ, {}, 1900, "", 10, DFRQ1_GLO * GLONASS_PRN.at(7), true, 23, "", 5.0, 1100);
    // Galileo E5b data component: 4-chip secondary code
    signal_trajectory<GalileoE5bDllPllTrackingHip>("Galileo E5b-I", "Galileo_E5b_DLL_PLL_Tracking", "Galileo_E5b_DLL_PLL_Tracking_HIP", 'E', "7X", 11, 12500000, GALILEO_E5B_CODE_CHIP_RATE_CPS,
        GALILEO_E5B_FREQ_HZ, {{"Tracking.track_pilot", "false"}}, 1300, GALILEO_E5B_I_SECONDARY_CODE, 4, 0.0, true, 24);
    // GPS L2C(M): 20 ms code periods, one per symbol
    signal_trajectory<GpsL2MDllPllTrackingHip>("GPS L2C(M)", "GPS_L2_M_DLL_PLL_Tracking", "GPS_L2_M_DLL_PLL_Tracking_HIP", 'G', "2S", 5, 2000000, GPS_L2_M_CODE_RATE_CPS, GPS_L2_FREQ_HZ,
        {{"Tracking.pll_bw_hz", "5.0"}, {"Tracking.dll_bw_hz", "0.5"}}, 120, "", 1, 0.0, true, 25, "", 30.0);
}


// ---- the hand-over as it happens in a flowgraph: the tracking block runs only when two code periods of input are there, the acquisition block consumes whatever it is
// offered, so at start_tracking the tracking block's read pointer is usually BEHIND Acq_samplestamp_samples.  The reference looks at its pull-in latch in every
// general_work, the pull-in call included (trk.cc:1910-1917): nitems_read - d_acq_sample_stamp wraps round (unsigned) and the transitory is over before the first period --
// bit synchronisation starts at once, a second earlier than with the read pointer ahead of the stamp.  Found by running the reference's own Channel (tests/host/test_channel.cc).
void test_handover_with_the_read_pointer_behind_the_stamp()
{
    const long fs = 4000000;
    const int n = 4000;
    const uint32_t prn = 14;
    const double fd = -1530.0;
    const std::string R = "Tracking";
    Props p{{"GNSS-SDR.internal_fs_sps", std::to_string(fs)}, {R + ".pll_bw_hz", "35.0"}, {R + ".dll_bw_hz", "2.0"}, {R + ".early_late_space_chips", "0.5"},
        {R + ".pull_in_time_s", "0"}, {R + ".hip_device", "0"}, {R + ".hip_register_input_buffer", "false"}};
    auto cfg = make_config(p);
    GpsL1CaDllPllTrackingHip hip(cfg.get(), R, 1, 1);
    if (hip.item_size() == 0) return;
    void* ref = make_ref("GPS_L1_CA_DLL_PLL_Tracking", R, p);
    const int n_periods = 700;
    std::vector<int8_t> symbols(static_cast<size_t>(n_periods) + 40, 1);
    const std::vector<int8_t> bits = random_symbols(static_cast<size_t>(n_periods) / 20 + 4, 31);
    for (size_t k = 0; k < symbols.size(); k++) symbols[k] = bits[std::min(bits.size() - 1, (k + 13) / 20)];
    std::vector<float> code(1023);
    gps_l1_ca_code_gen_float(code, static_cast<int32_t>(prn), 0);
    const auto x = synth(code, nullptr, 0.0, 1.023e6, 1, fs, fd, 1575.42e6, static_cast<size_t>(n_periods + 12) * n, amp_for_cn0(47.0, fs), symbols, nullptr, 17);
    Gnss_Synchro syn;
    // run_pair offers both blocks 2 n samples in standby: the read pointer stands at 8000 when start_tracking comes; the acquisition's stamp is 9500, its code phase
    // the 2500 samples from there to the next code period (the stream's periods start at multiples of ~4000)
    const TrajectoryStats st = run_pair(hip, ref, syn, x, n, n_periods, 'G', "1C", prn, 2500.0, fd + 9.0, static_cast<uint64_t>(2 * n + 1500));
    std::printf("hand-over with the read pointer behind the stamp: %d periods, %d with identical windows, symbols ref/hip/matched %d/%d/%d, first symbol at period %d / %d\n", st.periods,
        st.same_windows, st.symbols_ref, st.symbols_hip, st.symbols_matched, st.first_symbol_period_ref, st.first_symbol_period_hip);
    EXPECT(st.first_symbol_period_ref >= 0 && st.first_symbol_period_ref < 900, "the reference block did not leave the pull-in transitory at once (first symbol at period %d)", st.first_symbol_period_ref);
    EXPECT(st.first_symbol_period_hip == st.first_symbol_period_ref, "first symbol at period %d vs reference %d", st.first_symbol_period_hip, st.first_symbol_period_ref);
    EXPECT(st.symbols_hip == st.symbols_ref && st.symbols_matched == st.symbols_ref && st.symbols_ref >= 5, "symbol timing: ref %d hip %d matched %d", st.symbols_ref, st.symbols_hip, st.symbols_matched);
    EXPECT(st.same_windows >= st.periods * 9 / 10, "window positions identical for only %d of %d periods", st.same_windows, st.periods);
    EXPECT(st.hip_event == 0 && st.ref_events == 0, "no loss of lock expected (hip event %ld, reference events %d)", st.hip_event, st.ref_events);
    reftrk_destroy(ref);
    // ... and MORE than a code period behind: the pull-in lands on a code period that still starts before the stamp, so in the first periods of state 2 the
    // bit-synchronisation time limit's elapsed time (trk.cc:2002, the same unsigned difference) is ~2^64 / fs seconds: "time limit reached", the carrier fail counter
    // jumps to 300000 and the channel is dropped as soon as the lock test runs at all, i.e. when the C/N0 buffer has its 20 prompts.  A quirk; ours too.
    {
        Props p2 = p;
        p2[R + ".hip_shared_ring"] = "-1";  // (the stream starts again at sample 0: a ring of its own, not the one the first block left at the end of the stream)
        auto cfg2 = make_config(p2);
        GpsL1CaDllPllTrackingHip hip2(cfg2.get(), R, 1, 1);
        void* ref2 = make_ref("GPS_L1_CA_DLL_PLL_Tracking", R, p);
        Gnss_Synchro syn2;
        const TrajectoryStats s2 = run_pair(hip2, ref2, syn2, x, n, 200, 'G', "1C", prn, 2000.0, fd + 9.0, static_cast<uint64_t>(2 * n + 10000));
        std::printf("hand-over with the read pointer 2.5 code periods behind the stamp: loss of lock at period %d (HIP block) / %d (reference), %d periods with identical windows\n",
            s2.loss_period_hip, s2.loss_period_ref, s2.same_windows);
        EXPECT(s2.loss_period_ref >= 15 && s2.loss_period_ref <= 25, "the reference block kept the channel (loss at period %d): test set-up", s2.loss_period_ref);
        EXPECT(s2.loss_period_hip == s2.loss_period_ref, "loss of lock at period %d vs reference %d", s2.loss_period_hip, s2.loss_period_ref);
        EXPECT(s2.loss_item_hip && s2.loss_item_ref, "the loss must come with an item whose Flag_valid_symbol_output is false");
        reftrk_destroy(ref2);
    }
}

void test_loss_of_lock_on_noise()
{
    // (Round 4: cn0_min 40, not 35.  On noise the two loops part after a few hundred periods -- a rounding difference in a correlator output flips a window length,
    //  and from there the noise-driven trajectories are different ones; the M2M4 estimate of this noise wanders between 22 and 35.3 dB-Hz, so with the limit at 35
    //  the period in which the twenty-first fail comes depended on which trajectory it was.  At 40 every period after the pull-in transitory fails in both chains.)
    // noise only, with the lock thresholds tightened so that the detectors act within a few hundred periods: cn0_min 40 dB-Hz,
    // max_lock_fail 20 (dll_pll_conf.cc: <role>.cn0_min, <role>.max_lock_fail).  Both chains must drop the channel in the same period, with
    // "events" 3 and one item carrying Flag_valid_symbol_output = false (trk.cc:1208-1221, 2009-2014, 2285-2294).
    const long fs = 4000000;
    const int n = 4000;
    const std::string R = "Tracking";
    Props p{{"GNSS-SDR.internal_fs_sps", std::to_string(fs)}, {R + ".pull_in_time_s", "0"}, {R + ".hip_device", "0"}, {R + ".hip_register_input_buffer", "false"}, {R + ".cn0_min", "40"},
        {R + ".max_lock_fail", "20"}};
    Props ph = p;
    if (std::getenv("GSH_TEST_LOSS_DUMP") != nullptr)  // the HIP block's records to <file>1.dat (a debugging aid; the reference block is not given the key)
        {
            ph[R + ".dump"] = "true";
            ph[R + ".dump_mat"] = "false";
            ph[R + ".dump_filename"] = std::getenv("GSH_TEST_LOSS_DUMP");
        }
    auto cfg = make_config(ph);
    GpsL1CaDllPllTrackingHip hip(cfg.get(), R, 1, 1);
    if (hip.item_size() == 0) return;
    void* ref = make_ref("GPS_L1_CA_DLL_PLL_Tracking", R, p);
    EXPECT(hip.trk_conf().cn0_min == 40 && hip.trk_conf().max_code_lock_fail == 20, "cn0_min %d max_code_lock_fail %d", hip.trk_conf().cn0_min,
        hip.trk_conf().max_code_lock_fail);
    std::vector<float> code(1023, 1.0F);
    const auto x = synth(code, nullptr, 0.0, 1.023e6, 1, fs, 0.0, 1575.42e6, static_cast<size_t>(2600) * n, 0.0F, {}, nullptr, 21);  // noise only
    Gnss_Synchro syn;
    const TrajectoryStats st = run_pair(hip, ref, syn, x, n, 2500, 'G', "1C", 3, 100.0, 500.0, n);
    std::printf("noise only: loss of lock at period %d (HIP block) / %d (reference), %d periods with identical windows; last C/N0 %.2f / %.2f dB-Hz, Doppler %.1f / %.1f Hz\n",
        st.loss_period_hip, st.loss_period_ref, st.same_windows, st.final_cn0_hip, st.final_cn0_ref, st.final_doppler_hip, st.final_doppler_ref);
    EXPECT(st.loss_period_ref >= 0, "the reference did not drop the channel in %d periods -- test set-up", st.periods);
    EXPECT(st.loss_period_hip == st.loss_period_ref, "loss of lock at period %d vs reference %d", st.loss_period_hip, st.loss_period_ref);
    EXPECT(st.loss_item_hip && st.loss_item_ref, "loss of lock must come with an item whose Flag_valid_symbol_output is false (hip %d, reference %d)", st.loss_item_hip,
        st.loss_item_ref);
    reftrk_destroy(ref);
}

void test_unusable_configurations()
{
    const std::string R = "Tracking";
    Props p{{"GNSS-SDR.internal_fs_sps", "4000000"}, {R + ".item_type", "cshort"}, {R + ".hip_device", "0"}, {R + ".hip_register_input_buffer", "false"}};
    auto cfg = make_config(p);
    GpsL1CaDllPllTrackingHip hip(cfg.get(), R, 1, 1);
    EXPECT(hip.item_size() == 0, "cshort items: item_size must be 0 (gnss_block_factory.cc:1048-1052), got %zu", hip.item_size());
    Props q{{"GNSS-SDR.internal_fs_sps", "4000000"}, {R + ".hip_device", "63"}};
    auto cfg2 = make_config(q);
    GpsL1CaDllPllTrackingHip hip2(cfg2.get(), R, 1, 1);
    EXPECT(hip2.item_size() == 0, "absent device: item_size must be 0, got %zu", hip2.item_size());
}
#include "test_tracking_runtime_cases.inc"
}  // namespace

int main(int argc, char** argv)
{
    const bool conf_only = argc > 1 && std::string(argv[1]) == "conf";
    const std::string mode = argc > 1 ? argv[1] : "";
    const int host_threads = std::max(2, std::min(16, static_cast<int>(std::thread::hardware_concurrency())));
    if (mode == "bench")  // test_tracking_adapters bench [channels fs periods periods_per_call [seconds]]: the drop-in throughput leg (bench.py's `dropin`)
        {
            if (gsh_device_count() < 1)
                {
                    std::printf("no HIP device\n");
                    return 2;
                }
            const int ch = argc > 2 ? std::atoi(argv[2]) : 32;
            const long fs = argc > 3 ? std::atol(argv[3]) : 25000000L;
            const int periods = argc > 4 ? std::atoi(argv[4]) : 400;
            const int ppc = argc > 5 ? std::atoi(argv[5]) : 10;
            const double seconds = argc > 6 ? std::atof(argv[6]) : 0.0;  // > 0: replay a 2 400-period stream for that long (periods = the cap)
            return bench_dropin(ch, fs, periods, ppc, host_threads, seconds);
        }
    if (mode == "runtime")  // only the runtime cases (what the CPU suite runs against the fake engine): [channels periods periods_per_call]
        {
            if (gsh_device_count() < 1)
                {
                    std::printf("no HIP device\n");
                    return 2;
                }
            const int ch = argc > 2 ? std::atoi(argv[2]) : 8;
            const int periods = argc > 3 ? std::atoi(argv[3]) : 2400;
            const int ppc = argc > 4 ? std::atoi(argv[4]) : 4;
            test_shared_ring_follows_the_rf_chain();
            test_restart_on_the_same_block();
            test_many_periods_per_call();
            test_dump_tow_and_time_tags(1);
            test_dump_tow_and_time_tags(10);
            test_shared_runtime_threads(ch, periods, ppc, host_threads);
            test_shared_runtime_threads(std::max(2, ch / 2), 500, 1, host_threads);
            std::printf(fails == 0 ? "TRACKING RUNTIME OK\n" : "%d failure(s)\n", fails);
            return fails == 0 ? 0 : 1;
        }
    if (mode == "loss")  // only the loss-of-lock case
        {
            if (gsh_device_count() < 1) return 2;
            test_loss_of_lock_on_noise();
            std::printf(fails == 0 ? "LOSS OF LOCK OK\n" : "%d failure(s)\n", fails);
            return fails == 0 ? 0 : 1;
        }
    test_conf_mapping();
    if (conf_only)
        {
            std::printf(fails == 0 ? "TRACKING CONF OK\n" : "%d failure(s)\n", fails);
            return fails == 0 ? 0 : 1;
        }
    if (gsh_device_count() < 1)
        {
            std::printf("no HIP device: only the configuration mapping was checked\n");
            return fails == 0 ? 2 : 1;
        }
    test_unusable_configurations();
    test_gps_l1_trajectory();
    test_galileo_e1_pilot_trajectory();
    test_other_signal_trajectories();
    test_handover_with_the_read_pointer_behind_the_stamp();
    test_loss_of_lock_on_noise();
    test_shared_ring_follows_the_rf_chain();
    test_restart_on_the_same_block();
    test_many_periods_per_call();
    test_dump_tow_and_time_tags(1);
    test_dump_tow_and_time_tags(10);
    test_shared_runtime_threads(32, 2400, 10, host_threads);
    test_shared_runtime_threads(8, 500, 1, host_threads);
    std::printf(fails == 0 ? "TRACKING ADAPTERS OK\n" : "%d failure(s)\n", fails);
    return fails == 0 ? 0 : 1;
}
