/*!
 * \file dll_pll_conf_hip.cc
 * \brief Dll_Pll_Conf -> gsh_trk_conf; see dll_pll_conf_hip.h.  Pinned by tests/host/test_adapters.cc against the reference block's own
 *        constructor (oracle/_ref/libgnsssdr_ref_trk.so, reftrk_get_conf) field by field.
 */
#include "dll_pll_conf_hip.h"
#include "Beidou_B1I.h"
#include "Beidou_B3I.h"
#include "GLONASS_L1_L2_CA.h"
#include "GPS_L1_CA.h"
#include "GPS_L2C.h"
#include "GPS_L5.h"
#include "Galileo_E1.h"
#include "Galileo_E5a.h"
#include "Galileo_E5b.h"
#include "Galileo_E6.h"
#include "beidou_b1i_signal_replica.h"
#include "beidou_b3i_signal_replica.h"
#include "galileo_e1_signal_replica.h"
#include "galileo_e5_signal_replica.h"
#include "galileo_e6_signal_replica.h"
#include "glonass_l1_signal_replica.h"
#include "glonass_l2_signal_replica.h"
#include "gps_l2c_signal_replica.h"
#include "gps_l5_signal_replica.h"
#include "gps_sdr_signal_replica.h"
#include "qzss.h"
#include "qzss_signal_replica.h"
#include "tracking_discriminators.h"
#include <algorithm>
#include <array>
#include <complex>
#include <cstring>

namespace
{
void set_code_string(uint8_t (&dst)[GSH_MAX_SECONDARY], int32_t* len, const std::string& s)
{
    std::memset(dst, 0, sizeof(dst));
    const size_t n = std::min<size_t>(s.size(), sizeof(dst));
    std::memcpy(dst, s.data(), n);
    *len = static_cast<int32_t>(n);
}
}  // namespace

bool hip_fill_trk_conf(const Dll_Pll_Conf& p, gsh_trk_conf* c, Hip_Trk_Signal* sig, std::string* why)
{
    std::memset(c, 0, sizeof(*c));
    *sig = Hip_Trk_Signal{};
    const std::string signal_type(p.signal);
    sig->signal_type = signal_type;
    bool track_pilot = p.track_pilot;
    float spc = p.spc, slope = p.slope, y_intercept = p.y_intercept;
    std::string secondary, data_secondary;
    bool has_secondary = false, veml = false, no_histogram = false;
    // ---- per-signal constants, trk.cc:196-596
    if (p.system == 'G' && signal_type == "1C")
        {
            sig->system_name = "GPS";
            c->signal_carrier_freq = GPS_L1_FREQ_HZ;
            c->code_chip_rate = GPS_L1_CA_CODE_RATE_CPS;
            c->code_length_chips = static_cast<uint32_t>(GPS_L1_CA_CODE_LENGTH_CHIPS);
            c->code_samples_per_chip = 1;
            sig->correlation_length_ms = 1;
            track_pilot = false;  // no pilot component (trk.cc:214)
            slope = 1.0F;
            spc = p.early_late_space_chips;
            y_intercept = 1.0F;
            secondary = GPS_CA_PREAMBLE_SYMBOLS_STR;  // the bit-transition pattern that gives bit synchronisation (trk.cc:219-222)
            c->symbols_per_bit = GPS_CA_TELEMETRY_SYMBOLS_PER_BIT;
        }
    else if (p.system == 'G' && signal_type == "L5")
        {
            sig->system_name = "GPS";
            c->signal_carrier_freq = GPS_L5_FREQ_HZ;
            c->code_chip_rate = GPS_L5I_CODE_RATE_CPS;
            c->code_length_chips = static_cast<uint32_t>(GPS_L5I_CODE_LENGTH_CHIPS);
            c->code_samples_per_chip = 1;
            c->symbols_per_bit = GPS_L5_SAMPLES_PER_SYMBOL;
            sig->correlation_length_ms = 1;
            has_secondary = true;
            slope = 1.0F;
            spc = p.early_late_space_chips;
            y_intercept = 1.0F;
            if (track_pilot)
                {
                    secondary = GPS_L5Q_NH_CODE_STR;
                    data_secondary = GPS_L5I_NH_CODE_STR;
                }
            else
                {
                    secondary = GPS_L5I_NH_CODE_STR;
                    sig->interchange_iq = true;
                }
        }
    else if (p.system == 'E' && signal_type == "1B")
        {
            sig->system_name = "Galileo";
            c->signal_carrier_freq = GALILEO_E1_FREQ_HZ;
            c->code_chip_rate = GALILEO_E1_CODE_CHIP_RATE_CPS;
            c->code_length_chips = static_cast<uint32_t>(GALILEO_E1_B_CODE_LENGTH_CHIPS);
            c->code_samples_per_chip = 2;  // sinBOC(1,1) replica, 2 samples per chip (trk.cc:289)
            c->symbols_per_bit = 1;
            sig->correlation_length_ms = 4;
            veml = true;
            spc = p.early_late_space_chips;
            slope = static_cast<float>(-CalculateSlopeAbs(&SinBocCorrelationFunction<1, 1>, spc));
            y_intercept = static_cast<float>(GetYInterceptAbs(&SinBocCorrelationFunction<1, 1>, spc));
            if (track_pilot)
                {
                    has_secondary = true;
                    secondary = GALILEO_E1_C_SECONDARY_CODE;
                }
        }
    else if (p.system == 'E' && signal_type == "5X")
        {
            sig->system_name = "Galileo";
            c->signal_carrier_freq = GALILEO_E5A_FREQ_HZ;
            c->code_chip_rate = GALILEO_E5A_CODE_CHIP_RATE_CPS;
            c->code_length_chips = static_cast<uint32_t>(GALILEO_E5A_CODE_LENGTH_CHIPS);
            c->code_samples_per_chip = 1;
            c->symbols_per_bit = 20;
            sig->correlation_length_ms = 1;
            has_secondary = true;
            slope = 1.0F;
            spc = p.early_late_space_chips;
            y_intercept = 1.0F;
            if (track_pilot)
                {
                    // the pilot's 100-chip secondary code depends on the PRN: set at start_tracking (trk.cc:857); length known here
                    secondary = std::string(GALILEO_E5A_Q_SECONDARY_CODE_LENGTH, '0');
                    data_secondary = GALILEO_E5A_I_SECONDARY_CODE;
                    sig->interchange_iq = true;
                    sig->per_prn_secondary = true;
                }
            else
                {
                    secondary = GALILEO_E5A_I_SECONDARY_CODE;
                }
        }
    else if (p.system == 'G' && signal_type == "2S")
        {
            sig->system_name = "GPS";
            c->signal_carrier_freq = GPS_L2_FREQ_HZ;
            c->code_chip_rate = GPS_L2_M_CODE_RATE_CPS;
            c->code_length_chips = static_cast<uint32_t>(GPS_L2_M_CODE_LENGTH_CHIPS);
            c->symbols_per_bit = GPS_L2_SAMPLES_PER_SYMBOL;  // 1 tracking symbol (20 ms) per telemetry bit
            sig->correlation_length_ms = 20;
            c->code_samples_per_chip = 2;  // the CM code with the CL slots zeroed (trk.cc:236)
            track_pilot = false;
            slope = 1.0F;
            spc = p.early_late_space_chips;
            y_intercept = 1.0F;
        }
    else if (p.system == 'E' && signal_type == "7X")
        {
            sig->system_name = "Galileo";
            c->signal_carrier_freq = GALILEO_E5B_FREQ_HZ;
            c->code_chip_rate = GALILEO_E5B_CODE_CHIP_RATE_CPS;
            c->code_length_chips = static_cast<uint32_t>(GALILEO_E5B_CODE_LENGTH_CHIPS);
            c->code_samples_per_chip = 1;
            c->symbols_per_bit = 4;
            sig->correlation_length_ms = 1;
            has_secondary = true;
            slope = 1.0F;
            spc = p.early_late_space_chips;
            y_intercept = 1.0F;
            if (track_pilot)
                {
                    secondary = std::string(GALILEO_E5B_Q_SECONDARY_CODE_LENGTH, '0');  // per PRN, set at start_tracking (trk.cc:887)
                    data_secondary = GALILEO_E5B_I_SECONDARY_CODE;
                    sig->interchange_iq = true;
                    sig->per_prn_secondary = true;
                }
            else
                {
                    secondary = GALILEO_E5B_I_SECONDARY_CODE;
                }
        }
    else if (p.system == 'E' && signal_type == "E6")
        {
            sig->system_name = "Galileo";
            c->signal_carrier_freq = GALILEO_E6_FREQ_HZ;
            c->code_chip_rate = GALILEO_E6_B_CODE_CHIP_RATE_CPS;
            c->code_length_chips = static_cast<uint32_t>(GALILEO_E6_B_CODE_LENGTH_CHIPS);
            c->code_samples_per_chip = 1;
            c->symbols_per_bit = 1;
            sig->correlation_length_ms = 1;
            slope = 1.0F;
            spc = p.early_late_space_chips;
            y_intercept = 1.0F;
            if (track_pilot)
                {
                    has_secondary = true;
                    secondary = std::string(GALILEO_E6_C_SECONDARY_CODE_LENGTH_CHIPS, '0');  // per PRN (trk.cc:912)
                    sig->per_prn_secondary = true;
                }
        }
    else if (p.system == 'C' && (signal_type == "B1" || signal_type == "B3"))
        {
            const bool b1 = signal_type == "B1";
            sig->system_name = "Beidou";
            c->signal_carrier_freq = b1 ? BEIDOU_B1I_FREQ_HZ : BEIDOU_B3I_FREQ_HZ;
            c->code_chip_rate = b1 ? BEIDOU_B1I_CODE_RATE_CPS : BEIDOU_B3I_CODE_RATE_CPS;
            c->code_length_chips = static_cast<uint32_t>(b1 ? BEIDOU_B1I_CODE_LENGTH_CHIPS : BEIDOU_B3I_CODE_LENGTH_CHIPS);
            c->symbols_per_bit = b1 ? BEIDOU_B1I_TELEMETRY_SYMBOLS_PER_BIT : BEIDOU_B3I_TELEMETRY_SYMBOLS_PER_BIT;
            c->code_samples_per_chip = 1;
            sig->correlation_length_ms = 1;
            has_secondary = b1;  // trk.cc:420 (B1: true) / :441 (B3: false) -- what the constructor leaves; start_tracking sets it per PRN
            track_pilot = false;
            slope = 1.0F;
            spc = p.early_late_space_chips;
            y_intercept = 1.0F;
            secondary = b1 ? BEIDOU_B1I_SECONDARY_CODE_STR : BEIDOU_B3I_SECONDARY_CODE_STR;
            data_secondary = secondary;
            sig->per_prn_secondary = true;  // GEO satellites use the preamble instead (trk.cc:930-990)
        }
    else if (p.system == 'R' && (signal_type == "1G" || signal_type == "2G"))
        {
            const bool l1 = signal_type == "1G";
            sig->system_name = "Glonass";
            c->signal_carrier_freq = l1 ? GLONASS_L1_CA_FREQ_HZ : GLONASS_L2_CA_FREQ_HZ;
            c->code_chip_rate = l1 ? GLONASS_L1_CA_CODE_RATE_CPS : GLONASS_L2_CA_CODE_RATE_CPS;
            c->code_length_chips = static_cast<uint32_t>(l1 ? GLONASS_L1_CA_CODE_LENGTH_CHIPS : GLONASS_L2_CA_CODE_LENGTH_CHIPS);
            c->symbols_per_bit = GLONASS_GNAV_TELEMETRY_SYMBOLS_PER_BIT;
            c->code_samples_per_chip = 1;
            sig->correlation_length_ms = 1;
            track_pilot = false;
            slope = 1.0F;
            spc = p.early_late_space_chips;
            y_intercept = 1.0F;
            secondary = GLONASS_GNAV_PREAMBLE_STR;
            sig->per_prn_secondary = true;  // the FDMA channel offset comes with the satellite (trk.cc:996-1003)
            no_histogram = true;            // Manchester coding (trk.cc:1389)
        }
    else if (p.system == 'J' && signal_type == "J1")
        {
            sig->system_name = "QZSS";
            c->signal_carrier_freq = QZSS_L1_FREQ_HZ;
            c->code_chip_rate = QZSS_L1_CHIP_RATE;
            c->code_length_chips = static_cast<uint32_t>(QZSS_L1_CODE_LENGTH);
            c->code_samples_per_chip = 1;
            sig->correlation_length_ms = 1;
            track_pilot = false;
            slope = 1.0F;
            spc = p.early_late_space_chips;
            y_intercept = 1.0F;
            secondary = QZSS_CA_PREAMBLE_SYMBOLS_STR;
            c->symbols_per_bit = QZSS_CA_TELEMETRY_SYMBOLS_PER_BIT;
        }
    else if (p.system == 'J' && signal_type == "J5")
        {
            sig->system_name = "QZSS";
            c->signal_carrier_freq = QZSS_L5_FREQ_HZ;
            c->code_chip_rate = QZSS_L5_CHIP_RATE;
            c->code_length_chips = static_cast<uint32_t>(QZSS_L5_CODE_LENGTH);
            c->code_samples_per_chip = 1;
            c->symbols_per_bit = QZSS_L5_SAMPLES_PER_SYMBOL;
            sig->correlation_length_ms = 1;
            has_secondary = true;
            slope = 1.0F;
            spc = p.early_late_space_chips;
            y_intercept = 1.0F;
            if (track_pilot)
                {
                    secondary = QZSS_L5Q_NH_CODE_STR;
                    data_secondary = QZSS_L5I_NH_CODE_STR;
                }
            else
                {
                    secondary = QZSS_L5I_NH_CODE_STR;
                    sig->interchange_iq = true;
                }
        }
    else
        {
            if (why) *why = std::string("signal ") + p.system + "/" + signal_type + " is not one dll_pll_veml_tracking knows (trk.cc:196-596)";
            return false;
        }
    if (p.item_type != "gr_complex")
        {
            if (why) *why = p.item_type + ": unknown tracking item type";  // as the reference adapters (gps_l1_ca_dll_pll_tracking.cc:100-105)
            return false;
        }
    // ---- the rest of Dll_Pll_Conf, as the block reads it
    c->fs_in = p.fs_in;
    c->cfo_frequency_hz = 0.0;
    c->vector_length = p.vector_length;
    c->veml = veml ? 1 : 0;
    c->track_pilot = track_pilot ? 1 : 0;
    c->early_late_space_chips = p.early_late_space_chips;
    c->very_early_late_space_chips = p.very_early_late_space_chips;
    c->pll_bw_hz = p.pll_bw_hz;
    c->dll_bw_hz = p.dll_bw_hz;
    c->fll_bw_hz = p.fll_bw_hz;
    c->pll_filter_order = p.pll_filter_order;
    c->dll_filter_order = p.dll_filter_order;
    c->enable_fll_pull_in = p.enable_fll_pull_in ? 1 : 0;
    c->enable_fll_steady_state = p.enable_fll_steady_state ? 1 : 0;
    c->carrier_aiding = p.carrier_aiding ? 1 : 0;
    c->cloop = 1;  // d_cloop(true); start_tracking sets it again (trk.cc:1072)
    c->pull_in_time_s = p.pull_in_time_s;
    c->spc = spc;
    c->slope = slope;
    c->y_intercept = y_intercept;
    c->enable_lock_detectors = 1;
    c->cn0_samples = p.cn0_samples;
    c->cn0_min = p.cn0_min;
    c->max_code_lock_fail = p.max_code_lock_fail;
    c->max_carrier_lock_fail = p.max_carrier_lock_fail;
    c->cn0_smoother_samples = p.cn0_smoother_samples;
    c->carrier_lock_test_smoother_samples = p.carrier_lock_test_smoother_samples;
    c->cn0_smoother_alpha = p.cn0_smoother_alpha;
    c->carrier_lock_test_smoother_alpha = p.carrier_lock_test_smoother_alpha;
    c->carrier_lock_th = p.carrier_lock_th;
    c->enable_symbol_sync = 1;
    c->has_secondary = has_secondary ? 1 : 0;
    set_code_string(c->secondary_code, &c->secondary_code_length, secondary);
    set_code_string(c->data_secondary_code, &c->data_secondary_code_length, data_secondary);
    c->extend_correlation_symbols = p.extend_correlation_symbols > 1 ? p.extend_correlation_symbols : 1;  // trk.cc:650-658
    c->pll_bw_narrow_hz = p.pll_bw_narrow_hz;
    c->dll_bw_narrow_hz = p.dll_bw_narrow_hz;
    c->early_late_space_narrow_chips = p.early_late_space_narrow_chips;
    c->very_early_late_space_narrow_chips = p.very_early_late_space_narrow_chips;
    // configure_bit_synchronizer, trk.cc:1387-1406, runs at the END of start_tracking (trk.cc:1077), i.e. on what start_tracking has set for the satellite:
    // hip_make_tracking_codes evaluates it again; this is the value for a satellite that changes nothing
    c->use_histogram_bit_sync = (!has_secondary && c->symbols_per_bit > 1 && !no_histogram) ? 1 : 0;
    c->bs_min_events_for_lock = p.bs_min_events_for_lock;
    c->bs_stable_best_required = p.bs_stable_best_required;
    c->bs_use_phase_dot_detector = p.bs_use_phase_dot_detector ? 1 : 0;
    c->bs_min_prompt_mag = p.bs_min_prompt_mag;
    c->bs_dominance_ratio = p.bs_dominance_ratio;
    c->high_dyn = p.high_dyn ? 1 : 0;
    c->smoother_length = p.smoother_length;
    c->enable_bit_sync_time_limit = 1;  // trk.cc:2000-2007: the block applies the fail-safe unconditionally
    c->bit_synchronization_time_limit_s = p.bit_synchronization_time_limit_s;
    c->enable_doppler_correction = p.enable_doppler_correction ? 1 : 0;  // trk.cc:1326-1346
    return true;
}

bool hip_make_tracking_codes(const Hip_Trk_Signal& sig, gsh_trk_conf* conf, uint32_t prn, const char signal[3], std::vector<float>* code,
    std::vector<float>* data_code, std::string* why)
{
    const size_t n = static_cast<size_t>(conf->code_length_chips) * conf->code_samples_per_chip;
    code->assign(n, 0.0F);
    data_code->clear();
    if (conf->track_pilot) data_code->assign(n, 0.0F);
    const std::string& st = sig.signal_type;
    if (sig.system_name == "GPS" && st == "1C")
        {
            gps_l1_ca_code_gen_float(*code, static_cast<int32_t>(prn), 0);  // trk.cc:812-815
        }
    else if (sig.system_name == "GPS" && st == "L5")
        {
            if (conf->track_pilot)
                {
                    gps_l5q_code_gen_float(*code, prn);  // trk.cc:822-829
                    gps_l5i_code_gen_float(*data_code, prn);
                }
            else
                gps_l5i_code_gen_float(*code, prn);
        }
    else if (sig.system_name == "Galileo" && st == "1B")
        {
            std::array<char, 3> sig_{{signal[0], signal[1], '\0'}};
            if (conf->track_pilot)
                {
                    const std::array<char, 3> pilot_signal = {{'1', 'C', '\0'}};  // trk.cc:836-843
                    galileo_e1_code_gen_sinboc11_float(*code, pilot_signal, prn);
                    galileo_e1_code_gen_sinboc11_float(*data_code, sig_, prn);
                }
            else
                galileo_e1_code_gen_sinboc11_float(*code, sig_, prn);
        }
    else if (sig.system_name == "Galileo" && st == "5X")
        {
            std::vector<std::complex<float>> aux(conf->code_length_chips);  // trk.cc:851-875
            const std::array<char, 3> signal_type_ = {{'5', 'X', '\0'}};
            galileo_e5_a_code_gen_complex_primary(aux, static_cast<int32_t>(prn), signal_type_);
            if (conf->track_pilot)
                {
                    if (prn < 1 || prn > static_cast<uint32_t>(GALILEO_E5A_NUMBER_OF_CODES))
                        {
                            if (why) *why = "Galileo E5a pilot: no secondary code for PRN " + std::to_string(prn);
                            return false;
                        }
                    const std::string sec = GALILEO_E5A_Q_SECONDARY_CODE[prn - 1];
                    std::memset(conf->secondary_code, 0, sizeof(conf->secondary_code));
                    std::memcpy(conf->secondary_code, sec.data(), std::min<size_t>(sec.size(), sizeof(conf->secondary_code)));
                    conf->secondary_code_length = static_cast<int32_t>(std::min<size_t>(sec.size(), sizeof(conf->secondary_code)));
                    for (size_t i = 0; i < aux.size(); i++)
                        {
                            (*code)[i] = aux[i].imag();
                            (*data_code)[i] = aux[i].real();
                        }
                }
            else
                for (size_t i = 0; i < aux.size(); i++) (*code)[i] = aux[i].real();
        }
    else if (sig.system_name == "GPS" && st == "2S")
        {
            gps_l2c_m_code_gen_float_cl_zeroed(*code, prn);  // trk.cc:816-819
        }
    else if (sig.system_name == "Galileo" && st == "7X")
        {
            std::vector<std::complex<float>> aux(conf->code_length_chips);  // trk.cc:877-901
            const std::array<char, 3> signal_type_ = {{'7', 'X', '\0'}};
            galileo_e5_b_code_gen_complex_primary(aux, static_cast<int32_t>(prn), signal_type_);
            if (conf->track_pilot)
                {
                    if (prn < 1 || prn > static_cast<uint32_t>(GALILEO_E5B_NUMBER_OF_CODES))
                        {
                            if (why) *why = "Galileo E5b pilot: no secondary code for PRN " + std::to_string(prn);
                            return false;
                        }
                    set_code_string(conf->secondary_code, &conf->secondary_code_length, GALILEO_E5B_Q_SECONDARY_CODE[prn - 1]);
                    for (size_t i = 0; i < aux.size(); i++)
                        {
                            (*code)[i] = aux[i].imag();
                            (*data_code)[i] = aux[i].real();
                        }
                }
            else
                for (size_t i = 0; i < aux.size(); i++) (*code)[i] = aux[i].real();
        }
    else if (sig.system_name == "Galileo" && st == "E6")
        {
            if (conf->track_pilot)  // trk.cc:908-921
                {
                    set_code_string(conf->secondary_code, &conf->secondary_code_length, galileo_e6_c_secondary_code(static_cast<int32_t>(prn)));
                    galileo_e6_b_code_gen_float_primary(*data_code, prn);
                    galileo_e6_c_code_gen_float_primary(*code, prn);
                }
            else
                galileo_e6_b_code_gen_float_primary(*code, prn);
        }
    else if (sig.system_name == "Beidou" && (st == "B1" || st == "B3"))
        {
            const bool b1 = st == "B1";
            if (b1)
                beidou_b1i_code_gen_float(*code, static_cast<int32_t>(prn), 0);
            else
                beidou_b3i_code_gen_float(*code, static_cast<int32_t>(prn), 0);
            // GEO satellites carry D2 messages: 2 symbols per bit, no NH code -- the preamble takes the secondary code's place (trk.cc:930-990)
            if ((prn > 0 && prn < 6) || (prn > 58))
                {
                    conf->symbols_per_bit = b1 ? BEIDOU_B1I_GEO_TELEMETRY_SYMBOLS_PER_BIT : BEIDOU_B3I_GEO_TELEMETRY_SYMBOLS_PER_BIT;
                    conf->has_secondary = 0;
                    set_code_string(conf->secondary_code, &conf->secondary_code_length, b1 ? BEIDOU_B1I_GEO_PREAMBLE_SYMBOLS_STR : BEIDOU_B3I_GEO_PREAMBLE_SYMBOLS_STR);
                    set_code_string(conf->data_secondary_code, &conf->data_secondary_code_length, std::string());
                    conf->extend_correlation_symbols = std::min(conf->extend_correlation_symbols, conf->symbols_per_bit);
                }
            else
                {
                    conf->symbols_per_bit = b1 ? BEIDOU_B1I_TELEMETRY_SYMBOLS_PER_BIT : BEIDOU_B3I_TELEMETRY_SYMBOLS_PER_BIT;
                    conf->has_secondary = 1;
                    const std::string nh = b1 ? BEIDOU_B1I_SECONDARY_CODE_STR : BEIDOU_B3I_SECONDARY_CODE_STR;
                    set_code_string(conf->secondary_code, &conf->secondary_code_length, nh);
                    set_code_string(conf->data_secondary_code, &conf->data_secondary_code_length, nh);
                }
        }
    else if (sig.system_name == "Glonass")
        {
            if (GLONASS_PRN.find(prn) == GLONASS_PRN.end())
                {
                    if (why) *why = "Glonass: no frequency channel for slot " + std::to_string(prn);  // the reference's .at() throws here
                    return false;
                }
            if (st == "1G")  // trk.cc:994-1003
                {
                    glonass_l1_ca_code_gen_float(*code, 0);
                    conf->cfo_frequency_hz = DFRQ1_GLO * GLONASS_PRN.at(prn);
                }
            else
                {
                    glonass_l2_ca_code_gen_float(*code, 0);
                    conf->cfo_frequency_hz = DFRQ2_GLO * GLONASS_PRN.at(prn);
                }
            conf->symbols_per_bit = GLONASS_GNAV_TELEMETRY_SYMBOLS_PER_BIT;
            conf->has_secondary = 0;
            set_code_string(conf->secondary_code, &conf->secondary_code_length, GLONASS_GNAV_PREAMBLE_STR);
            set_code_string(conf->data_secondary_code, &conf->data_secondary_code_length, std::string());
            conf->extend_correlation_symbols = std::min(conf->extend_correlation_symbols, static_cast<int32_t>(GLONASS_GNAV_TELEMETRY_SYMBOLS_PER_BIT));
        }
    else if (sig.system_name == "QZSS" && st == "J1")
        {
            qzss_l1_code_gen_float(*code, prn);
        }
    else if (sig.system_name == "QZSS" && st == "J5")
        {
            if (conf->track_pilot)
                {
                    qzss_l5q_code_gen_float(*code, prn);
                    qzss_l5i_code_gen_float(*data_code, prn);
                }
            else
                qzss_l5i_code_gen_float(*code, prn);
        }
    else
        {
            if (why) *why = "no replica generator for " + sig.system_name + " " + st;
            return false;
        }
    // configure_bit_synchronizer (trk.cc:1077, 1387-1389) on the satellite's settings; Glonass uses Manchester coding
    conf->use_histogram_bit_sync = (!conf->has_secondary && conf->symbols_per_bit > 1 && sig.system_name != "Glonass") ? 1 : 0;
    return true;
}
