// Tracking dump file in the block's own binary layout (dll_pll_veml_tracking::log_data, trk.cc:1599-1702): 108 bytes per logged period, the
// format save_matfile (trk.cc:1705-1716) and utils/matlab/libs/dll_pll_veml_read_tracking_dump.m parse (the Python reader under utils/python
// predates the TOW / week-number fields).
// Host-only code: turns gsh_trk_epoch records into that file so that the reference's plotting / analysis tooling works on a
// device-closed loop unchanged.
#include "gsh_internal.h"
#include <cmath>
#include <cstdio>

extern "C"
{
    int gsh_trk_write_dump(const char* path, int append, const gsh_trk_conf* conf, uint32_t prn, const gsh_trk_epoch* records, int n_records,
        const uint64_t* tow_ms, const uint32_t* wn)
    {
        GSH_REQUIRE(path != nullptr && conf != nullptr, "null argument");
        GSH_REQUIRE(n_records >= 0 && (n_records == 0 || records != nullptr), "bad record array");
        FILE* f = std::fopen(path, append ? "ab" : "wb");
        if (f == nullptr) return gsh::set_error(GSH_ERR_INVALID, "cannot open %s for writing", path);
        const int veml = conf->veml ? 1 : 0;
        const int p = veml ? 2 : 1;  // index of the prompt in corr[]
        auto mag = [](const float* c) { return std::hypot(c[0], c[1]); };  // std::abs<float>(gr_complex), trk.cc:1626-1636
        for (int i = 0; i < n_records; i++)
            {
                const gsh_trk_epoch& r = records[i];
                if (r.flags & 2) continue;  // loss of lock: the reference does not call log_data in that period (trk.cc:2009-2014)
                if ((r.state == 3 || r.state == 4) && !(r.symbol_flags & 1)) continue;  // coherent integration / narrow tracking log once per telemetry symbol (trk.cc:2164-2166, 2212-2218)
                struct __attribute__((packed)) Rec
                {
                    float ve, e, pr, l, vl, prompt_i, prompt_q;
                    uint64_t prn_start_sample;
                    float acc_carrier_phase_rad, carrier_doppler_hz, carrier_doppler_rate_hz_s, code_freq_chips, code_freq_rate, carr_error_hz,
                        carr_error_filt_hz, code_error_chips, code_error_filt_chips, cn0_db_hz, carrier_lock_test, rem_code_phase_samples;
                    double sample_stamp;
                    uint32_t prn;
                    uint64_t tow;
                    uint32_t wn;
                } o;
                static_assert(sizeof(Rec) == 108, "log_data writes 108 bytes per period (trk.cc:1705-1710)");
                o.ve = veml ? mag(&r.accu[0]) : 0.0f;                       // |d_VE_accu| .. |d_VL_accu|, trk.cc:1624-1636
                o.e = mag(&r.accu[2 * (p - 1)]);
                o.pr = mag(&r.accu[2 * p]);
                o.l = mag(&r.accu[2 * (p + 1)]);
                o.vl = veml ? mag(&r.accu[8]) : 0.0f;
                o.prompt_i = conf->track_pilot ? r.prompt_data[0] : r.corr[2 * p];      // trk.cc:1614-1623
                o.prompt_q = conf->track_pilot ? r.prompt_data[1] : r.corr[2 * p + 1];
                o.prn_start_sample = r.sample_counter + static_cast<uint64_t>(r.prn_length_samples);  // nitems_read + d_current_prn_length_samples, :1650
                o.acc_carrier_phase_rad = static_cast<float>(r.acc_carrier_phase_rad);
                o.carrier_doppler_hz = static_cast<float>(r.carrier_doppler_hz);
                o.carrier_doppler_rate_hz_s = static_cast<float>(r.carrier_phase_rate_step_rad * conf->fs_in * conf->fs_in / 6.283185307179586);  // :1659 (TWO_PI of MATH_CONSTANTS.h differs in the 13th digit: below float precision)
                o.code_freq_chips = static_cast<float>(r.code_freq_chips);
                o.code_freq_rate = static_cast<float>(r.code_phase_rate_step_chips * conf->fs_in * conf->fs_in);  // :1664
                o.carr_error_hz = static_cast<float>(r.carr_phase_error_hz);
                o.carr_error_filt_hz = static_cast<float>(r.carr_error_filt_hz);
                o.code_error_chips = static_cast<float>(r.code_error_chips);
                o.code_error_filt_chips = static_cast<float>(r.code_error_filt_chips);
                o.cn0_db_hz = r.cn0_db_hz;
                o.carrier_lock_test = static_cast<float>(r.carrier_lock_test);
                o.rem_code_phase_samples = static_cast<float>(r.rem_code_phase_samples);
                o.sample_stamp = static_cast<double>(r.sample_counter + static_cast<uint64_t>(r.prn_length_samples));  // :1684
                o.prn = prn;
                o.tow = tow_ms != nullptr ? tow_ms[i] : 0ull;  // d_tow_from_telemetry_ms, d_wn_from_telemetry (trk.cc:1921-1935)
                o.wn = wn != nullptr ? wn[i] : 0u;
                if (std::fwrite(&o, sizeof(o), 1, f) != 1)
                    {
                        std::fclose(f);
                        return gsh::set_error(GSH_ERR_INVALID, "short write to %s", path);
                    }
            }
        std::fclose(f);
        return GSH_OK;
    }
}
