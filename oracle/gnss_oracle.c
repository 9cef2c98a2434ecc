/*
 * TEST INFRASTRUCTURE ONLY -- see gnss_oracle.h.  CPU restatement of the
 * gnss-sdr tracking-correlator arithmetic, written from the behaviour of the
 * reference (citations per function), pinned against oracle/_ref.
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fcx-limited-range).
 * The FP flags matter: every float32 operation below must round exactly once,
 * in the order written, for the chip indices to be bit-exact with the
 * reference's _generic protokernels.
 */
#include "gnss_oracle.h"
#if defined(__FAST_MATH__) || defined(__FMA__)
#error "oracle must be built without -ffast-math / -mfma (see oracle/Makefile FPFLAGS)"
#endif
#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* GPS L1 C/A code, gps_sdr_signal_replica.cc:24-100                          */
/* ------------------------------------------------------------------------- */

#define CA_LEN 1023

/* G2 code-phase delays in chips: PRN 1..32 (IS-GPS-200), then SBAS PRN 120..138.
 * Same table as gps_sdr_signal_replica.cc:40-43. */
static const int ca_g2_delay[51] = {
    5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258,
    469, 470, 471, 472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862,
    145, 175, 52, 21, 237, 235, 886, 657, 634, 762, 355, 1012, 176, 603, 130, 359, 595, 68, 386};

static int ca_chips(int8_t* chips, int prn, unsigned int chip_shift)
{
    int slot;
    if (prn >= 120 && prn <= 138)
        slot = prn - 88; /* gps_sdr_signal_replica.cc:46-49 */
    else
        slot = prn - 1;
    if (slot < 0 || slot >= 51) return -1;

    /* Two 10-stage LFSRs, all-ones start.  Bit 0 of the word is stage 10 (the
     * output), bit 9 is stage 1 (where the feedback enters) -- the layout the
     * reference uses at :56-78.  G1: stages 3,10.  G2: stages 2,3,6,8,9,10. */
    uint8_t g1[CA_LEN], g2[CA_LEN];
    unsigned r1 = 0x3FF, r2 = 0x3FF;
    for (int i = 0; i < CA_LEN; i++)
        {
            g1[i] = r1 & 1u;
            g2[i] = r2 & 1u;
            unsigned f1 = ((r1 >> 7) ^ r1) & 1u;
            unsigned f2 = ((r2 >> 8) ^ (r2 >> 7) ^ (r2 >> 4) ^ (r2 >> 2) ^ (r2 >> 1) ^ r2) & 1u;
            r1 = (r1 >> 1) | (f1 << 9);
            r2 = (r2 >> 1) | (f2 << 9);
        }
    unsigned d = (CA_LEN - (unsigned)ca_g2_delay[slot] + chip_shift) % CA_LEN; /* :81-83 */
    for (unsigned i = 0; i < CA_LEN; i++)
        {
            unsigned bit = g1[(i + chip_shift) % CA_LEN] ^ g2[d]; /* :88 */
            chips[i] = bit ? 1 : -1;                               /* :89-96, true -> +1 */
            d = (d + 1) % CA_LEN;
        }
    return 0;
}

int oracle_gps_l1_ca_code_gen_float(float* dest, int prn, unsigned int chip_shift)
{
    int8_t c[CA_LEN];
    if (ca_chips(c, prn, chip_shift) != 0) return -1;
    for (int i = 0; i < CA_LEN; i++) dest[i] = (float)c[i];
    return 0;
}

/* gps_sdr_signal_replica.cc:135-173.  One code period sampled at fs, chip picked
 * by floor(ts*i/tc) in float32, last sample forced to chip 1022; the chip value
 * goes to the imaginary part (:117-131). */
int oracle_gps_l1_ca_code_gen_complex_sampled(float* dest_iq, unsigned int prn, int fs, unsigned int chip_shift)
{
    int8_t c[CA_LEN];
    if (ca_chips(c, (int)prn, chip_shift) != 0) return -1;
    const float tc = 1.0F / 1023000.0F;
    const int samples_per_code = (int)((double)fs / (1023000.0 / 1023.0));
    const float ts = 1.0F / (float)fs;
    for (int i = 0; i < samples_per_code; i++)
        {
            int k = (int)floorf(ts * (float)i / tc);
            if (i == samples_per_code - 1) k = CA_LEN - 1;
            dest_iq[2 * i] = 0.0F;
            dest_iq[2 * i + 1] = (float)c[k];
        }
    return samples_per_code;
}

/* ------------------------------------------------------------------------- */
/* code NCO: chip index of sample n for one tap                               */
/* ------------------------------------------------------------------------- */

/* K/volk_gnsssdr_32f_xn_resampler_32f_xn.h:75-76 -- wrap a possibly negative
 * index into [0, L). */
static inline int wrap_chip(int k, unsigned int code_len)
{
    if (k < 0) k += (int)code_len * (abs(k) / code_len + 1);
    return (int)((unsigned int)k % code_len);
}

/* K/...resampler_32f_xn.h:73 : (step*(float)n + shift) - rem, each op rounded to float32 */
static inline int chip_std(float step, unsigned int n, float shift, float rem, unsigned int code_len)
{
    const float a = step * (float)n;
    const float b = a + shift;
    const float c = b - rem;
    return wrap_chip((int)floor(c), code_len);
}

/* K/...high_dynamics_resampler_32f_xn.h:76 : ((step*n + rate*(float)(n*n)) + shift0) - rem,
 * n*n in unsigned int (wraps for n >= 65536, as in the reference) */
static inline int chip_hd(float step, float rate, unsigned int n, float shift0, float rem, unsigned int code_len)
{
    const float a = step * (float)n;
    const float q = rate * (float)(n * n);
    const float b = a + q;
    const float c = b + shift0;
    const float d = c - rem;
    return wrap_chip((int)floor(d), code_len);
}

/* K/...high_dynamics_resampler_32f_xn.h:82-90: taps 1..T-1 are tap 0 rotated left by
 * shift_samples, accumulated in unsigned int from (int)round(delta_shift/step). */
static void hd_tap_rotations(unsigned int* rot, const float* shifts, float step, int n_taps)
{
    unsigned int acc = 0;
    rot[0] = 0;
    for (int t = 1; t < n_taps; t++)
        {
            acc += (int)round((shifts[t] - shifts[t - 1]) / step);
            rot[t] = acc;
        }
}

void oracle_code_indices(int32_t* idx, float rem, float step, float rate, const float* shifts,
    unsigned int code_len, int n_taps, unsigned int n, int high_dyn)
{
    if (!high_dyn)
        {
            for (int t = 0; t < n_taps; t++)
                for (unsigned int i = 0; i < n; i++) idx[(size_t)t * n + i] = chip_std(step, i, shifts[t], rem, code_len);
            return;
        }
    unsigned int rot[64];
    hd_tap_rotations(rot, shifts, step, n_taps);
    for (unsigned int i = 0; i < n; i++) idx[i] = chip_hd(step, rate, i, shifts[0], rem, code_len);
    for (int t = 1; t < n_taps; t++)
        for (unsigned int i = 0; i < n; i++)
            {
                /* rotation by rot[t] (the reference requires rot <= n; larger is UB there) */
                unsigned int src = i + rot[t];
                if (src >= n) src -= n;
                idx[(size_t)t * n + i] = idx[src % n];
            }
}

void oracle_resampler(float** result, const float* code, float rem, float step, const float* shifts,
    unsigned int code_len, int n_taps, unsigned int n)
{
    for (int t = 0; t < n_taps; t++)
        for (unsigned int i = 0; i < n; i++) result[t][i] = code[chip_std(step, i, shifts[t], rem, code_len)];
}

void oracle_hd_resampler(float** result, const float* code, float rem, float step, float rate,
    const float* shifts, unsigned int code_len, int n_taps, unsigned int n)
{
    unsigned int rot[64];
    hd_tap_rotations(rot, shifts, step, n_taps);
    for (unsigned int i = 0; i < n; i++) result[0][i] = code[chip_hd(step, rate, i, shifts[0], rem, code_len)];
    for (int t = 1; t < n_taps; t++)
        {
            const unsigned int r = rot[t];
            memcpy(result[t], result[0] + r, (size_t)(n - r) * sizeof(float));
            memcpy(result[t] + (n - r), result[0], (size_t)r * sizeof(float));
        }
}

/* ------------------------------------------------------------------------- */
/* carrier wipe-off + multiply-accumulate, float32, reference order           */
/* ------------------------------------------------------------------------- */

typedef float _Complex cf32;

/* K/volk_gnsssdr_32fc_32f_rotator_dot_prod_32fc_xn.h:66-98 */
static void rotator_dot_prod(cf32* result, const cf32* in, cf32 phase_inc, cf32* phase, float* const* taps,
    int n_taps, unsigned int n)
{
    for (int t = 0; t < n_taps; t++) result[t] = 0.0F;
    cf32 ph = *phase;
    for (unsigned int i = 0; i < n; i++)
        {
            const cf32 wiped = in[i] * ph; /* product formed BEFORE the renormalisation (:77) */
            if ((i & 255u) == 0) ph /= hypotf(crealf(ph), cimagf(ph)); /* :80-89 */
            ph *= phase_inc;                                            /* :91 */
            for (int t = 0; t < n_taps; t++) result[t] += wiped * taps[t][i]; /* :92-96 */
        }
    *phase = ph;
}

/* K/volk_gnsssdr_32fc_32f_high_dynamic_rotator_dot_prod_32fc_xn.h:68-109 */
static void hd_rotator_dot_prod(cf32* result, const cf32* in, cf32 phase_inc, cf32 phase_inc_rate, cf32* phase,
    float* const* taps, int n_taps, unsigned int n)
{
    for (int t = 0; t < n_taps; t++) result[t] = 0.0F;
    cf32 ph = *phase;
    cf32 ph_doppler = *phase; /* never renormalised in the reference (:73,:97) */
    for (unsigned int i = 0; i < n; i++)
        {
            if ((i & 255u) == 0) ph /= hypotf(crealf(ph), cimagf(ph)); /* :86-93 */
            const cf32 wiped = in[i] * ph;                              /* :94 */
            ph_doppler *= phase_inc;                                    /* :95 */
            cf32 ph_rate = cpowf(phase_inc_rate, (float)(i * i) + 0.0F * I); /* :100, i*i unsigned */
            ph_rate /= hypotf(crealf(ph_rate), cimagf(ph_rate));             /* :101 */
            ph = ph_doppler * ph_rate;                                       /* :103 */
            for (int t = 0; t < n_taps; t++) result[t] += wiped * taps[t][i]; /* :105-108 */
        }
    *phase = ph;
}

int oracle_mcorr(const float* code, int code_len, const float* shifts, int n_taps, const float* in_iq, int n,
    float rem_carr, float phase_step, float phase_rate_step, float rem_code, float code_step,
    float code_rate_step, int high_dyn, float* out_iq)
{
    if (n_taps <= 0 || n_taps > 64 || n <= 0 || code_len <= 0) return -1;
    float* taps[64];
    float* block = (float*)malloc((size_t)n_taps * (size_t)n * sizeof(float));
    if (!block) return -2;
    for (int t = 0; t < n_taps; t++) taps[t] = block + (size_t)t * n;

    /* update_local_code, cpu_multicorrelator_real_codes.cc:75-100 */
    if (high_dyn)
        oracle_hd_resampler(taps, code, rem_code, code_step, code_rate_step, shifts, (unsigned)code_len, n_taps, (unsigned)n);
    else
        oracle_resampler(taps, code, rem_code, code_step, shifts, (unsigned)code_len, n_taps, (unsigned)n);

    /* cpu_multicorrelator_real_codes.cc:113-124 */
    cf32 phase = cosf(rem_carr) + (-sinf(rem_carr)) * I;
    const cf32 inc = cexpf(0.0F + (-phase_step) * I);
    cf32 res[64];
    /* high_dyn: 0 = standard kernels; 1 = 7-argument call with the flag set (hd resampler + hd rotator,
     * mcorr.cc:117-120); 2 = 6-argument overload with the flag set (hd resampler via update_local_code,
     * standard rotator, mcorr.cc:137-142) */
    if (high_dyn == 1)
        {
            const cf32 inc_rate = cexpf(0.0F + (-phase_rate_step) * I);
            hd_rotator_dot_prod(res, (const cf32*)in_iq, inc, inc_rate, &phase, taps, n_taps, (unsigned)n);
        }
    else
        rotator_dot_prod(res, (const cf32*)in_iq, inc, &phase, taps, n_taps, (unsigned)n);
    for (int t = 0; t < n_taps; t++)
        {
            out_iq[2 * t] = crealf(res[t]);
            out_iq[2 * t + 1] = cimagf(res[t]);
        }
    free(block);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* the 16-bit family: Cpu_Multicorrelator_16sc (the tail)      */
/* ------------------------------------------------------------------------- */

/* the two phasors as T/cpu_multicorrelator_16sc.cc:89-93 forms them: (cos rem, -sin rem) with libm's float functions and std::exp of the
 * imaginary float argument (libstdc++ -> cexpf).  out4: phase0 re, im, increment re, im */
void oracle_mcorr16_phasors(float rem_carr, float phase_step, float* out4)
{
    out4[0] = cosf(rem_carr);
    out4[1] = -sinf(rem_carr);
    const cf32 inc = cexpf(0.0F + (-phase_step) * I);
    out4[2] = crealf(inc);
    out4[3] = cimagf(inc);
}

static inline int16_t sat16(int32_t v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/*
 * K/volk_gnsssdr_16ic_xn_resampler_16ic_xn.h:60-78 (chip selection: the float32 expression of the 32f resampler, floor taken in double of a
 * float -- the same integer) followed by K/volk_gnsssdr_16ic_x2_rotator_dot_prod_16ic_xn.h:66-102 (generic), phasors handed in:
 * every sample is rotated in float32, rounded to the nearest even integer and kept as int16 (low 16 bits), the phasor is renormalised at
 * n = 0, 256, 512 ... AFTER it has been used and BEFORE it is advanced, the product with the code sample is the low 16 bits of the complex
 * integer product, and every tap's two sums saturate at each addition (so the result depends on the order: n ascending).
 * in / code / out: interleaved int16 I, Q.
 */
int oracle_mcorr16_phasor(const int16_t* code_iq, int code_len, const float* shifts, int n_taps, const int16_t* in_iq, int n,
    float ph_re, float ph_im, float inc_re, float inc_im, float rem_code, float code_step, int16_t* out_iq)
{
    if (n_taps <= 0 || n_taps > 64 || n < 0 || code_len <= 0) return -1;
    int16_t acc_r[64], acc_i[64];
    for (int t = 0; t < n_taps; t++) acc_r[t] = acc_i[t] = 0;
    for (int i = 0; i < n; i++)
        {
            const float xr = (float)in_iq[2 * i], xi = (float)in_iq[2 * i + 1];
            const float yr = xr * ph_re - xi * ph_im, yi = xr * ph_im + xi * ph_re; /* :77 */
            const int32_t wr = (int16_t)(int32_t)rintf(yr), wi = (int16_t)(int32_t)rintf(yi); /* :78 */
            if ((i & 255) == 0) /* :81-90 */
                {
                    const float h = hypotf(ph_re, ph_im);
                    ph_re /= h;
                    ph_im /= h;
                }
            {
                const float nr = ph_re * inc_re - ph_im * inc_im, ni = ph_re * inc_im + ph_im * inc_re; /* :92 */
                ph_re = nr;
                ph_im = ni;
            }
            for (int t = 0; t < n_taps; t++)
                {
                    const int k = chip_std(code_step, (unsigned)i, shifts[t], rem_code, (unsigned)code_len);
                    const int32_t cr = code_iq[2 * k], ci = code_iq[2 * k + 1];
                    const int16_t pr = (int16_t)(wr * cr - wi * ci), pi = (int16_t)(wr * ci + wi * cr); /* :95 */
                    acc_r[t] = sat16((int32_t)acc_r[t] + pr); /* :97 */
                    acc_i[t] = sat16((int32_t)acc_i[t] + pi);
                }
        }
    for (int t = 0; t < n_taps; t++)
        {
            out_iq[2 * t] = acc_r[t];
            out_iq[2 * t + 1] = acc_i[t];
        }
    return 0;
}

/* One Cpu_Multicorrelator_16sc::Carrier_wipeoff_multicorrelator_resampler call, T/cpu_multicorrelator_16sc.cc:80-96 */
int oracle_mcorr16(const int16_t* code_iq, int code_len, const float* shifts, int n_taps, const int16_t* in_iq, int n,
    float rem_carr, float phase_step, float rem_code, float code_step, int16_t* out_iq)
{
    float p[4];
    oracle_mcorr16_phasors(rem_carr, phase_step, p);
    return oracle_mcorr16_phasor(code_iq, code_len, shifts, n_taps, in_iq, n, p[0], p[1], p[2], p[3], rem_code, code_step, out_iq);
}

/* float64 truth; chip selection is the reference's float32 expression, the rest exact. */
int oracle_mcorr_f64(const float* code, int code_len, const float* shifts, int n_taps, const float* in_iq, int n,
    float rem_carr, float phase_step, float phase_rate_step, float rem_code, float code_step,
    float code_rate_step, int high_dyn, double* out_iq, double* sum_abs)
{
    if (n_taps <= 0 || n_taps > 64 || n <= 0 || code_len <= 0) return -1;
    int32_t* idx = (int32_t*)malloc((size_t)n_taps * (size_t)n * sizeof(int32_t));
    if (!idx) return -2;
    oracle_code_indices(idx, rem_code, code_step, code_rate_step, shifts, (unsigned)code_len, n_taps, (unsigned)n, high_dyn);
    double accr[64], acci[64];
    for (int t = 0; t < n_taps; t++) accr[t] = acci[t] = 0.0;
    double sabs = 0.0;
    for (int i = 0; i < n; i++)
        {
            double ph = (double)rem_carr + (double)i * (double)phase_step;
            if (high_dyn == 1 && i > 0)
                {
                    /* sample i is rotated with inc_rate^((i-1)^2): the rate factor computed in
                     * iteration i-1 from (i-1)*(i-1) (unsigned, then float) is applied to sample i
                     * (K/...high_dynamic_rotator...:94-103) */
                    const unsigned int m = (unsigned int)(i - 1);
                    ph += (double)phase_rate_step * (double)(float)(m * m);
                }
            const double c = cos(ph), s = -sin(ph);
            const double xr = in_iq[2 * i], xi = in_iq[2 * i + 1];
            const double wr = xr * c - xi * s, wi = xr * s + xi * c;
            sabs += sqrt(xr * xr + xi * xi);
            for (int t = 0; t < n_taps; t++)
                {
                    const double cv = code[idx[(size_t)t * n + i]];
                    accr[t] += wr * cv;
                    acci[t] += wi * cv;
                }
        }
    for (int t = 0; t < n_taps; t++)
        {
            out_iq[2 * t] = accr[t];
            out_iq[2 * t + 1] = acci[t];
        }
    if (sum_abs) *sum_abs = sabs;
    free(idx);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* small acquisition-side kernels                                             */
/* ------------------------------------------------------------------------- */

void oracle_sincos(float* out_iq, float phase_inc, float* phase, unsigned int n)
{
    float p = *phase;
    for (unsigned int i = 0; i < n; i++)
        {
            out_iq[2 * i] = cosf(p);
            out_iq[2 * i + 1] = sinf(p);
            p += phase_inc;
        }
    *phase = p;
}

void oracle_index_max(uint32_t* target, const float* src, uint32_t n)
{
    if (n == 0) return;
    float best = src[0];
    uint32_t at = 0;
    for (uint32_t i = 1; i < n; i++)
        if (src[i] > best)
            {
                best = src[i];
                at = i;
            }
    *target = at;
}

/* ------------------------------------------------------------------------- */
/* timing harness                                                             */
/* ------------------------------------------------------------------------- */

typedef struct
{
    const float* codes;
    int code_len;
    const float* shifts;
    int n_taps;
    const float* stream;
    long stream_len;
    int n;
    int n_channels;
    int epochs;
    const float* params;
    float* out;
    int next;
    pthread_mutex_t mu;
} time_job;

#define TIME_BLOCK 16

/* work items are (channel, block of TIME_BLOCK epochs) so that every host core has work */
static void* time_worker(void* arg)
{
    time_job* j = (time_job*)arg;
    const int blocks = (j->epochs + TIME_BLOCK - 1) / TIME_BLOCK;
    const int n_items = j->n_channels * blocks;
    const long span = j->stream_len - j->n;
    for (;;)
        {
            pthread_mutex_lock(&j->mu);
            const int item = j->next++;
            pthread_mutex_unlock(&j->mu);
            if (item >= n_items) break;
            const int ch = item / blocks;
            const int e0 = (item % blocks) * TIME_BLOCK;
            const int e1 = e0 + TIME_BLOCK < j->epochs ? e0 + TIME_BLOCK : j->epochs;
            const float* p = j->params + 6 * ch;
            float out[128];
            for (int e = e0; e < e1; e++)
                {
                    const long pos = ((long)p[4] + (long)e * j->n) % (span > 0 ? span : 1);
                    oracle_mcorr(j->codes + (size_t)ch * j->code_len, j->code_len, j->shifts, j->n_taps,
                        j->stream + 2 * pos, j->n, p[0], p[1], 0.0F, p[2], p[3], 0.0F, 0, out);
                }
            if (e1 == j->epochs) memcpy(j->out + (size_t)ch * 2 * j->n_taps, out, sizeof(float) * 2 * j->n_taps);
        }
    return NULL;
}

double oracle_mcorr_time(const float* codes, int code_len, const float* shifts, int n_taps,
    const float* stream_iq, long stream_len, int n, int n_channels, int epochs, int n_threads,
    const float* params, float* out_iq)
{
    time_job j = {codes, code_len, shifts, n_taps, stream_iq, stream_len, n, n_channels, epochs, params, out_iq, 0,
        PTHREAD_MUTEX_INITIALIZER};
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, time_worker, &j);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
