/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the DLL/PLL loop closure of gnss-sdr's
 * dll_pll_veml_tracking, the checker of gnss-sdr_amd/csrc/tracking_loop.hip.
 *
 * Parity status: PINNED for the discriminators that use libm (fll_diff_atan, pll_cloop_two_quadrant_atan,
 * dll_nc_e_minus_l_normalized, dll_nc_vemlp_normalized) and for both loop filters -- exact equality with the
 * reference's own objects compiled into oracle/_ref (tests/test_oracle_loop.py) and with the known answers of the
 * reference's unit tests (tracking_loop_filter_test.cc:22-206, discriminator_test.cc:35-174).
 * UNPINNED for pll_four_quadrant_atan: the reference calls gr::fast_atan2f (GNU Radio, un-vendored, version unpinned,
 * a 255-entry table approximation); atan2f is used here and in the oracle/_ref shim.
 *
 * Citations are relative to /root/reference/:  trk.cc = src/algorithms/tracking/gnuradio_blocks/dll_pll_veml_tracking.cc,
 * T/ = src/algorithms/tracking/libs/.
 */
#include "gnss_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* the reference's pi is the GNSS ICD value, MATH_CONSTANTS.h:47-49 */
#define ORA_PI 3.1415926535898
#define ORA_HALF_PI (ORA_PI / 2.0)
#define ORA_TWO_PI (2.0 * ORA_PI)

/* T/tracking_discriminators.cc:27-41 */
static double phase_unwrap(double phase_rad)
{
    if (phase_rad >= ORA_HALF_PI) return phase_rad - ORA_PI;
    if (phase_rad <= -ORA_HALF_PI) return phase_rad + ORA_PI;
    return phase_rad;
}

/* T/tracking_discriminators.cc:68-76: atan(Q2/I2) - atan(Q1/I1); std::atan of a float IS the float overload, so the
 * two arctangents and their difference are float32, only then widened; NaN (0/0) -> 0 */
double oracle_fll_diff_atan(float p1re, float p1im, float p2re, float p2im, double t1, double t2)
{
    double d = atanf(p2im / p2re) - atanf(p1im / p1re);
    if (isnan(d)) d = 0.0;
    return phase_unwrap(d) / (t2 - t1);
}

/* T/tracking_discriminators.cc:86-89 (gr::fast_atan2f -> atan2f, see the header of this file) */
double oracle_pll_four_quadrant_atan(float re, float im) { return (double)atan2f(im, re); }

/* T/tracking_discriminators.cc:99-106 */
double oracle_pll_cloop_two_quadrant_atan(float re, float im)
{
    if (re != 0.0) return (double)atanf(im / re);  /* std::atan(float): the float overload */
    return 0.0;
}

static double cabs_f(float re, float im) { return (double)hypotf(re, im); }

/* T/tracking_discriminators.cc:117-127: std::abs(gr_complex) is a float hypot, widened to double */
double oracle_dll_nc_e_minus_l_normalized(float ere, float eim, float lre, float lim, float spc, float slope, float y_intercept)
{
    const double pe = cabs_f(ere, eim);
    const double pl = cabs_f(lre, lim);
    const double s = pe + pl;
    if (s == 0.0) return 0.0;
    return ((y_intercept - slope * spc) / slope) * (pe - pl) / s;
}

/* T/tracking_discriminators.cc:139-149: float sums of squares, float sqrt, widened to double */
double oracle_dll_nc_vemlp_normalized(float vere, float veim, float ere, float eim, float lre, float lim, float vlre, float vlim)
{
    const double early = sqrtf(vere * vere + veim * veim + ere * ere + eim * eim);
    const double late = sqrtf(lre * lre + lim * lim + vlre * vlre + vlim * vlim);
    const double s = early + late;
    if (s == 0.0) return 0.0;
    return (early - late) / s;
}

/* ---- Tracking_loop_filter (T/tracking_loop_filter.cc): bilinear-transform loop of order 1..3 ---------------- */
#define HIST 4 /* MAX_LOOP_HISTORY_LENGTH, :31 */

/* update_coefficients, T/tracking_loop_filter.cc:101-196 (Kaplan & Hegarty table 5.6).  The float / double mix of
 * every expression is the reference's: products of float gains stay float, divisions by the literal 2.0 widen. */
void oracle_loop_filter_design(oracle_loop_filter* f, float update_interval, float noise_bandwidth, int order, int include_last_integrator)
{
    memset(f, 0, sizeof(*f));
    f->order = order;
    f->last_integrator = include_last_integrator;
    const float T = update_interval;
    const float zeta = 1.0F / sqrtf(2.0F);
    float g1, g2, g3, wn;
    switch (order)
        {
        case 1:
            wn = noise_bandwidth * 4.0F;
            g1 = wn;
            if (include_last_integrator)
                {
                    f->n_in = 2;
                    f->in_c[0] = (float)(g1 * T / 2.0);
                    f->in_c[1] = (float)(g1 * T / 2.0);
                    f->n_out = 1;
                    f->out_c[0] = 1.0F;
                }
            else
                {
                    f->n_in = 1;
                    f->in_c[0] = g1;
                    f->n_out = 0;
                }
            break;
        case 2:
            wn = noise_bandwidth * (8.0F * zeta) / (4.0F * zeta * zeta + 1.0F);
            g1 = wn * wn;
            g2 = wn * 2.0F * zeta;
            if (include_last_integrator)
                {
                    f->n_in = 3;
                    f->in_c[0] = (float)(T / 2.0 * (g1 * T / 2.0 + g2));
                    f->in_c[1] = (float)(T * T / 2.0 * g1);
                    f->in_c[2] = (float)(T / 2.0 * (g1 * T / 2.0 - g2));
                    f->n_out = 2;
                    f->out_c[0] = 2.0F;
                    f->out_c[1] = -1.0F;
                }
            else
                {
                    f->n_in = 2;
                    f->in_c[0] = (float)(g1 * T / 2.0 + g2);
                    f->in_c[1] = (float)(g1 * T / 2.0 - g2);
                    f->n_out = 1;
                    f->out_c[0] = 1.0F;
                }
            break;
        default:
            {
                wn = noise_bandwidth / 0.7845F;
                const float a3 = 1.1;
                const float b3 = 2.4;
                g1 = wn * wn * wn;
                g2 = a3 * wn * wn;
                g3 = b3 * wn;
                if (include_last_integrator)
                    {
                        f->n_in = 4;
                        f->in_c[0] = (float)(T / 2.0 * (g3 + T / 2.0 * (g2 + T / 2.0 * g1)));
                        f->in_c[1] = (float)(T / 2.0 * (-g3 + T / 2.0 * (g2 + 3.0 * T / 2.0 * g1)));
                        f->in_c[2] = (float)(T / 2.0 * (-g3 - T / 2.0 * (g2 - 3.0 * T / 2.0 * g1)));
                        f->in_c[3] = (float)(T / 2.0 * (g3 - T / 2.0 * (g2 - T / 2.0 * g1)));
                        f->n_out = 3;
                        f->out_c[0] = 3.0F;
                        f->out_c[1] = -3.0F;
                        f->out_c[2] = 1.0F;
                    }
                else
                    {
                        f->n_in = 3;
                        f->in_c[0] = (float)(g3 + T / 2.0 * (g2 + T / 2.0 * g1));
                        f->in_c[1] = (float)(g1 * T * T / 2.0 - 2.0 * g3);
                        f->in_c[2] = (float)(g3 + T / 2.0 * (-g2 + T / 2.0 * g1));
                        f->n_out = 2;
                        f->out_c[0] = 2.0F;
                        f->out_c[1] = -1.0F;
                    }
            }
            break;
        }
    oracle_loop_filter_initialize(f, 0.0F);
}

/* initialize, T/tracking_loop_filter.cc:266-271 */
void oracle_loop_filter_initialize(oracle_loop_filter* f, float initial_output)
{
    for (int i = 0; i < HIST; i++)
        {
            f->in_h[i] = 0.0F;
            f->out_h[i] = initial_output;
        }
    f->idx = HIST - 1;
}

/* apply, T/tracking_loop_filter.cc:63-98: old outputs first, then move the ring index, then the inputs */
float oracle_loop_filter_apply(oracle_loop_filter* f, float x)
{
    float r = 0.0F;
    for (int i = 0; i < f->n_out; i++) r += f->out_c[i] * f->out_h[(f->idx + i) % HIST];
    f->idx--;
    if (f->idx < 0) f->idx += HIST;
    f->in_h[f->idx] = x;
    for (int i = 0; i < f->n_in; i++) r += f->in_c[i] * f->in_h[(f->idx + i) % HIST];
    f->out_h[f->idx] = r;
    return r;
}

/* ---- Tracking_FLL_PLL_filter (T/tracking_FLL_PLL_filter.cc): Kaplan 2nd ed. FLL-assisted PLL ------------------ */
/* set_params :23-54 */
void oracle_fll_pll_design(oracle_fll_pll_filter* f, float fll_bw_hz, float pll_bw_hz, int order)
{
    memset(f, 0, sizeof(*f));
    f->order = order;
    if (order == 3)
        {
            f->b3 = 2.400;
            f->a3 = 1.100;
            f->a2 = 1.414;
            f->w0p = pll_bw_hz / 0.7845F;
            f->w0p2 = f->w0p * f->w0p;
            f->w0p3 = f->w0p2 * f->w0p;
            f->w0f = fll_bw_hz / 0.53F;
            f->w0f2 = f->w0f * f->w0f;
        }
    else
        {
            f->a2 = 1.414;
            f->w0p = pll_bw_hz / 0.53F;
            f->w0p2 = f->w0p * f->w0p;
            f->w0f = fll_bw_hz / 0.25F;
        }
}

/* initialize :57-69 */
void oracle_fll_pll_initialize(oracle_fll_pll_filter* f, float acq_carrier_doppler_hz)
{
    if (f->order == 3)
        {
            f->x = 2.0F * acq_carrier_doppler_hz;
            f->w = 0;
        }
    else
        {
            f->w = acq_carrier_doppler_hz;
            f->x = 0;
        }
}

/* get_carrier_error :72-99 */
float oracle_fll_pll_carrier_error(oracle_fll_pll_filter* f, float fll_disc, float pll_disc, float t)
{
    float out;
    if (f->order == 3)
        {
            f->w = f->w + t * (f->w0p3 * pll_disc + f->w0f2 * fll_disc);
            f->x = f->x + t * (0.5F * f->w + f->a2 * f->w0f * fll_disc + f->a3 * f->w0p2 * pll_disc);
            out = 0.5F * f->x + f->b3 * f->w0p * pll_disc;
        }
    else
        {
            const float w_new = f->w + pll_disc * f->w0p2 * t + fll_disc * f->w0f * t;
            out = 0.5F * (w_new + f->w) + f->a2 * f->w0p * pll_disc;
            f->w = w_new;
        }
    return out;
}

/* ---- direct resampler: direct_resampler_conditioner_cc.cc, statement by statement ---------------------------- */
void oracle_direct_resampler_init(oracle_direct_resampler_t* r, double fs_in, double fs_out)  /* :39-61 */
{
    const double two_32 = 4294967296.0;
    r->fs_in = fs_in;
    r->fs_out = fs_out;
    r->phase = 0;
    r->lphase = 0;
    double v;
    if (fs_in >= fs_out)
        v = floor(two_32 * fs_out / fs_in);
    else
        v = floor(two_32 * fs_in / fs_out);
    r->phase_step = v >= two_32 ? 0u : (uint32_t)v; /* a ratio of one does not fit the uint32 cast: 0 = every sample passes */
}

int oracle_direct_resampler_work(oracle_direct_resampler_t* r, const float* in_iq, int n_in, float* out_iq, int noutput_items, int* consumed)  /* :72-129 */
{
    int lcv = 0, count = 0;
    const float* in = in_iq;
    if (r->fs_in >= r->fs_out)
        {
            while (lcv < noutput_items && count < n_in)   /* (the block trusts forecast; the bound keeps the checker inside its buffer) */
                {
                    if (r->phase <= r->lphase)
                        {
                            out_iq[2 * lcv] = in[0];
                            out_iq[2 * lcv + 1] = in[1];
                            lcv++;
                        }
                    r->lphase = r->phase;
                    r->phase += r->phase_step;
                    in += 2;
                    count++;
                }
        }
    else
        {
            while (lcv < noutput_items)
                {
                    const uint32_t lph = r->phase;
                    const uint32_t ph = r->phase + r->phase_step;
                    if (ph <= lph && count + 1 >= n_in) break; /* the next sample is not in this buffer yet */
                    r->lphase = lph;
                    r->phase = ph;
                    if (r->phase <= r->lphase)
                        {
                            in += 2;
                            count++;
                        }
                    out_iq[2 * lcv] = in[0];
                    out_iq[2 * lcv + 1] = in[1];
                    lcv++;
                }
        }
    *consumed = count < n_in ? count : n_in;
    return lcv;
}

/* ---- histogram bit synchroniser, T/bit_synchronizer.cc ---------------------------------------------------------- */
void oracle_bit_sync_init(oracle_bit_sync* b, int bins, int min_events_for_lock, int stable_best_required, double dominance_ratio,
    float min_prompt_mag, int use_phase_dot_detector)
{
    memset(b, 0, sizeof(*b));  /* reset(), :20-38 */
    b->bins = bins > 0 ? bins : 0;
    b->min_events_for_lock = min_events_for_lock;
    b->stable_best_required = stable_best_required;
    b->dominance_ratio = dominance_ratio;
    b->min_prompt_mag = min_prompt_mag;
    b->use_phase_dot_detector = use_phase_dot_detector;
    b->edge_phase = -1;
    b->last_sign = +1;
}

int oracle_bit_sync_update(oracle_bit_sync* b, float p_re, float p_im, int tracking_quality_ok)  /* :41-124 */
{
    const int N = b->bins;
    const int phase = (N > 0) ? (int)(b->epoch_count % N) : 0;
    ++b->epoch_count;  /* always advance: even if gated out the phase stays consistent */
    if (!tracking_quality_ok || (hypotf(p_re, p_im) < b->min_prompt_mag))
        {
            b->last_prompt[0] = p_re;
            b->last_prompt[1] = p_im;
            b->has_last_prompt = 1;
            return 0;
        }
    int edge_event = 0;
    if (b->use_phase_dot_detector)
        {
            if (b->has_last_prompt)
                {
                    /* dot = Re(Pk * conj(Pk-1)) in complex<float> arithmetic, widened afterwards */
                    const float dot_f = p_re * b->last_prompt[0] + p_im * b->last_prompt[1];
                    edge_event = ((double)dot_f < 0.0);
                }
            b->last_prompt[0] = p_re;
            b->last_prompt[1] = p_im;
            b->has_last_prompt = 1;
        }
    else
        {
            const int s = (p_re >= 0.0F) ? +1 : -1;
            if (b->has_last_sign) edge_event = (s != b->last_sign);
            b->last_sign = s;
            b->has_last_sign = 1;
        }
    if (edge_event && N > 0)
        {
            ++b->hist[phase];
            ++b->total_events;
        }
    if (!b->locked && (b->total_events >= b->min_events_for_lock))
        {
            int best_bin = 0, best_count = N > 0 ? b->hist[0] : 0;  /* best_bin_and_count, :149-161: first maximum */
            for (int i = 1; i < N; i++)
                if (b->hist[i] > best_count)
                    {
                        best_count = b->hist[i];
                        best_bin = i;
                    }
            const double ratio = (b->total_events > 0) ? ((double)best_count / (double)b->total_events) : 0.0;
            if (!b->has_last_best_bin || (best_bin != b->last_best_bin))
                {
                    b->last_best_bin = best_bin;
                    b->has_last_best_bin = 1;
                    b->stable_best_count = 1;
                }
            else
                {
                    ++b->stable_best_count;
                }
            if ((ratio >= b->dominance_ratio) && (b->stable_best_count >= b->stable_best_required))
                {
                    b->locked = 1;
                    b->edge_phase = best_bin;
                    return 1;
                }
        }
    return 0;
}

int oracle_bit_sync_epochs_until_next_edge(const oracle_bit_sync* b)  /* :164-182 */
{
    if (!b->locked || b->edge_phase < 0) return -1;
    const int B = b->bins;
    if (B <= 0) return -1;
    const int64_t k_now = b->epoch_count - 1;
    const int cur_phase = (int)(k_now % B);
    return (b->edge_phase - cur_phase + B) % B;
}

/* ---- lock detectors and C/N0 ---------------------------------------------------------------------------------
 * T/lock_detectors.cc:61-110: second / fourth moment estimator, everything float32, sequential sums */
float oracle_cn0_m2m4_estimator(const float* prompt_iq, int length, float coh_integration_time_s)
{
    float SNR_aux = 0.0F, Psig = 0.0F, m_2 = 0.0F, m_4 = 0.0F, aux;
    const float n = (float)length;
    if (length == 0 || coh_integration_time_s == 0.0) return -100.0F;
    for (int i = 0; i < length; i++)
        {
            const float re = prompt_iq[2 * i], im = prompt_iq[2 * i + 1];
            Psig += fabsf(re);
            aux = im * im + re * re;
            m_2 += aux;
            m_4 += (aux * aux);
        }
    Psig /= n;
    Psig = Psig * Psig;
    m_2 /= n;
    m_4 /= n;
    aux = sqrtf(2.0F * m_2 * m_2 - m_4);
    float denominator;
    if (isnan(aux))
        {
            denominator = m_2 - Psig;
            if (denominator == 0) return -100.0F;
            SNR_aux = Psig / denominator;
        }
    else
        {
            denominator = m_2 - aux;
            if (denominator == 0) return -100.0F;
            SNR_aux = aux / denominator;
        }
    if (SNR_aux == 0) return -100.0F;
    return 10.0F * log10f(SNR_aux) - 10.0F * log10f(coh_integration_time_s);
}

/* T/lock_detectors.cc:113-133 */
float oracle_carrier_lock_detector(const float* prompt_iq, int length)
{
    float tmp_sum_I = 0.0F, tmp_sum_Q = 0.0F;
    for (int i = 0; i < length; i++)
        {
            tmp_sum_I += prompt_iq[2 * i];
            tmp_sum_Q += prompt_iq[2 * i + 1];
        }
    const float NBP = tmp_sum_I * tmp_sum_I + tmp_sum_Q * tmp_sum_Q;
    const float NBD = tmp_sum_I * tmp_sum_I - tmp_sum_Q * tmp_sum_Q;
    if (NBP == 0) return 0.0F;
    return NBD / NBP;
}

/* Exponential_Smoother, T/exponential_smoother.cc:28-112.  The initialisation buffer is only ever summed front to back in
 * float (std::accumulate with a 0.0F seed, :93), which a running float sum reproduces exactly. */
void oracle_smoother_init(oracle_smoother* s, float alpha, int samples_for_initialization, float min_value, float offset)
{
    memset(s, 0, sizeof(*s));
    s->alpha = alpha < 0 ? 0 : (alpha > 1 ? 1 : alpha);  /* set_alpha :28-40 */
    s->one_minus_alpha = 1.0F - s->alpha;
    s->samples_for_initialization = samples_for_initialization <= 0 ? 1 : samples_for_initialization;  /* :49-58 */
    s->min_value = min_value;
    s->offset = offset;
    s->initializing = 1;
}

void oracle_smoother_reset(oracle_smoother* s)  /* :61-66 */
{
    s->initializing = 1;
    s->init_counter = 0;
    s->init_sum = 0.0F;
}

float oracle_smoother_smooth(oracle_smoother* s, float raw)  /* :83-112 */
{
    float smoothed;
    if (s->initializing)
        {
            s->init_counter++;
            smoothed = raw;
            s->init_sum += smoothed;
            if (s->init_counter == s->samples_for_initialization)
                {
                    s->old_value = s->init_sum / (float)s->init_counter;
                    if (s->old_value < (s->min_value + s->offset))
                        {
                            s->init_counter = 0; /* flush buffer and start again */
                            s->init_sum = 0.0F;
                        }
                    else
                        {
                            s->initializing = 0;
                        }
                }
        }
    else
        {
            smoothed = s->alpha * raw + s->one_minus_alpha * s->old_value;
            s->old_value = smoothed;
        }
    return smoothed;
}

/* trk.cc:676-692 (buffers and smoothers as the constructor sets them) */
void oracle_lock_init(oracle_lock_state* st, const oracle_trk_conf* c)
{
    memset(st, 0, sizeof(*st));
    const double code_period = (double)c->code_length_chips / c->code_chip_rate;
    int cn0_init = 200;  /* Exponential_Smoother default, T/exponential_smoother.h:66 */
    if (code_period > 0.0) cn0_init = c->cn0_smoother_samples / (int)(code_period * 1000.0);  /* trk.cc:683-686 */
    oracle_smoother_init(&st->cn0_smoother, c->cn0_smoother_alpha, cn0_init, 25.0F, 12.0F);  /* defaults of T/exponential_smoother.h:64-65 */
    oracle_smoother_init(&st->carrier_lock_test_smoother, c->carrier_lock_test_smoother_alpha, c->carrier_lock_test_smoother_samples, -1.0F, 0.0F);  /* :688-692 */
    st->carrier_lock_test = 1.0;  /* d_carrier_lock_test(1.0) in the constructor and in clear_tracking_vars (trk.cc:112, 1040) */
}

/* cn0_and_tracking_lock_status, trk.cc:1167-1224 */
int oracle_lock_status(oracle_lock_state* st, const oracle_trk_conf* c, float p_re, float p_im, double coh_integration_time_s, int pull_in_transitory)
{
    const int ns = c->cn0_samples;
    if (st->cn0_estimation_counter < ns)
        {
            st->prompt_buffer[2 * st->cn0_estimation_counter] = p_re;  /* fill buffer with prompt correlator output values */
            st->prompt_buffer[2 * st->cn0_estimation_counter + 1] = p_im;
            st->cn0_estimation_counter++;
            return 1;
        }
    st->prompt_buffer[2 * (st->cn0_estimation_counter % ns)] = p_re;
    st->prompt_buffer[2 * (st->cn0_estimation_counter % ns) + 1] = p_im;
    st->cn0_estimation_counter++;
    const float cn0_raw = oracle_cn0_m2m4_estimator(st->prompt_buffer, ns, (float)coh_integration_time_s);
    st->cn0_db_hz = oracle_smoother_smooth(&st->cn0_smoother, cn0_raw);
    /* carrier_lock_detector(d_Prompt_buffer.data(), 1): length ONE -- only the buffer's first entry is looked at (trk.cc:1184).
     * d_carrier_lock_test is a double: the double overload of smooth() narrows, smooths in float, widens (T/exponential_smoother.cc:75-80) */
    st->carrier_lock_test = (double)oracle_smoother_smooth(&st->carrier_lock_test_smoother, (float)(double)oracle_carrier_lock_detector(st->prompt_buffer, 1));
    if (!pull_in_transitory)
        {
            if (st->carrier_lock_test < c->carrier_lock_th)
                st->carrier_lock_fail_counter++;
            else if (st->carrier_lock_fail_counter > 0)
                st->carrier_lock_fail_counter--;
            if (st->cn0_db_hz < (float)c->cn0_min)
                st->code_lock_fail_counter++;
            else if (st->code_lock_fail_counter > 0)
                st->code_lock_fail_counter--;
        }
    if (st->carrier_lock_fail_counter > c->max_carrier_lock_fail || st->code_lock_fail_counter > c->max_code_lock_fail)
        {
            st->carrier_lock_fail_counter = 0;
            st->code_lock_fail_counter = 0;
            return 0;
        }
    return 1;
}

/* ---- the closed loop of one channel -------------------------------------------------------------------------
 * trk.cc state 2 (:1975-2001) per code period: do_correlation_step (:1232-1257) -> accumulators (:1978-1985)
 * -> run_dll_pll (:1260-1347) -> update_tracking_vars (:1409-1483) -> consume d_current_prn_length_samples (:2287),
 * initial conditions of start_tracking (:796-866) and of the pull-in state (:1949-1964).
 * With enable_lock_detectors: cn0_and_tracking_lock_status (:1167-1224) between the correlation and run_dll_pll, as state 2
 * orders them (:2008-2018); a loss of lock ends the channel (flags bit 1), the period's record carries the correlators only.
 * Not modelled (out of this row): bit / secondary-code synchronisation, extended integration,
 * the experimental Doppler correction (:1326-1346), high_dyn smoothing (:1425-1443).
 */
int oracle_trk_run(const oracle_trk_conf* c, const float* code, const float* data_code, int code_len, const float* stream_iq,
    uint64_t n_stream, uint64_t start_sample, uint64_t acq_sample_stamp, double acq_carrier_doppler_hz, int n_epochs,
    oracle_trk_epoch* rec)
{
    return oracle_trk_run_flags(c, code, data_code, code_len, stream_iq, n_stream, start_sample, acq_sample_stamp, acq_carrier_doppler_hz, n_epochs, rec, 0U);
}

/* trk.cc:1910-1917: d_pull_in_transitory is looked at by EVERY general_work call, the pull-in call (state 1) included, with the block's read pointer as it is THEN:
 *   pull_in_time_s < (nitems_read(0) - d_acq_sample_stamp) / (int)fs_in   (unsigned 64-bit)
 * -- a read pointer still behind the acquisition's stamp wraps the difference and ends the transitory before the first period. */
int oracle_pull_in_over(const oracle_trk_conf* c, uint64_t nitems_read, uint64_t acq_sample_stamp)
{
    if ((int)c->fs_in <= 0) return 0;
    return (uint64_t)c->pull_in_time_s < (nitems_read - acq_sample_stamp) / (uint64_t)((int)c->fs_in) ? 1 : 0;
}

/* flags bit 0 (ORACLE_TRK_PULL_IN_OVER): the transitory was already over at the pull-in call (oracle_pull_in_over) */
int oracle_trk_run_flags(const oracle_trk_conf* c, const float* code, const float* data_code, int code_len, const float* stream_iq,
    uint64_t n_stream, uint64_t start_sample, uint64_t acq_sample_stamp, double acq_carrier_doppler_hz, int n_epochs,
    oracle_trk_epoch* rec, unsigned flags)
{
    const int n_taps = c->veml ? 5 : 3;
    const int prompt = c->veml ? 2 : 1;
    float shifts[5] = {0, 0, 0, 0, 0};
    const float spcf = (float)c->code_samples_per_chip;
    if (c->veml)  /* trk.cc:829-835 */
        {
            shifts[0] = -c->very_early_late_space_chips * spcf;
            shifts[1] = -c->early_late_space_chips * spcf;
            shifts[3] = c->early_late_space_chips * spcf;
            shifts[4] = c->very_early_late_space_chips * spcf;
        }
    else
        {
            shifts[0] = -c->early_late_space_chips * spcf;
            shifts[2] = c->early_late_space_chips * spcf;
        }
    const float zero_shift[1] = {0.0F};
    const double code_period = (double)c->code_length_chips / c->code_chip_rate;  /* d_code_period */
    oracle_loop_filter dll;
    oracle_fll_pll_filter pll;
    oracle_loop_filter_design(&dll, (float)code_period, c->dll_bw_hz, c->dll_filter_order, 0);  /* trk.cc:604, 845-846 */
    oracle_fll_pll_design(&pll, c->fll_bw_hz, c->pll_bw_hz, c->pll_filter_order);                 /* trk.cc:605, 844 */
    oracle_fll_pll_initialize(&pll, (float)acq_carrier_doppler_hz);                               /* trk.cc:848 */
    oracle_loop_filter_initialize(&dll, 0.0F);                                                    /* trk.cc:849 */

    /* start_tracking :803-826 + pull-in :1956-1958 */
    double carrier_doppler_hz = acq_carrier_doppler_hz;
    double carrier_phase_step_rad = ORA_TWO_PI * (carrier_doppler_hz + c->cfo_frequency_hz) / c->fs_in;  /* trk.cc:801; Glonass :1003 */
    double code_freq_chips = c->code_chip_rate;
    double code_phase_step_chips = code_freq_chips / c->fs_in;
    double rem_code_phase_samples = 0.0, rem_code_phase_chips = 0.0, acc_carrier_phase_rad = 0.0;
    float rem_carr_phase_rad = 0.0F;
    float p_old_re = 0.0F, p_old_im = 0.0F;  /* d_P_accu_old */
    double corr_time = code_period;          /* d_current_correlation_time_s, trk.cc:841; extended integration stretches it (:2118) */
    uint64_t pos = start_sample;
    oracle_lock_state lock;
    int pull_in_latched = 1;  /* d_pull_in_transitory, cleared once (trk.cc:1910-1917) */
    /* a stamp beyond start_sample: the pull-in call's read pointer (<= start_sample) was below the stamp, the unsigned difference wrapped and the reference's latch
     * was released at that call for good -- never to come back when pos passes the stamp */
    if (acq_sample_stamp > start_sample) flags |= 1U;
    /* symbol synchronisation (trk.cc:2026-2104) and narrow tracking (state 4, :2197-2252) */
    int state = 2, cloop = c->cloop;
    float ring[2 * ORACLE_MAX_SECONDARY];  /* d_Prompt_circular_buffer: boost::circular_buffer of capacity d_secondary_code_length */
    int ring_count = 0, ring_head = 0;     /* ring_head: index of the oldest element once the buffer is full */
    int current_symbol = 0, current_data_symbol = 0, flag_pll_180 = 0, acc_phase_initialized = 0;
    float p_data_accu[2] = {0.0F, 0.0F};
    /* extended integration (states 3 / 4, trk.cc:2114-2149, 2156-2195, 2241-2251) */
    const int extend = (c->enable_symbol_sync && c->extend_correlation_symbols > 1) ? c->extend_correlation_symbols : 1;
    float accv[10] = {0};  /* d_VE_accu .. d_VL_accu */
    int ext_count = 0;     /* d_extend_correlation_symbols_count */
    float spc_now = c->spc;
    /* high dynamics: rate-of-change estimates of both NCO steps from two adjacent averages over smoother_length periods (trk.cc:1425-1443, 1458-1480) */
    const int hd = c->high_dyn ? 1 : 0;
    const int SL = (int)c->smoother_length;
    double carr_hist[2 * ORACLE_MAX_SMOOTHER][2], code_hist[2 * ORACLE_MAX_SMOOTHER][2];  /* boost::circular_buffer<pair<double,double>>(2 * smoother_length) */
    int carr_hist_n = 0, code_hist_n = 0;   /* size; once full the oldest sits at index (pushes % capacity) */
    long carr_pushes = 0, code_pushes = 0;
    double carrier_phase_rate_step_rad = 0.0, code_phase_rate_step_chips = 0.0;
    if (hd && (SL < 1 || SL > ORACLE_MAX_SMOOTHER)) return -1;
    /* histogram bit synchroniser (trk.cc:2046-2072); switched off after its first lock */
    oracle_bit_sync bs;
    /* experimental Doppler correction (trk.cc:1326-1346): d_dll_filt_history is a boost::circular_buffer<float>(1000) that is only ever filled and
     * cleared, and std::accumulate(begin, end, 0.0) adds its floats to a double in push order -- a running double sum is the same arithmetic */
    double dll_filt_sum = 0.0;
    int dll_filt_count = 0, corrected_doppler = 0;
    int use_hist = c->enable_symbol_sync && c->use_histogram_bit_sync && !c->has_secondary && c->symbols_per_bit > 1;
    int wait_for_bit_edge = 0;
    int64_t bit_sync_target_epoch = 0;
    if (use_hist)
        {
            if (c->symbols_per_bit > ORACLE_MAX_BITSYNC_BINS) return -1;
            oracle_bit_sync_init(&bs, c->symbols_per_bit, c->bs_min_events_for_lock, c->bs_stable_best_required, c->bs_dominance_ratio, c->bs_min_prompt_mag,
                c->bs_use_phase_dot_detector);
        }
    if (c->enable_symbol_sync && (c->secondary_code_length < 0 || c->secondary_code_length > ORACLE_MAX_SECONDARY ||
                                     c->data_secondary_code_length < 0 || c->data_secondary_code_length > ORACLE_MAX_SECONDARY))
        return -1;
    if (c->enable_lock_detectors)
        {
            if (c->cn0_samples < 1 || c->cn0_samples > ORACLE_MAX_CN0_SAMPLES) return -1;
            oracle_lock_init(&lock, c);
        }

    for (int e = 0; e < n_epochs; e++)
        {
            if (pos + c->vector_length > n_stream) return e;
            oracle_trk_epoch* r = &rec[e];
            memset(r, 0, sizeof(*r));
            /* trk.cc:1912-1915: pull-in ends once more than pull_in_time_s whole seconds have passed since acquisition */
            const int pull_in = !(flags & 1U) && !((uint64_t)c->pull_in_time_s < (pos - acq_sample_stamp) / (uint64_t)((int)c->fs_in));
            float out[16];
            /* do_correlation_step, trk.cc:1232-1257 (rate terms are 0 outside high_dyn) */
            const float rem_code = (float)rem_code_phase_chips * spcf;
            const float code_step = (float)code_phase_step_chips * spcf;
            const float code_rate = (float)code_phase_rate_step_chips * spcf;
            oracle_mcorr(code, code_len, shifts, n_taps, stream_iq + 2 * pos, (int)c->vector_length, rem_carr_phase_rad,
                (float)carrier_phase_step_rad, (float)carrier_phase_rate_step_rad, rem_code, code_step, code_rate, hd, out);
            memcpy(r->corr, out, sizeof(float) * 2 * n_taps);
            if (c->track_pilot && data_code)
                {
                    float pd[2];
                    oracle_mcorr(data_code, code_len, zero_shift, 1, stream_iq + 2 * pos, (int)c->vector_length, rem_carr_phase_rad,
                        (float)carrier_phase_step_rad, (float)carrier_phase_rate_step_rad, rem_code, code_step, code_rate, hd, pd);
                    r->prompt_data[0] = pd[0];
                    r->prompt_data[1] = pd[1];
                }
            r->state = state;
            if (state == 3 || state == 4)
                {
                    /* save_correlation_results, trk.cc:1486-1596, into accumulators that were zeroed at the end of the previous period
                     * (:2241-2246): the correlators enter the loop multiplied by the secondary code chip */
                    float sgn = 1.0F;
                    if (c->has_secondary)
                        {
                            sgn = c->secondary_code[current_symbol] == '0' ? 1.0F : -1.0F;
                            current_symbol = (current_symbol + 1) % c->secondary_code_length;
                        }
                    for (int t = 0; t < 2 * n_taps; t++)
                        {
                            accv[t] = accv[t] + sgn * out[t];  /* the float += / -= of :1493-1512 */
                            out[t] = accv[t];                  /* the loop works on the accumulators from here on */
                        }
                    const float* pd = (c->track_pilot && data_code) ? r->prompt_data : (const float*)(r->corr + 2 * prompt);
                    if (c->symbols_per_bit > 1)
                        {
                            if (c->data_secondary_code_length > 0)
                                {
                                    const float ds = c->data_secondary_code[current_data_symbol] == '0' ? 1.0F : -1.0F;
                                    p_data_accu[0] += ds * pd[0];
                                    p_data_accu[1] += ds * pd[1];
                                    current_data_symbol = (current_data_symbol + 1) % c->data_secondary_code_length;
                                }
                            else
                                {
                                    p_data_accu[0] += pd[0];
                                    p_data_accu[1] += pd[1];
                                    current_data_symbol = (current_data_symbol + 1) % c->symbols_per_bit;
                                }
                        }
                    else
                        {
                            p_data_accu[0] = pd[0];
                            p_data_accu[1] = pd[1];
                        }
                    cloop = c->track_pilot ? 0 : 1;  /* :1587-1595: pilot tracking disables the Costas loop */
                }
            memcpy(r->accu, out, sizeof(float) * 2 * n_taps);  /* d_VE_accu .. d_VL_accu as run_dll_pll / log_data see them (trk.cc:1624-1636) */
            const float* P = out + 2 * prompt;
            const float* E = out + 2 * (prompt - 1);
            const float* L = out + 2 * (prompt + 1);
            double carr_phase_error_hz = 0.0, carr_freq_error_hz = 0.0, carr_error_filt_hz = 0.0, code_error_chips = 0.0, code_error_filt_chips = 0.0;
            if (state == 3)  /* coherent integration: no lock test, no loop update (trk.cc:2156-2161); the detector values stand */
                {
                    if (c->enable_lock_detectors)
                        {
                            r->cn0_db_hz = lock.cn0_db_hz;
                            r->carrier_lock_test = lock.carrier_lock_test;
                        }
                    goto update_vars;
                }
            if (c->enable_lock_detectors)
                {
                    if (pull_in_latched && !pull_in)  /* trk.cc:1912-1916: leaving the pull-in transitory clears both fail counters */
                        {
                            pull_in_latched = 0;
                            lock.carrier_lock_fail_counter = 0;
                            lock.code_lock_fail_counter = 0;
                        }
                    /* trk.cc:2000-2007, state 2 only: no secondary-code / bit synchronisation within the time limit forces the loss-of-lock condition */
                    if (c->enable_bit_sync_time_limit && c->enable_symbol_sync && state == 2 &&
                        (uint64_t)c->bit_synchronization_time_limit_s < (pos - acq_sample_stamp) / (uint64_t)((int)c->fs_in))
                        lock.carrier_lock_fail_counter = 300000;
                    const int locked = oracle_lock_status(&lock, c, P[0], P[1], state == 4 ? code_period * (double)extend : code_period, pull_in);  /* trk.cc:2008, :2203 */
                    r->cn0_db_hz = lock.cn0_db_hz;
                    r->carrier_lock_test = lock.carrier_lock_test;
                    if (!locked)  /* trk.cc:2009-2014: clear_tracking_vars, d_state = 0 */
                        {
                            r->sample_counter = pos;
                            r->prn_length_samples = 0;
                            r->flags = (pull_in ? 1 : 0) | 2;
                            return e + 1;
                        }
                }

            /* run_dll_pll, trk.cc:1260-1324 */
            {
            float carr_error_filt;
            if (cloop)
                carr_phase_error_hz = oracle_pll_cloop_two_quadrant_atan(P[0], P[1]) / ORA_TWO_PI;
            else
                carr_phase_error_hz = oracle_pll_four_quadrant_atan(P[0], P[1]) / ORA_TWO_PI;
            if ((pull_in && c->enable_fll_pull_in) || c->enable_fll_steady_state)
                {
                    carr_freq_error_hz = oracle_fll_diff_atan(p_old_re, p_old_im, P[0], P[1], 0, corr_time) / ORA_TWO_PI;
                    p_old_re = P[0];
                    p_old_im = P[1];
                    if (pull_in && c->enable_fll_pull_in)
                        carr_error_filt = oracle_fll_pll_carrier_error(&pll, (float)carr_freq_error_hz, 0.0F, (float)corr_time);
                    else
                        carr_error_filt = oracle_fll_pll_carrier_error(&pll, (float)carr_freq_error_hz, (float)carr_phase_error_hz, (float)corr_time);
                }
            else
                {
                    carr_error_filt = oracle_fll_pll_carrier_error(&pll, 0, (float)carr_phase_error_hz, (float)corr_time);
                }
            carr_error_filt_hz = carr_error_filt;
            carrier_doppler_hz = carr_error_filt_hz;
            if (c->veml)
                code_error_chips = oracle_dll_nc_vemlp_normalized(out[0], out[1], out[2], out[3], out[6], out[7], out[8], out[9]);
            else
                code_error_chips = oracle_dll_nc_e_minus_l_normalized(E[0], E[1], L[0], L[1], spc_now, c->slope, c->y_intercept);
            code_error_filt_chips = oracle_loop_filter_apply(&dll, (float)code_error_chips);
            code_freq_chips = c->code_chip_rate - code_error_filt_chips;
            if (c->carrier_aiding) code_freq_chips += carrier_doppler_hz * c->code_chip_rate / c->signal_carrier_freq;
            /* trk.cc:1326-1346 */
            if (c->enable_doppler_correction && !pull_in && !corrected_doppler)
                {
                    dll_filt_sum += (double)(float)code_error_filt_chips;
                    dll_filt_count++;
                    if (dll_filt_count == 1000)
                        {
                            const float avg_code_error_chips_s = (float)dll_filt_sum / (float)1000;
                            if (fabs((double)avg_code_error_chips_s) > 1.0)
                                {
                                    const float carrier_doppler_error_hz = (float)c->signal_carrier_freq * avg_code_error_chips_s / (float)c->code_chip_rate;
                                    oracle_fll_pll_initialize(&pll, (float)carrier_doppler_hz - carrier_doppler_error_hz);
                                    corrected_doppler = 1;
                                }
                            dll_filt_sum = 0.0;
                            dll_filt_count = 0;
                        }
                }
            }

        update_vars:;
            /* update_tracking_vars, trk.cc:1409-1483 */
            const double t_chip = 1.0 / code_freq_chips;
            const double t_prn = t_chip * (double)c->code_length_chips;
            const double t_prn_samples = t_prn * c->fs_in;
            const double k_blk = t_prn_samples + rem_code_phase_samples;
            const int32_t prn_len = (int32_t)floor(k_blk);
            carrier_phase_step_rad = ORA_TWO_PI * (carrier_doppler_hz + c->cfo_frequency_hz) / c->fs_in;
            if (hd)  /* :1425-1443 */
                {
                    const int cap = 2 * SL;
                    carr_hist[carr_pushes % cap][0] = carrier_phase_step_rad;
                    carr_hist[carr_pushes % cap][1] = (double)prn_len;
                    carr_pushes++;
                    if (carr_hist_n < cap) carr_hist_n++;
                    if (carr_hist_n == cap)
                        {
                            const long oldest = carr_pushes % cap;  /* index 0 of the circular buffer */
                            double tmp_cp1 = 0.0, tmp_cp2 = 0.0, tmp_samples = 0.0;
                            for (int k = 0; k < SL; k++)
                                {
                                    tmp_cp1 += carr_hist[(oldest + k) % cap][0];
                                    tmp_cp2 += carr_hist[(oldest + cap - k - 1) % cap][0];
                                    tmp_samples += carr_hist[(oldest + cap - k - 1) % cap][1];
                                }
                            tmp_cp1 /= (double)SL;
                            tmp_cp2 /= (double)SL;
                            carrier_phase_rate_step_rad = (tmp_samples != 0) ? (tmp_cp2 - tmp_cp1) / tmp_samples : 0.0;
                        }
                }
            rem_carr_phase_rad += (float)(carrier_phase_step_rad * (double)prn_len + 0.5 * carrier_phase_rate_step_rad * (double)prn_len * (double)prn_len);
            rem_carr_phase_rad = (float)fmod(rem_carr_phase_rad, ORA_TWO_PI);
            acc_carrier_phase_rad -= (carrier_phase_step_rad * (double)prn_len + 0.5 * carrier_phase_rate_step_rad * (double)prn_len * (double)prn_len);
            code_phase_step_chips = code_freq_chips / c->fs_in;
            if (hd)  /* :1458-1480 */
                {
                    const int cap = 2 * SL;
                    code_hist[code_pushes % cap][0] = code_phase_step_chips;
                    code_hist[code_pushes % cap][1] = (double)prn_len;
                    code_pushes++;
                    if (code_hist_n < cap) code_hist_n++;
                    if (code_hist_n == cap)
                        {
                            const long oldest = code_pushes % cap;
                            double tmp_cp1 = 0.0, tmp_cp2 = 0.0, tmp_samples = 0.0;
                            for (int k = 0; k < SL; k++)
                                {
                                    tmp_cp1 += code_hist[(oldest + k) % cap][0];
                                    tmp_cp2 += code_hist[(oldest + cap - k - 1) % cap][0];
                                    tmp_samples += code_hist[(oldest + cap - k - 1) % cap][1];
                                }
                            tmp_cp1 /= (double)SL;
                            tmp_cp2 /= (double)SL;
                            if (tmp_samples >= 1.0) code_phase_rate_step_chips = (tmp_cp2 - tmp_cp1) / tmp_samples;
                        }
                }
            rem_code_phase_samples = k_blk - (double)prn_len;
            rem_code_phase_chips = code_freq_chips * rem_code_phase_samples / c->fs_in;

            r->sample_counter = pos;
            r->prn_length_samples = prn_len;
            r->flags = pull_in ? 1 : 0;
            r->carrier_doppler_hz = carrier_doppler_hz;
            r->code_freq_chips = code_freq_chips;
            r->carr_phase_error_hz = carr_phase_error_hz;
            r->carr_freq_error_hz = carr_freq_error_hz;
            r->carr_error_filt_hz = carr_error_filt_hz;
            r->code_error_chips = code_error_chips;
            r->code_error_filt_chips = code_error_filt_chips;
            r->rem_code_phase_samples = rem_code_phase_samples;
            r->acc_carrier_phase_rad = acc_carrier_phase_rad;
            r->rem_carr_phase_rad = rem_carr_phase_rad;
            r->carrier_phase_rate_step_rad = carrier_phase_rate_step_rad;
            r->code_phase_rate_step_chips = code_phase_rate_step_chips;
            if (c->enable_symbol_sync && state == 3)
                {
                    /* trk.cc:2162-2194: a telemetry symbol may complete inside the coherent integration; then count the period */
                    r->p_data_accu[0] = p_data_accu[0];
                    r->p_data_accu[1] = p_data_accu[1];
                    if (current_data_symbol == 0)
                        {
                            r->symbol_flags |= 1;
                            p_data_accu[0] = p_data_accu[1] = 0.0F;
                        }
                    if (flag_pll_180) r->symbol_flags |= 2;
                    ext_count++;
                    if (ext_count == extend - 1)
                        {
                            ext_count = 0;
                            state = 4;
                        }
                }
            else if (c->enable_symbol_sync)
                {
                    if (state == 2)
                        {
                            int next_state = 0;
                            if (!pull_in)  /* trk.cc:2026 */
                                {
                                    if (!c->has_secondary && c->symbols_per_bit > 1 && use_hist)  /* :2046-2072 */
                                        {
                                            const int lock_event = oracle_bit_sync_update(&bs, r->corr[2 * prompt], r->corr[2 * prompt + 1], 1);
                                            if (lock_event)
                                                {
                                                    wait_for_bit_edge = 1;
                                                    const int64_t k_now = bs.epoch_count - 1;
                                                    int wait = oracle_bit_sync_epochs_until_next_edge(&bs) - 1;
                                                    if (wait < 0) wait = wait + bs.bins;
                                                    bit_sync_target_epoch = k_now + wait;
                                                }
                                            if (wait_for_bit_edge)
                                                {
                                                    const int64_t k_now = bs.epoch_count - 1;
                                                    if (k_now == bit_sync_target_epoch)
                                                        {
                                                            next_state = 1;
                                                            wait_for_bit_edge = 0;
                                                            use_hist = 0;
                                                        }
                                                }
                                        }
                                    if (!next_state && (c->has_secondary || c->symbols_per_bit > 1))  /* :2028-2045, :2074-2089 */
                                        {
                                            /* d_Prompt_circular_buffer.push_back(*d_Prompt) */
                                            const float* pr = r->corr + 2 * prompt;
                                            const int len = c->secondary_code_length;
                                            if (ring_count < len)
                                                {
                                                    ring[2 * ring_count] = pr[0];
                                                    ring[2 * ring_count + 1] = pr[1];
                                                    ring_count++;
                                                }
                                            else if (len > 0)
                                                {
                                                    ring[2 * ring_head] = pr[0];
                                                    ring[2 * ring_head + 1] = pr[1];
                                                    ring_head = (ring_head + 1) % len;
                                                }
                                            if (len > 0 && ring_count == len)
                                                {
                                                    /* acquire_secondary, trk.cc:1118-1160: sign pattern of the buffered prompts against the code */
                                                    int corr_value = 0;
                                                    for (int i = 0; i < len; i++)
                                                        {
                                                            const float re = ring[2 * ((ring_head + i) % len)];
                                                            if (re < 0.0)
                                                                corr_value += c->secondary_code[i] == '0' ? 1 : -1;
                                                            else
                                                                corr_value += c->secondary_code[i] == '0' ? -1 : 1;
                                                        }
                                                    if (abs(corr_value) == len)
                                                        {
                                                            flag_pll_180 = corr_value < 0 ? 1 : 0;
                                                            next_state = 1;
                                                        }
                                                }
                                        }
                                    if (!c->has_secondary && c->symbols_per_bit <= 1)
                                        {
                                            next_state = 1;  /* :2091-2094 */
                                        }
                                }
                            if (next_state)  /* :2101-2112 + :2151-2154 (no extended integration) */
                                {
                                    p_data_accu[0] = p_data_accu[1] = 0.0F;
                                    ring_count = ring_head = 0;
                                    current_symbol = 0;
                                    current_data_symbol = 0;
                                    for (int t = 0; t < 10; t++) accv[t] = 0.0F;
                                    if (extend > 1)  /* trk.cc:2114-2149: stretch the integration time, narrow the loops and the correlator spacing */
                                        {
                                            ext_count = 0;
                                            corr_time = (double)((float)extend * (float)code_period);
                                            state = 3;
                                            oracle_loop_filter narrow_dll;
                                            oracle_loop_filter_design(&narrow_dll, (float)corr_time, c->dll_bw_narrow_hz, c->dll_filter_order, 0);
                                            memcpy(dll.in_c, narrow_dll.in_c, sizeof(dll.in_c));   /* update_coefficients keeps the histories (T/tracking_loop_filter.cc:101-196) */
                                            memcpy(dll.out_c, narrow_dll.out_c, sizeof(dll.out_c));
                                            dll.n_in = narrow_dll.n_in;
                                            dll.n_out = narrow_dll.n_out;
                                            oracle_fll_pll_filter narrow_pll;
                                            oracle_fll_pll_design(&narrow_pll, c->fll_bw_hz, c->pll_bw_narrow_hz, c->pll_filter_order);  /* set_params keeps d_pll_w / d_pll_x */
                                            narrow_pll.w = pll.w;
                                            narrow_pll.x = pll.x;
                                            pll = narrow_pll;
                                            if (c->veml)
                                                {
                                                    shifts[0] = -c->very_early_late_space_narrow_chips * spcf;
                                                    shifts[1] = -c->early_late_space_narrow_chips * spcf;
                                                    shifts[3] = c->early_late_space_narrow_chips * spcf;
                                                    shifts[4] = c->very_early_late_space_narrow_chips * spcf;
                                                }
                                            else
                                                {
                                                    shifts[0] = -c->early_late_space_narrow_chips * spcf;
                                                    shifts[2] = c->early_late_space_narrow_chips * spcf;
                                                }
                                            spc_now = c->early_late_space_narrow_chips;
                                        }
                                    else
                                        {
                                            state = 4;
                                        }
                                }
                        }
                    else
                        {
                            /* state 4 after run_dll_pll / update_tracking_vars: check_carrier_phase_coherent_initialization (:1350-1357) */
                            if (!acc_phase_initialized)
                                {
                                    acc_carrier_phase_rad = -(double)rem_carr_phase_rad;
                                    acc_phase_initialized = 1;
                                    r->acc_carrier_phase_rad = acc_carrier_phase_rad;
                                }
                            r->p_data_accu[0] = p_data_accu[0];
                            r->p_data_accu[1] = p_data_accu[1];
                            if (current_data_symbol == 0)  /* :2212-2236: one telemetry symbol leaves the block */
                                {
                                    r->symbol_flags |= 1;
                                    p_data_accu[0] = p_data_accu[1] = 0.0F;
                                }
                            for (int t = 0; t < 10; t++) accv[t] = 0.0F;  /* :2241-2246 reset extended correlator */
                            if (extend > 1) state = 3;                  /* :2247-2250 */
                        }
                    if (flag_pll_180) r->symbol_flags |= 2;
                }
            pos += (uint64_t)prn_len;
        }
    return n_epochs;
}
