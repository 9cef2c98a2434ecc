/*
 * TEST INFRASTRUCTURE ONLY -- never linked into the product.
 *
 * C driver for the reference's own input-filter blocks, compiled in place by oracle/Makefile into oracle/_ref/libgnsssdr_ref_filt.so:
 *   kind 0  pulse_blanking_cc   (src/algorithms/input_filter/gnuradio_blocks/pulse_blanking_cc.cc)
 *   kind 1  Notch               (.../notch_cc.cc)
 *   kind 2  NotchLite           (.../notch_lite_cc.cc)
 * and, through refconv_* (round 6), the front-end blocks whose arithmetic the engine's casts and resampler must reproduce bit for bit:
 *   kind 3 / 4 / 5  direct_resampler_conditioner_cc / _cb / _cs   (src/algorithms/resampler/gnuradio_blocks/: gr_complex, lv_8sc_t, lv_16sc_t items)
 *   kind 6  cshort_to_gr_complex                (src/algorithms/data_type_adapter/gnuradio_blocks/)
 *   kind 7  interleaved_byte_to_complex_byte    kind 8  interleaved_byte_to_complex_short    kind 9  interleaved_short_to_complex_short
 * A block is driven through its own general_work by a harness that plays the GNU Radio scheduler: the caller hands `n_items` input items (with the
 * block's history already in front, as the scheduler would) and room for `noutput_items`, and learns what the block consumed and produced.
 * Used by tests/test_notch_oracle_pinned.py to pin oracle/notch_oracle.py and the pulse-blanking restatement in oracle/fir_oracle.py.
 */
#include <algorithm>
#include <any>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <gnuradio/block.h>
#include <gnuradio/fft/fft.h>
#include <volk_gnsssdr/volk_gnsssdr_alloc.h>
#include "gnss_block_interface.h"
#include "gnss_sdr_fft.h"

#include "cshort_to_gr_complex.h"
#include "interleaved_byte_to_complex_byte.h"
#include "interleaved_byte_to_complex_short.h"
#include "interleaved_short_to_complex_short.h"
#define private public
#include "direct_resampler_conditioner_cb.h"
#include "direct_resampler_conditioner_cc.h"
#include "direct_resampler_conditioner_cs.h"
#include "notch_cc.h"
#include "notch_lite_cc.h"
#include "pulse_blanking_cc.h"
#undef private

namespace
{
struct Handle
{
    int kind{0};
    pulse_blanking_cc_sptr pb;
    notch_sptr notch;
    notch_lite_sptr lite;
    gr::block* blk{nullptr};
};
}  // namespace

extern "C"
{
    /* volk_gnsssdr::vector's allocator (volk_gnsssdr_alloc.h) */
    void* volk_gnsssdr_malloc(size_t size, size_t alignment)
    {
        void* p = nullptr;
        if (alignment < sizeof(void*)) alignment = sizeof(void*);
        if (posix_memalign(&p, alignment, size ? size : 1) != 0) return nullptr;
        return p;
    }
    void volk_gnsssdr_free(void* p) { std::free(p); }
    size_t volk_gnsssdr_get_alignment(void) { return 32; }

    /* p: pfa, p_c_factor (notch kinds), length, n_segments_est, n_segments_reset, n_segments_coeff (NotchLite) */
    void* reffilt_create(int kind, float pfa, float p_c_factor, int32_t length, int32_t n_segments_est, int32_t n_segments_reset, int32_t n_segments_coeff)
    {
        try
            {
                auto h = std::make_unique<Handle>();
                h->kind = kind;
                switch (kind)
                    {
                    case 0:
                        h->pb = make_pulse_blanking_cc(pfa, length, n_segments_est, n_segments_reset);
                        h->blk = h->pb.get();
                        break;
                    case 1:
                        h->notch = make_notch_filter(pfa, p_c_factor, length, n_segments_est, n_segments_reset);
                        h->blk = h->notch.get();
                        break;
                    case 2:
                        h->lite = make_notch_filter_lite(p_c_factor, pfa, length, n_segments_est, n_segments_reset, n_segments_coeff);
                        h->blk = h->lite.get();
                        break;
                    default:
                        return nullptr;
                    }
                return h.release();
            }
        catch (const std::exception& e)
            {
                std::cerr << "reffilt_create: " << e.what() << '\n';
                return nullptr;
            }
    }

    void reffilt_destroy(void* hv) { delete static_cast<Handle*>(hv); }

    /* one scheduler call; returns general_work's return value (items produced), *consumed = what it passed to consume_each */
    int reffilt_general_work(void* hv, const float* in_iq, int n_items, int noutput_items, float* out_iq, int* consumed)
    {
        auto* h = static_cast<Handle*>(hv);
        gr_vector_int ninput{n_items};
        gr_vector_const_void_star in{static_cast<const void*>(in_iq)};
        gr_vector_void_star out{static_cast<void*>(out_iq)};
        h->blk->consumed_last = 0;
        const int r = h->blk->general_work(noutput_items, ninput, in, out);
        *consumed = h->blk->consumed_last;
        return r;
    }

    /* threshold, noise power estimate, segment counter, filter state, last output (notch kinds) */
    void reffilt_state(void* hv, float* thres, float* noise_pow_est, int32_t* n_segments, int32_t* filter_state, float* last_out_iq, int32_t* n_segments_coeff,
        float* z0_iq)
    {
        auto* h = static_cast<Handle*>(hv);
        *n_segments_coeff = 0;
        last_out_iq[0] = last_out_iq[1] = 0.0f;
        z0_iq[0] = z0_iq[1] = 0.0f;
        switch (h->kind)
            {
            case 0:
                *thres = h->pb->thres_;
                *noise_pow_est = h->pb->noise_power_estimation_;
                *n_segments = h->pb->n_segments_;
                *filter_state = h->pb->last_filtered_ ? 1 : 0;
                break;
            case 1:
                *thres = h->notch->thres_;
                *noise_pow_est = h->notch->noise_pow_est_;
                *n_segments = h->notch->n_segments_;
                *filter_state = h->notch->filter_state_ ? 1 : 0;
                last_out_iq[0] = h->notch->last_out_.real();
                last_out_iq[1] = h->notch->last_out_.imag();
                z0_iq[0] = h->notch->z_0_.real();
                z0_iq[1] = h->notch->z_0_.imag();
                break;
            default:
                *thres = h->lite->thres_;
                *noise_pow_est = h->lite->noise_pow_est_;
                *n_segments = h->lite->n_segments_;
                *filter_state = h->lite->filter_state_ ? 1 : 0;
                last_out_iq[0] = h->lite->last_out_.real();
                last_out_iq[1] = h->lite->last_out_.imag();
                *n_segments_coeff = h->lite->n_segments_coeff_;
                z0_iq[0] = h->lite->z_0_.real();
                z0_iq[1] = h->lite->z_0_.imag();
                break;
            }
    }

    /* ---- front-end blocks: resamplers and data-type adapters (items of any size; the caller knows them) */
    struct ConvHandle
    {
        std::shared_ptr<gr::block> blk;
        int in_item{0}, out_item{0};
    };

    void* refconv_create(int kind, double fs_in, double fs_out)
    {
        try
            {
                auto h = std::make_unique<ConvHandle>();
                switch (kind)
                    {
                    case 3:
                        h->blk = direct_resampler_make_conditioner_cc(fs_in, fs_out);
                        h->in_item = h->out_item = 8;
                        break;
                    case 4:
                        h->blk = direct_resampler_make_conditioner_cb(fs_in, fs_out);
                        h->in_item = h->out_item = 2;
                        break;
                    case 5:
                        h->blk = direct_resampler_make_conditioner_cs(fs_in, fs_out);
                        h->in_item = h->out_item = 4;
                        break;
                    case 6:
                        h->blk = make_cshort_to_gr_complex();
                        h->in_item = 4;
                        h->out_item = 8;
                        break;
                    case 7:
                        h->blk = make_interleaved_byte_to_complex_byte();
                        h->in_item = 1;
                        h->out_item = 2;
                        break;
                    case 8:
                        h->blk = make_interleaved_byte_to_complex_short();
                        h->in_item = 1;
                        h->out_item = 4;
                        break;
                    case 9:
                        h->blk = make_interleaved_short_to_complex_short();
                        h->in_item = 2;
                        h->out_item = 4;
                        break;
                    default:
                        return nullptr;
                    }
                return h.release();
            }
        catch (const std::exception& e)
            {
                std::cerr << "refconv_create: " << e.what() << '\n';
                return nullptr;
            }
    }

    void refconv_destroy(void* hv) { delete static_cast<ConvHandle*>(hv); }
    int refconv_item_sizes(void* hv, int* in_item, int* out_item)
    {
        auto* h = static_cast<ConvHandle*>(hv);
        *in_item = h->in_item;
        *out_item = h->out_item;
        return 0;
    }

    /* the block's forecast for noutput_items outputs (what the scheduler would make available before calling) */
    int refconv_forecast(void* hv, int noutput_items)
    {
        auto* h = static_cast<ConvHandle*>(hv);
        gr_vector_int req{0};
        h->blk->forecast(noutput_items, req);
        return req[0];
    }

    /* one scheduler call over n_items input items; returns items produced, *consumed = what the block passed to consume_each; the stream positions advance */
    int refconv_general_work(void* hv, const void* in, int n_items, int noutput_items, void* out, int* consumed)
    {
        auto* h = static_cast<ConvHandle*>(hv);
        gr_vector_int ninput{n_items};
        gr_vector_const_void_star iv{in};
        gr_vector_void_star ov{out};
        h->blk->consumed_last = 0;
        const int r = h->blk->general_work(noutput_items, ninput, iv, ov);
        *consumed = h->blk->consumed_last;
        h->blk->mock_advance(r);
        return r;
    }
}
