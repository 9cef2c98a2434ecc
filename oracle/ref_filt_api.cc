/*
 * TEST INFRASTRUCTURE ONLY -- never linked into the product.
 *
 * C driver for the reference's own input-filter blocks, compiled in place by oracle/Makefile into oracle/_ref/libgnsssdr_ref_filt.so:
 *   kind 0  pulse_blanking_cc   (src/algorithms/input_filter/gnuradio_blocks/pulse_blanking_cc.cc)
 *   kind 1  Notch               (.../notch_cc.cc)
 *   kind 2  NotchLite           (.../notch_lite_cc.cc)
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

#define private public
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
}
