/*!
 * \file hip_tracking_loop.cc
 * \brief Host-side class over the device-closed DLL/PLL loop; see hip_tracking_loop.h.
 */
#include "hip_tracking_loop.h"
#include <algorithm>

Hip_Tracking_Loop::Hip_Tracking_Loop(int device, const gsh_trk_conf& conf, int max_code_length, std::shared_ptr<Hip_Sample_Ring> shared_ring,
    uint64_t ring_capacity_samples)
    : d_conf(conf), d_ring(std::move(shared_ring)), d_done(1, 0)
{
    if (!d_ring)
        {
            // forecast asks for 2 * vector_length per call (trk.cc:747-754); eight windows leave room for a scheduler that runs ahead
            const uint64_t cap = ring_capacity_samples ? ring_capacity_samples : 8ULL * conf.vector_length;
            d_ring = std::make_shared<Hip_Sample_Ring>(device, cap, 2U * conf.vector_length);
            if (!d_ring->ok())
                {
                    d_error = "sample ring: " + d_ring->last_error();
                    return;
                }
        }
    if (gsh_trk_create(device, &d_conf, 1, max_code_length, &d_trk) != GSH_OK)
        {
            fail("gsh_trk_create");
            d_trk = nullptr;
            return;
        }
    if (gsh_trk_set_stream_ring(d_trk, d_ring->handle()) != GSH_OK)
        {
            fail("gsh_trk_set_stream_ring");
            gsh_trk_destroy(d_trk);
            d_trk = nullptr;
        }
}

Hip_Tracking_Loop::~Hip_Tracking_Loop()
{
    if (d_trk != nullptr) gsh_trk_destroy(d_trk);
}

bool Hip_Tracking_Loop::fail(const char* what)
{
    d_error = std::string(what) + ": " + gsh_last_error();
    return false;
}

bool Hip_Tracking_Loop::start(const float* code, const float* data_code, int code_length, uint64_t nitems_read, double acq_delay_samples,
    double acq_doppler_hz, uint64_t acq_samplestamp_samples, int32_t* samples_offset)
{
    if (!ok()) return false;
    int32_t offset = 0, first_len = 0;
    double acc0 = 0.0;
    if (gsh_trk_pull_in(&d_conf, nitems_read, acq_delay_samples, acq_samplestamp_samples, acq_doppler_hz, &offset, &first_len, &acc0) != GSH_OK)
        return fail("gsh_trk_pull_in");
    const uint64_t start_sample = nitems_read + static_cast<uint64_t>(std::max(offset, 0));
    if (gsh_trk_start_ex(d_trk, 0, code, data_code, code_length, start_sample, acq_samplestamp_samples, acq_doppler_hz, acc0) != GSH_OK)
        return fail("gsh_trk_start_ex");
    if (samples_offset != nullptr) *samples_offset = offset;
    d_next_window = start_sample;
    d_tracking = true;
    return true;
}

void Hip_Tracking_Loop::stop()
{
    if (ok() && d_tracking) (void)gsh_trk_stop(d_trk, 0);
    d_tracking = false;
}

bool Hip_Tracking_Loop::push(const std::complex<float>* samples, uint64_t first_index, uint64_t n)
{
    if (!ok()) return false;
    uint64_t oldest = 0, next = 0;
    d_ring->range(&oldest, &next);
    if (first_index + n <= next) return true;  // somebody (another channel of the stream, an earlier call) has pushed them
    if (first_index > next)
        {
            if (next == oldest)
                {
                    // an empty ring starts wherever its first user is
                    if (!d_ring->seek(first_index))
                        {
                            d_error = "sample ring: " + d_ring->last_error();
                            return false;
                        }
                    next = first_index;
                }
            else
                {
                    d_error = "push: samples " + std::to_string(first_index) + ".. leave a gap after the ring's " + std::to_string(next);
                    return false;
                }
        }
    const uint64_t skip = next - first_index;
    if (d_ring->push(samples + skip, n - skip) == UINT64_MAX)
        {
            d_error = "sample ring: " + d_ring->last_error();
            return false;
        }
    return true;
}

int Hip_Tracking_Loop::run(int max_periods, gsh_trk_epoch* records)
{
    if (!ok() || !d_tracking || max_periods <= 0) return ok() ? 0 : -1;
    if (gsh_trk_run(d_trk, max_periods, records, d_done.data()) != GSH_OK)
        {
            fail("gsh_trk_run");
            return -1;
        }
    const int done = d_done[0];
    for (int e = 0; e < done; e++)
        {
            if (records[e].flags & 2)
                {
                    d_tracking = false;  // loss of lock: the device has stopped the channel (trk.cc:2009-2014)
                    break;
                }
            d_next_window = records[e].sample_counter + static_cast<uint64_t>(records[e].prn_length_samples);
        }
    return done;
}
