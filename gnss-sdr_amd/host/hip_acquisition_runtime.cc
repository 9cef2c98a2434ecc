/*!
 * \file hip_acquisition_runtime.cc
 * \brief Rendezvous of the acquisition blocks of one stream; see hip_acquisition_runtime.h.
 */
#include "hip_acquisition_runtime.h"
#include <algorithm>
#include <cstring>

Hip_Acquisition_Runtime::Hip_Acquisition_Runtime(int device, const gsh_acq_conf& conf, int max_channels, std::chrono::microseconds max_wait)
    : d_device(device), d_conf(conf), d_max_wait(max_wait)
{
    d_conf.max_prn = static_cast<uint32_t>(std::max(max_channels, 1));
    d_conf.no_grid = 1;                  // statistics on chip: nothing of a batch is kept between dwells
    d_conf.num_doppler_bins_step2 = 0;   // the fine-Doppler step runs on the blocks' own handles
    d_conf.doppler_center = 0;
    d_conf.doppler_bias = 0;
    if (gsh_acq_create(device, &d_conf, &d_handle) != GSH_OK)
        {
            d_error = gsh_last_error();
            d_handle = nullptr;
            return;
        }
    d_used.assign(d_conf.max_prn, 0);
    d_announced.assign(d_conf.max_prn, -1);
}


Hip_Acquisition_Runtime::~Hip_Acquisition_Runtime()
{
    if (d_handle != nullptr) gsh_acq_destroy(d_handle);
}


bool Hip_Acquisition_Runtime::same_geometry(const gsh_acq_conf& o) const
{
    const gsh_acq_conf& c = d_conf;
    return c.fs_in == o.fs_in && c.fft_size == o.fft_size && c.effective_fft_size == o.effective_fft_size && c.consumed_samples == o.consumed_samples &&
           c.num_doppler_bins == o.num_doppler_bins && c.doppler_max == o.doppler_max && c.doppler_step == o.doppler_step && c.samples_per_chip == o.samples_per_chip &&
           c.samples_per_code == o.samples_per_code && c.bit_transition_flag == o.bit_transition_flag && c.use_cfar == o.use_cfar && c.fold == o.fold &&
           c.transform_path == o.transform_path;
}


uint64_t Hip_Acquisition_Runtime::next_window(uint64_t sample_index) const
{
    const uint64_t L = std::max<uint32_t>(d_conf.consumed_samples, 1U);
    return (sample_index + L - 1) / L * L;
}


int Hip_Acquisition_Runtime::attach()
{
    std::lock_guard<std::mutex> lk(d_mutex);
    for (size_t i = 0; i < d_used.size(); i++)
        if (!d_used[i])
            {
                d_used[i] = 1;
                d_announced[i] = -1;
                return static_cast<int>(i);
            }
    return -1;
}


void Hip_Acquisition_Runtime::detach(int slot)
{
    withdraw(slot);
    std::lock_guard<std::mutex> lk(d_mutex);
    if (slot >= 0 && slot < static_cast<int>(d_used.size())) d_used[static_cast<size_t>(slot)] = 0;
}


bool Hip_Acquisition_Runtime::set_local_code(int slot, const std::complex<float>* code)
{
    if (d_handle == nullptr || slot < 0 || slot >= static_cast<int>(d_used.size()) || code == nullptr) return false;
    std::lock_guard<std::mutex> hl(d_handle_mutex);
    if (gsh_acq_set_local_code(d_handle, static_cast<uint32_t>(slot), reinterpret_cast<const float*>(code)) != GSH_OK)
        {
            std::lock_guard<std::mutex> lk(d_mutex);
            d_error = gsh_last_error();
            return false;
        }
    return true;
}


void Hip_Acquisition_Runtime::searching(int slot)
{
    std::lock_guard<std::mutex> lk(d_mutex);
    if (slot >= 0 && slot < static_cast<int>(d_announced.size()) && d_announced[static_cast<size_t>(slot)] == -1) d_announced[static_cast<size_t>(slot)] = -2;
}


void Hip_Acquisition_Runtime::announce(int slot, uint64_t window_start)
{
    {
        std::lock_guard<std::mutex> lk(d_mutex);
        if (slot >= 0 && slot < static_cast<int>(d_announced.size())) d_announced[static_cast<size_t>(slot)] = static_cast<int64_t>(window_start);
    }
    d_cv.notify_all();  // a batch that waited for this (until now undecided) channel looks again: it may be heading elsewhere
}


void Hip_Acquisition_Runtime::withdraw(int slot)
{
    {
        std::lock_guard<std::mutex> lk(d_mutex);
        if (slot < 0 || slot >= static_cast<int>(d_announced.size())) return;
        d_announced[static_cast<size_t>(slot)] = -1;
    }
    d_cv.notify_all();  // a batch that waited for this channel must look again
}


int Hip_Acquisition_Runtime::announced_for(uint64_t window_start) const
{
    int n = 0;
    for (const int64_t w : d_announced)
        if (w == static_cast<int64_t>(window_start) || w == -2) n++;
    return n;
}


// launch one closed batch (exactly one thread per batch; d_mutex is released around the device work)
void Hip_Acquisition_Runtime::run(const std::shared_ptr<Batch>& b, std::unique_lock<std::mutex>& lk, bool timed_out)
{
    lk.unlock();
    int rc;
    std::string err;
    b->results.assign(b->slots.size(), gsh_acq_result{});
    {
        std::lock_guard<std::mutex> hl(d_handle_mutex);
        rc = gsh_acq_dwell_slots(d_handle, reinterpret_cast<const float*>(b->data), static_cast<uint32_t>(b->slots.size()), b->slots.data(), b->results.data());
        if (rc != GSH_OK) err = gsh_last_error();
    }
    lk.lock();
    b->status = rc;
    b->error = err;
    b->done = true;
    d_stats.batches++;
    d_stats.dwells += b->slots.size();
    d_stats.timeouts += timed_out ? 1U : 0U;
    d_stats.largest_batch = std::max<uint32_t>(d_stats.largest_batch, static_cast<uint32_t>(b->slots.size()));
    d_cv.notify_all();
}


bool Hip_Acquisition_Runtime::dwell(int slot, uint64_t window_start, const std::complex<float>* window, gsh_acq_result* out)
{
    if (d_handle == nullptr || window == nullptr || out == nullptr || slot < 0 || slot >= static_cast<int>(d_used.size())) return false;
    std::unique_lock<std::mutex> lk(d_mutex);
    auto& entry = d_batches[window_start];
    if (!entry || entry->taken)
        {
            // (a batch of this window that has already been closed -- this block was late -- is left to its participants: open the next one)
            entry = std::make_shared<Batch>();
            entry->first_arrival = std::chrono::steady_clock::now();
            entry->data = window;
        }
    std::shared_ptr<Batch> b = entry;
    const size_t my_index = b->slots.size();
    b->slots.push_back(static_cast<uint32_t>(slot));
    d_announced[static_cast<size_t>(slot)] = -1;  // joined: no longer awaited
    const bool first = my_index == 0;
    if (announced_for(window_start) == 0)
        {
            // nobody else is still buffering this window: close the batch and launch from this very thread
            b->taken = true;
            d_batches.erase(window_start);
            run(b, lk, false);
        }
    else if (first)
        {
            // first arriver: bounded wait for the channels that announced this window; whoever completes the batch takes it
            const auto deadline = b->first_arrival + d_max_wait;
            while (!b->taken && announced_for(window_start) > 0)
                if (d_cv.wait_until(lk, deadline) == std::cv_status::timeout) break;
            if (!b->taken)
                {
                    b->taken = true;
                    const bool timed_out = announced_for(window_start) > 0;
                    auto it = d_batches.find(window_start);
                    if (it != d_batches.end() && it->second == b) d_batches.erase(it);
                    run(b, lk, timed_out);
                }
        }
    else
        {
            d_cv.notify_all();  // the first arriver re-evaluates (this arrival may have completed the batch)
        }
    while (!b->done) d_cv.wait(lk);
    if (b->status != GSH_OK)
        {
            d_error = b->error;
            return false;
        }
    *out = b->results[my_index];
    return true;
}


Hip_Acquisition_Runtime::Stats Hip_Acquisition_Runtime::stats() const
{
    std::lock_guard<std::mutex> lk(d_mutex);
    return d_stats;
}
