// Internal C++ interface between the C-ABI layer (gsh_api.hip) and the correlator kernels
// (multicorrelator.hip).  Not part of the ABI.
#ifndef GSH_MULTICORRELATOR_H
#define GSH_MULTICORRELATOR_H

#include "gsh_internal.h"

namespace gsh
{
// a job whose early tap may be read next to its late tap (mcorr_device.h packed_trip): E/P/L, prompt at exactly 0, late - early exactly 1 chip,
// standard mode, the code running forward
__host__ __device__ inline bool mcorr_pair_eligible(int n_taps, const float* shifts, float code_step, int mode)
{
    return n_taps == 3 && mode == 0 && shifts[1] == 0.0f && (static_cast<double>(shifts[2]) - static_cast<double>(shifts[0]) == 1.0) && code_step > 0.0f;
}

struct McorrArgs
{
    const float2* stream;            // device, complex64 IF samples
    unsigned long long stream_len;   // samples
    const gsh_corr_job* jobs;        // device, n_jobs entries
    const float* codes;              // device, n_slots * code_stride floats
    const int* code_lens;            // device, n_slots
    int code_stride;
    float2* out;                     // device, n_jobs * GSH_MAX_TAPS
    float2* partials;                // device, n_jobs * splits * GSH_MAX_TAPS (splits > 1 only)
    int n_jobs;
    int splits;
    int window_floats;               // > 0: LDS holds only a window of the code per work-group (all jobs: mode 0, code_step >= 0); 0: the whole code
    const int* job_list;             // device, n_launch job indices this launch works on (order kept), or nullptr: jobs 0..n_launch-1
    int n_launch;                    // jobs in this launch (n_jobs stays the size of the whole table: outputs / partials are indexed by job)
    unsigned long long sample_base;  // added to every job's sample_offset at launch (gsh_bank_set_sample_base): the same resident job table serves block after block
    unsigned long long ring_capacity;// > 0: the stream is a gsh_stream ring, positions are taken modulo its capacity (windows stay contiguous: mirror)
    int packed;                      // 1: the packed four-samples-per-lane body may be used (default); 0: the round-1 body (A/B runs)
    int fac;                         // 1: the carrier seeds come from the work-group's factor table (default); 0: two evaluations per lane (A/B runs, GSH_MC_FAC=0)
    int pair;                        // 1: every job with two or three taps in this batch is an E/P/L set with a zero-shift prompt, early and late exactly one chip
                                     // apart and the code running forward (the host checked): the 3-tap launch reads early next to late (mcorr_device.h)
    const int* aux;                  // device, n_jobs, or nullptr.  aux[j] >= 0: job j also computes the single tap of job aux[j] (same window and
                                     // NCO, another code) and writes its output row; -2: job j is computed by its leader; -1: plain job
};

// Largest n_taps over the jobs and which mode combinations occur decide the template
// instance; all jobs of one launch must share `mode` (gsh_corr_job::high_dyn).
// 1 unless the environment says GSH_MC_PACKED_BODY=0 (read once; A/B switch for profiles/ab/mcorr_ab.py)
int mcorr_packed_default();
// 1 unless the environment says GSH_MC_FAC=0 (read once; A/B switch)
int mcorr_fac_default();

int mcorr_launch(const McorrArgs& args, int max_taps, int mode, int max_code_len, hipStream_t stream);

// A batch whose jobs differ in tap count is launched per kernel flavour (1, <= 3, <= 5, <= 8 taps) so that a 3-tap job does not pay for
// five: `list` holds the job indices grouped by class (device), offset / count per class, aux[c] whether class c contains fused leaders.
struct McorrClassPlan
{
    const int* list;
    int offset[4];
    int count[4];
    bool aux[4];
};
int mcorr_launch_classes(const McorrArgs& args, const McorrClassPlan& plan, int mode, int max_code_len, hipStream_t stream);
// the same through the 128-thread kernels (csrc/multicorrelator_t128.hip); mcorr_launch / mcorr_launch_classes choose, callers do not
int mcorr_launch_t128(const McorrArgs& args, int max_taps, int mode, int max_code_len, hipStream_t stream);
int mcorr_launch_classes_t128(const McorrArgs& args, const McorrClassPlan& plan, int mode, int max_code_len, hipStream_t stream);

// dynamic LDS bytes the kernel needs for a code of max_code_len samples
size_t mcorr_lds_bytes(int max_code_len);
// the same when only `window_floats` code samples are staged per work-group
size_t mcorr_lds_bytes_window(int window_floats);
// LDS bytes with room for the fused correlator's second code table (aux != nullptr)
size_t mcorr_lds_bytes_fused(int max_code_len, int window_floats);
}  // namespace gsh

#endif
