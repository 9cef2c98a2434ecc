// Internal C++ interface of the PCPS acquisition kernels (pcps_fft.hip) used by acquisition_api.hip.
// Not part of the ABI.
#ifndef GSH_PCPS_FFT_H
#define GSH_PCPS_FFT_H

#include "gsh_internal.h"
#include <vector>

namespace gsh
{
constexpr int FFT_MAX_PASSES = 16;

// radix schedule of one LDS-resident sub-transform
struct SubPlan
{
    int len{1};
    int n_pass{0};
    int radix[FFT_MAX_PASSES]{};
};

// N = n1 * n2, both sub-transforms LDS resident ("four-step" without the transposes, see pcps_fft.hip)
struct FftPlan
{
    int n{0}, n1{0}, n2{0};
    SubPlan p1, p2;
    int tile_cols{8};  // columns per work-group in the column pass
    int tile_rows{1};  // rows per work-group in the row pass
    float2* d_tw_n{nullptr};  // exp(-2 pi i j / n),  j < n
    float2* d_tw_1{nullptr};  // exp(-2 pi i j / n1), j < n1
    float2* d_tw_2{nullptr};  // exp(-2 pi i j / n2), j < n2
};

// host-only helpers (no device work): testable without a GPU
bool factor_length(int len, SubPlan* out);            // false when a prime factor > 31 remains
bool choose_split(int n, int* n1, int* n2);           // false when no LDS-resident split exists

int plan_create(int n, FftPlan* plan);                // allocates + fills the twiddle tables on the current device
void plan_destroy(FftPlan* plan);

struct RowStat
{
    float maxv;
    unsigned idx;
    float sum;
    float second;  // on-chip path: second peak of the row with +-samples_per_chip around its peak blanked
};

struct DevAcqResult
{
    unsigned index_time;
    unsigned index_doppler;
    float peak;
    float input_power;
    float second_peak;
    float test_statistics;
};

// ---- launches (all asynchronous on `s`) ----------------------------------------------------------------
// Forward transform of `batch` sequences into the permuted-spectrum layout [k1][k2] (k = k1 + n1*k2).
//   src: batch sequences of n_in complex samples, src_stride apart (0 = all batches read the same sequence),
//        placed at [place_off, place_off + n_in) inside a zero-padded length-n buffer (acq.cc:230-247,657-664);
//   wipe_hz != nullptr: sequence b is multiplied by exp(-j 2 pi wipe_hz[b] n / fs) on load (acq.cc:275-281,531).
//   tmp: batch * n complex scratch; dst: batch * n complex.
int fft_forward(const FftPlan& p, const float2* src, size_t src_stride, int n_in, int place_off, const float* wipe_hz, double fs,
    float2* tmp, float2* dst, int batch, hipStream_t s, int fold = 1);

// Circular correlation of n_prn code spectra with n_bins signal spectra (both in permuted layout) and
// |.|^2 into grid[prn][bin][effective]: y = IFFT(X_bin * conj(FFT(code_prn))), unnormalised (acq.cc:538-553).
//   spectra: n_bins * n; codes: n_prn * n (UNconjugated forward FFT of the placed code);
//   tmp: n_prn * n_bins * n complex scratch; grid rows are `effective` floats, taken from y[grid_off .. grid_off+effective).
int correlate_grid(const FftPlan& p, const float2* spectra, const float2* codes, float2* tmp, float* grid, int n_prn, int n_bins,
    int grid_off, int effective, int accumulate, float weight, hipStream_t s);

// per-row (max, lowest arg-max, sum) then the two statistics of acq.cc:409-519 per PRN
int grid_statistics(const float* grid, RowStat* rows, DevAcqResult* results, int n_prn, int n_bins, int effective,
    int samples_per_chip, int use_cfar, unsigned dwell_count, hipStream_t s);

// ---- whole-transform-on-chip path (pcps_onchip.hip): lengths with a plan in fft_onchip.h -------------------
// Spectra are in NATURAL order here (the four-step path above uses its permuted [k1][k2] layout).
bool onchip_supported(int n);
// N = S * M with M planned (GSH_OC_SPLIT_PLANS): S, or 0.  A cell is then S independent work-groups; no second-peak statistic, no folding.
int onchip_split(int n);
// split plans that run decimation in time (pcps_onchip.hip, oc_subcell_dit_kernel + oc_combine_dit_kernel): their spectra -- signal and code, everything
// onchip_forward writes for this length -- are stored residue-major, and onchip_correlate needs the scratch `z` (n_prn * n_bins * n values)
bool onchip_dit(int n);
// Z of a decimation-in-time split into `split` sub-cells is accessed with the non-temporal hint, and the batches of the pipelined loop run their cells in order
bool onchip_dit_nontemporal(int split);
int onchip_forward(int n, const float2* src, size_t src_stride, int n_in, int place_off, const float* wipe_hz, double fs, float2* dst,
    int batch, hipStream_t s, int fold = 1);
// one work-group per (PRN, bin) cell: spectrum product, inverse transform, |.|^2, row statistics, and (by the last cell
// of each PRN, counted in `arrivals`, n_prn zero-initialised counters) the PRN's statistic into `results`; the grid is
// read only when `accumulate` and written only when `store_grid`
// lags [offset, offset + effective) of the transform enter the search (offset + effective == n; offset != 0: bit_transition_flag);
// subrows: n_prn * n_bins * onchip_split(n) records (split plans only, else nullptr)
// waverows: n_prn * n_bins * max(onchip_split(n), 1) * ONCHIP_MAX_WAVES per-wave partial records: every flavour but the single-transform peak-ratio search lets the
// cells' waves leave their partials there and forms rows and statistic in a small kernel queued behind the cells (no barrier, release or ticket at the end of a cell)
constexpr int ONCHIP_MAX_WAVES = 16;
int onchip_correlate(int n, const float2* spectra, const float2* codes, float* grid, RowStat* rows, RowStat* subrows, DevAcqResult* results,
    unsigned* arrivals, int n_prn, int n_bins, int offset, int effective, int accumulate, int store_grid, int samples_per_chip, int use_cfar,
    unsigned dwell_count, float weight, hipStream_t s, float2* z = nullptr, RowStat* waverows = nullptr);
}  // namespace gsh
#endif
