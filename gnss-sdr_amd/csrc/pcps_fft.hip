// PCPS acquisition kernels for MI355X (gfx950): batched mixed-radix FFT correlation + peak statistics.
//
// Replaces the arithmetic of pcps_acquisition (gnss-sdr, src/algorithms/acquisition/gnuradio_blocks/
// pcps_acquisition.cc): set_local_code :218-251, update_local_carrier :275-281, doppler_grid :522-560,
// max_to_input_power_statistic :409-449, first_vs_second_peak_statistic :452-519 -- whose FFTs live in
// GNU Radio/FFTW and whose element-wise passes live in VOLK.  Nothing here is derived from those libraries.
//
// Transform structure.  N = n1*n2 (e.g. 25 000 = 125 * 200).  With n = n2' + n2*n1' and k = k1 + n1*k2
//   X[k1 + n1 k2] = sum_{n2'} W_n2^{k2 n2'} * W_N^{k1 n2'} * sum_{n1'} x[n1' n2 + n2'] W_n1^{k1 n1'}
// so a forward transform is a "column pass" (n2 strided length-n1 transforms, twiddled by W_N^{k1 n2'}) followed by
// a "row pass" (n1 contiguous length-n2 transforms), and the result is left in the PERMUTED layout [k1][k2].
// Signal spectra and code spectra share that layout, so their product is element-wise, and the inverse runs the
// two passes in the opposite order (rows, twiddle, columns) and lands in natural order -- no transposes at all.
// Both passes keep their sub-transforms in LDS (Stockham autosort, radix 2/3/4/5/8 butterflies in registers, a
// generic O(r^2) butterfly for other primes <= 31), read and write HBM in >=64-byte contiguous runs, and fuse
// the neighbouring element-wise work: Doppler wipe-off on load (never stored: the reference keeps D tables of N
// complex values), spectrum product on load, |.|^2 (+ non-coherent accumulation) on store.
// The inverse uses IFFT(Y) = conj(FFT(conj(Y))); the final conj is dropped because only |y|^2 is kept.
#include "pcps_fft.h"
#include <cmath>

namespace gsh
{
namespace
{
constexpr int FFT_THREADS = 256;
constexpr int MAX_GENERIC_RADIX = 61;  // 6.625 Msps (ten of the reference's example configurations) is 6625 = 5^3 * 53 samples per ms
constexpr int MAX_SUB_LEN_COLS = 1024;  // column-pass sub-transform (x tile_cols x 16 B of LDS)
constexpr int MAX_SUB_LEN_ROWS = 2048;  // row-pass sub-transform
constexpr int LDS_BUDGET = 64 * 1024;

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -j
__device__ __forceinline__ float2 mulmj(float2 a) { return make_float2(a.y, -a.x); }

// ---- forward DFT butterflies, in place: a[k] <- sum_j a[j] exp(-2 pi i j k / R)
template <int R>
struct Butterfly;

template <>
struct Butterfly<2>
{
    static __device__ __forceinline__ void run(float2 (&a)[2])
    {
        const float2 t = a[0];
        a[0] = cadd(t, a[1]);
        a[1] = csub(t, a[1]);
    }
};

template <>
struct Butterfly<3>
{
    static __device__ __forceinline__ void run(float2 (&a)[3])
    {
        constexpr float S = 0.86602540378443864676f;  // sin(2 pi / 3)
        const float2 t = cadd(a[1], a[2]);
        const float2 d = csub(a[1], a[2]);
        const float2 u = make_float2(fmaf(-0.5f, t.x, a[0].x), fmaf(-0.5f, t.y, a[0].y));
        const float2 v = make_float2(S * d.y, -S * d.x);  // -j * S * d
        a[0] = cadd(a[0], t);
        a[1] = cadd(u, v);
        a[2] = csub(u, v);
    }
};

template <>
struct Butterfly<4>
{
    static __device__ __forceinline__ void run(float2 (&a)[4])
    {
        const float2 s02 = cadd(a[0], a[2]), d02 = csub(a[0], a[2]);
        const float2 s13 = cadd(a[1], a[3]), d13 = mulmj(csub(a[1], a[3]));
        a[0] = cadd(s02, s13);
        a[2] = csub(s02, s13);
        a[1] = cadd(d02, d13);
        a[3] = csub(d02, d13);
    }
};

template <>
struct Butterfly<5>
{
    static __device__ __forceinline__ void run(float2 (&a)[5])
    {
        constexpr float C1 = 0.30901699437494742410f;   // cos(2 pi / 5)
        constexpr float C2 = -0.80901699437494742410f;  // cos(4 pi / 5)
        constexpr float S1 = 0.95105651629515357212f;   // sin(2 pi / 5)
        constexpr float S2 = 0.58778525229247312917f;   // sin(4 pi / 5)
        const float2 t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]);
        const float2 t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
        const float2 m1 = make_float2(a[0].x + C1 * t1.x + C2 * t2.x, a[0].y + C1 * t1.y + C2 * t2.y);
        const float2 m2 = make_float2(a[0].x + C2 * t1.x + C1 * t2.x, a[0].y + C2 * t1.y + C1 * t2.y);
        const float2 n1 = make_float2(S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y);
        const float2 n2 = make_float2(S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y);
        a[0] = cadd(a[0], cadd(t1, t2));
        // X1 = m1 - j n1, X4 = m1 + j n1, X2 = m2 - j n2, X3 = m2 + j n2
        a[1] = make_float2(m1.x + n1.y, m1.y - n1.x);
        a[4] = make_float2(m1.x - n1.y, m1.y + n1.x);
        a[2] = make_float2(m2.x + n2.y, m2.y - n2.x);
        a[3] = make_float2(m2.x - n2.y, m2.y + n2.x);
    }
};

template <>
struct Butterfly<8>
{
    static __device__ __forceinline__ void run(float2 (&a)[8])
    {
        constexpr float H = 0.70710678118654752440f;
        float2 e[4] = {a[0], a[2], a[4], a[6]};
        float2 o[4] = {a[1], a[3], a[5], a[7]};
        Butterfly<4>::run(e);
        Butterfly<4>::run(o);
        // W8^1 = (1 - j)/sqrt2, W8^2 = -j, W8^3 = (-1 - j)/sqrt2
        const float2 o1 = make_float2(H * (o[1].x + o[1].y), H * (o[1].y - o[1].x));
        const float2 o2 = mulmj(o[2]);
        const float2 o3 = make_float2(H * (o[3].y - o[3].x), -H * (o[3].x + o[3].y));
        a[0] = cadd(e[0], o[0]);
        a[4] = csub(e[0], o[0]);
        a[1] = cadd(e[1], o1);
        a[5] = csub(e[1], o1);
        a[2] = cadd(e[2], o2);
        a[6] = csub(e[2], o2);
        a[3] = cadd(e[3], o3);
        a[7] = csub(e[3], o3);
    }
};

// One Stockham pass over n_inst interleaved-by-s0 sequences of length len held in LDS.
// Element (inst, idx, lane) lives at inst*len*s0 + idx*s0 + lane; `s` already includes s0.
//   in:  x[q + s*(p + m*j)],  j < R      out: y[q + s*(R*p + k)] = DFT_R(in)[k] * W_n^{p k},  n = R*m
// tw = exp(-2 pi i e / len) table in LDS; W_n^{e} = tw[e * (len / n)].
template <int R>
__device__ __forceinline__ void stockham_pass(const float2* __restrict__ x, float2* __restrict__ y, const float2* __restrict__ tw,
    int len_s0, int n_inst, int m, int s, int tw_stride)
{
    const int per_inst = m * s;
    const int total = per_inst * n_inst;
    for (int b = threadIdx.x; b < total; b += FFT_THREADS)
        {
            const int inst = b / per_inst;
            const int rem = b - inst * per_inst;
            const int p = rem / s;
            const int q = rem - p * s;
            const float2* xi = x + inst * len_s0 + q + s * p;
            float2* yo = y + inst * len_s0 + q + s * R * p;
            float2 a[R];
#pragma unroll
            for (int j = 0; j < R; j++) a[j] = xi[s * m * j];
            Butterfly<R>::run(a);
            yo[0] = a[0];
#pragma unroll
            for (int k = 1; k < R; k++) yo[s * k] = cmul(a[k], tw[p * k * tw_stride]);
        }
}

// generic prime radix (7, 11, 13, ... 61): O(r^2) butterfly, root powers from the twiddle table
__device__ __forceinline__ void stockham_pass_generic(int r, const float2* __restrict__ x, float2* __restrict__ y,
    const float2* __restrict__ tw, int len, int len_s0, int n_inst, int m, int s, int tw_stride)
{
    const int per_inst = m * s;
    const int total = per_inst * n_inst;
    const int root_stride = len / r;  // exp(-2 pi i e / r) = tw[(e mod r) * root_stride]
    for (int b = threadIdx.x; b < total; b += FFT_THREADS)
        {
            const int inst = b / per_inst;
            const int rem = b - inst * per_inst;
            const int p = rem / s;
            const int q = rem - p * s;
            const float2* xi = x + inst * len_s0 + q + s * p;
            float2* yo = y + inst * len_s0 + q + s * r * p;
            // odd prime r: with s_j = a_j + a_{r-j}, d_j = a_j - a_{r-j} (j = 1 .. h = (r-1)/2)
            //   X[k], X[r-k] = (a_0 + sum_j s_j cos(2 pi j k / r))  -+  i (sum_j d_j sin(2 pi j k / r))
            // -- real coefficients times complex values: half the multiplies and half the table reads of r^2 complex products.
            // s_j and d_j are formed from the pass's INPUT buffer (LDS, untouched during the pass) wherever they are used: a per-thread array of them indexed
            // by a run-time j lived in scratch (496 B per thread until round 4 -- private memory behind the vector memory path, slower than the LDS reads that
            // replace it); the sums are the same operations on the same operands, so the results are bit for bit what they were.
            const int h = (r - 1) >> 1;
            const int sm = s * m;
            const float2 a0 = xi[0];
            float2 sum = a0;
            for (int j = 1; j <= h; j++) sum = cadd(sum, cadd(xi[sm * j], xi[sm * (r - j)]));
            yo[0] = sum;
            for (int k = 1; k <= h; k++)
                {
                    float2 pp = a0, qq = make_float2(0.0f, 0.0f);
                    int e = 0;
                    for (int j = 1; j <= h; j++)
                        {
                            e += k;
                            if (e >= r) e -= r;
                            const float2 w = tw[e * root_stride];  // (cos, -sin) of 2 pi e / r
                            const float2 u = xi[sm * j], v = xi[sm * (r - j)];
                            const float2 sj = cadd(u, v);
                            const float2 dj = make_float2(u.x - v.x, u.y - v.y);
                            pp.x = fmaf(sj.x, w.x, pp.x);
                            pp.y = fmaf(sj.y, w.x, pp.y);
                            qq.x = fmaf(dj.x, w.y, qq.x);  // w.y = -sin: qq = -sum d sin
                            qq.y = fmaf(dj.y, w.y, qq.y);
                        }
                    // X[k] = P - i Q with Q = sum d sin = -qq  ->  P + i qq = (P.x - qq.y, P.y + qq.x);  X[r-k] = P - i qq
                    const float2 xk = make_float2(pp.x - qq.y, pp.y + qq.x);
                    const float2 xrk = make_float2(pp.x + qq.y, pp.y - qq.x);
                    yo[s * k] = cmul(xk, tw[p * k * tw_stride]);
                    yo[s * (r - k)] = cmul(xrk, tw[p * (r - k) * tw_stride]);
                }
        }
}

// Full sub-transform; returns the buffer that holds the result (x or y).
__device__ __forceinline__ float2* lds_fft(float2* x, float2* y, const float2* tw, const SubPlan& sp, int s0, int n_inst)
{
    int n = sp.len;
    int s = s0;
    const int len_s0 = sp.len * s0;
    for (int pass = 0; pass < sp.n_pass; pass++)
        {
            const int r = sp.radix[pass];
            const int m = n / r;
            const int tw_stride = sp.len / n;
            switch (r)
                {
                case 2:
                    stockham_pass<2>(x, y, tw, len_s0, n_inst, m, s, tw_stride);
                    break;
                case 3:
                    stockham_pass<3>(x, y, tw, len_s0, n_inst, m, s, tw_stride);
                    break;
                case 4:
                    stockham_pass<4>(x, y, tw, len_s0, n_inst, m, s, tw_stride);
                    break;
                case 5:
                    stockham_pass<5>(x, y, tw, len_s0, n_inst, m, s, tw_stride);
                    break;
                case 8:
                    stockham_pass<8>(x, y, tw, len_s0, n_inst, m, s, tw_stride);
                    break;
                default:
                    stockham_pass_generic(r, x, y, tw, sp.len, len_s0, n_inst, m, s, tw_stride);
                    break;
                }
            __syncthreads();
            float2* t = x;
            x = y;
            y = t;
            n = m;
            s *= r;
        }
    return x;
}

// exp(-j 2 pi f n / fs), exact phase: the product f*n is exact in double, the fraction is reduced before the
// float sincos (the reference accumulates the phase in float32, K/volk_gnsssdr_s32f_sincos_32fc.h:390-400)
__device__ __forceinline__ float2 wipeoff(float f_hz, int n, double inv_fs)
{
    double rev = static_cast<double>(f_hz) * static_cast<double>(n) * inv_fs;
    rev -= rint(rev);
    float s, c;
    sincospif(static_cast<float>(2.0 * rev), &s, &c);
    return make_float2(c, -s);
}

// ------------------------------------------------------------------------------------------------------------
// column pass of the FORWARD transform: wipe-off (optional) -> length-n1 transforms -> * W_N^{k1 n2'} -> tmp[b][k1][n2']
// ------------------------------------------------------------------------------------------------------------
struct FwdColsArgs
{
    const float2* src;
    size_t src_stride;
    int n_in;
    int place_off;
    const float* wipe_hz;
    double inv_fs;
    float2* dst;
    const float2* tw_n;
    const float2* tw_1;
    int n1, n2, tile;
    int fold;  // > 1: the wiped-off input is summed over `fold` segments of n1*n2 samples (pcps_quicksync_acquisition_cc.cc:243-263)
    SubPlan sp;
};

__global__ __launch_bounds__(FFT_THREADS) void fwd_cols_kernel(FwdColsArgs a)
{
    extern __shared__ __align__(16) float2 lds2[];
    const int tile = a.tile;
    float2* bx = lds2;
    float2* by = lds2 + a.n1 * tile;
    float2* tw = lds2 + 2 * a.n1 * tile;
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * tile;
    const float2* __restrict__ src = a.src + static_cast<size_t>(b) * a.src_stride;
    const bool wipe = a.wipe_hz != nullptr;
    const float f_hz = wipe ? a.wipe_hz[b] : 0.0f;
    for (int i = threadIdx.x; i < a.n1; i += FFT_THREADS) tw[i] = a.tw_1[i];
    const int elems = a.n1 * tile;
    for (int i = threadIdx.x; i < elems; i += FFT_THREADS)
        {
            const int r = i / tile, c = i - r * tile;
            const int col = c0 + c;
            float2 v = make_float2(0.0f, 0.0f);
            if (col < a.n2)
                {
                    const int n = r * a.n2 + col;
                    const int k = n - a.place_off;
                    if (a.fold > 1)
                        {
                            // quicksync.cc:251-263: product with the wipe-off first, then the segments are added in order
                            const int len = a.n1 * a.n2;
                            for (int seg = 0; seg < a.fold; seg++)
                                {
                                    const int ks = k + seg * len;
                                    if (ks < 0 || ks >= a.n_in) continue;
                                    float2 u = src[ks];
                                    if (wipe) u = cmul(u, wipeoff(f_hz, n + seg * len, a.inv_fs));
                                    v.x += u.x;
                                    v.y += u.y;
                                }
                        }
                    else if (k >= 0 && k < a.n_in)
                        {
                            v = src[k];
                            if (wipe) v = cmul(v, wipeoff(f_hz, n, a.inv_fs));
                        }
                }
            bx[i] = v;
        }
    __syncthreads();
    const float2* res = lds_fft(bx, by, tw, a.sp, tile, 1);
    float2* __restrict__ dst = a.dst + static_cast<size_t>(b) * a.n1 * a.n2;
    for (int i = threadIdx.x; i < elems; i += FFT_THREADS)
        {
            const int k1 = i / tile, c = i - k1 * tile;
            const int col = c0 + c;
            if (col < a.n2) dst[static_cast<size_t>(k1) * a.n2 + col] = cmul(res[i], a.tw_n[k1 * col]);
        }
}

// ------------------------------------------------------------------------------------------------------------
// row pass.  MODE 0: plain (second half of the forward transform):  dst[b][k1][:] = FFT_n2(src[b][k1][:])
//            MODE 1: first half of the inverse: v = conj(spectra[bin][k1][:]) * codes[prn][k1][:],
//                    dst[cell][k1][n2'] = FFT_n2(v)[n2'] * W_N^{k1 n2'}     (cell = prn*n_bins + bin)
// ------------------------------------------------------------------------------------------------------------
struct RowsArgs
{
    const float2* src;    // MODE 0: batch*n ; MODE 1: spectra (n_bins*n)
    const float2* codes;  // MODE 1 only
    float2* dst;
    const float2* tw_n;
    const float2* tw_2;
    int n1, n2, rows_per_wg, n_bins;
    SubPlan sp;
};

template <int MODE>
__global__ __launch_bounds__(FFT_THREADS) void rows_kernel(RowsArgs a)
{
    extern __shared__ __align__(16) float2 lds2[];
    const int rows = a.rows_per_wg;
    float2* bx = lds2;
    float2* by = lds2 + rows * a.n2;
    float2* tw = lds2 + 2 * rows * a.n2;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows;
    const int nrows = min(rows, a.n1 - r0);
    for (int i = threadIdx.x; i < a.n2; i += FFT_THREADS) tw[i] = a.tw_2[i];
    const size_t n = static_cast<size_t>(a.n1) * a.n2;
    const int elems = nrows * a.n2;
    if (MODE == 0)
        {
            const float2* __restrict__ src = a.src + static_cast<size_t>(b) * n + static_cast<size_t>(r0) * a.n2;
            for (int i = threadIdx.x; i < elems; i += FFT_THREADS) bx[i] = src[i];
        }
    else
        {
            const int prn = b / a.n_bins, bin = b - prn * a.n_bins;
            const float2* __restrict__ sx = a.src + static_cast<size_t>(bin) * n + static_cast<size_t>(r0) * a.n2;
            const float2* __restrict__ sc = a.codes + static_cast<size_t>(prn) * n + static_cast<size_t>(r0) * a.n2;
            for (int i = threadIdx.x; i < elems; i += FFT_THREADS)
                {
                    const float2 x = sx[i], c = sc[i];
                    // conj(x) * c
                    bx[i] = make_float2(fmaf(x.x, c.x, x.y * c.y), fmaf(x.x, c.y, -(x.y * c.x)));
                }
        }
    __syncthreads();
    const float2* res = lds_fft(bx, by, tw, a.sp, 1, nrows);
    float2* __restrict__ dst = a.dst + static_cast<size_t>(b) * n + static_cast<size_t>(r0) * a.n2;
    if (MODE == 0)
        {
            for (int i = threadIdx.x; i < elems; i += FFT_THREADS) dst[i] = res[i];
        }
    else
        {
            for (int i = threadIdx.x; i < elems; i += FFT_THREADS)
                {
                    const int r = i / a.n2, c = i - r * a.n2;
                    dst[i] = cmul(res[i], a.tw_n[(r0 + r) * c]);
                }
        }
}

// ------------------------------------------------------------------------------------------------------------
// column pass of the INVERSE: length-n1 transforms over k1 for a tile of n2' columns, then |.|^2 into the grid
// (natural order tau = n1' * n2 + n2').  The conj of conj(FFT(conj Y)) is dropped: |.|^2 does not see it.
// ------------------------------------------------------------------------------------------------------------
struct InvColsArgs
{
    const float2* src;  // cells * n, layout [k1][n2']
    float* grid;        // cells * effective
    const float2* tw_1;
    int n1, n2, tile;
    int grid_off, effective, accumulate;
    float weight;  // every |.|^2 is scaled before it reaches the grid (pcps_tong_acquisition_cc.cc:243-249); 1 otherwise
    SubPlan sp;
};

__global__ __launch_bounds__(FFT_THREADS) void inv_cols_kernel(InvColsArgs a)
{
    extern __shared__ __align__(16) float2 lds2[];
    const int tile = a.tile;
    float2* bx = lds2;
    float2* by = lds2 + a.n1 * tile;
    float2* tw = lds2 + 2 * a.n1 * tile;
    const int cell = blockIdx.y;
    const int c0 = blockIdx.x * tile;
    const float2* __restrict__ src = a.src + static_cast<size_t>(cell) * a.n1 * a.n2;
    for (int i = threadIdx.x; i < a.n1; i += FFT_THREADS) tw[i] = a.tw_1[i];
    const int elems = a.n1 * tile;
    for (int i = threadIdx.x; i < elems; i += FFT_THREADS)
        {
            const int r = i / tile, c = i - r * tile;
            const int col = c0 + c;
            bx[i] = (col < a.n2) ? src[static_cast<size_t>(r) * a.n2 + col] : make_float2(0.0f, 0.0f);
        }
    __syncthreads();
    const float2* res = lds_fft(bx, by, tw, a.sp, tile, 1);
    float* __restrict__ grid = a.grid + static_cast<size_t>(cell) * a.effective;
    for (int i = threadIdx.x; i < elems; i += FFT_THREADS)
        {
            const int r = i / tile, c = i - r * tile;
            const int col = c0 + c;
            if (col >= a.n2) continue;
            const int tau = r * a.n2 + col - a.grid_off;
            if (tau < 0 || tau >= a.effective) continue;
            const float2 v = res[i];
            const float mag = fmaf(v.x, v.x, v.y * v.y) * a.weight;
            grid[tau] = a.accumulate ? grid[tau] + mag : mag;
        }
}

// ------------------------------------------------------------------------------------------------------------
// statistics
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void argmax_combine(float& v, unsigned& i, float ov, unsigned oi)
{
    // lowest index wins ties (K/volk_gnsssdr_32f_index_max_32u.h:457 uses '>' in ascending order)
    if (ov > v || (ov == v && oi < i))
        {
            v = ov;
            i = oi;
        }
}

__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ grid, RowStat* __restrict__ rows, int effective)
{
    __shared__ float s_v[4];
    __shared__ unsigned s_i[4];
    __shared__ float s_s[4];
    const float* __restrict__ g = grid + static_cast<size_t>(blockIdx.x) * effective;
    float best = -1.0f;
    unsigned at = 0xFFFFFFFFu;
    float sum = 0.0f;
    for (int i = threadIdx.x; i < effective; i += 256)
        {
            const float v = g[i];
            sum += v;
            if (v > best)
                {
                    best = v;
                    at = static_cast<unsigned>(i);
                }
        }
    for (int off = 32; off > 0; off >>= 1)
        {
            const float ov = __shfl_down(best, off, 64);
            const unsigned oi = __shfl_down(at, off, 64);
            sum += __shfl_down(sum, off, 64);
            argmax_combine(best, at, ov, oi);
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        {
            s_v[wave] = best;
            s_i[wave] = at;
            s_s[wave] = sum;
        }
    __syncthreads();
    if (threadIdx.x == 0)
        {
            for (int w = 1; w < 4; w++)
                {
                    argmax_combine(best, at, s_v[w], s_i[w]);
                    sum += s_s[w];
                }
            RowStat r;
            r.maxv = best;
            r.idx = at;
            r.sum = sum;
            r.second = 0.0f;
            rows[blockIdx.x] = r;
        }
}

// one work-group per PRN: the bin scan of acq.cc:417-426 / :463-474, then either statistic
__global__ __launch_bounds__(256) void final_stats_kernel(const float* __restrict__ grid, const RowStat* __restrict__ rows,
    DevAcqResult* __restrict__ results, int n_bins, int effective, int samples_per_chip, int use_cfar, unsigned dwell_count)
{
    __shared__ unsigned s_bin, s_tau;
    __shared__ float s_peak;
    __shared__ float s_w[4];
    const int prn = blockIdx.x;
    const RowStat* __restrict__ rs = rows + static_cast<size_t>(prn) * n_bins;
    if (threadIdx.x == 0)
        {
            float gmax = 0.0f;  // acq.cc:412 / :459: starts at 0, strict '>' keeps the first bin on ties
            unsigned bin = 0, tau = 0;
            for (int d = 0; d < n_bins; d++)
                {
                    if (rs[d].maxv > gmax)
                        {
                            gmax = rs[d].maxv;
                            bin = static_cast<unsigned>(d);
                            tau = rs[d].idx;
                        }
                }
            s_bin = bin;
            s_tau = tau;
            s_peak = gmax;
        }
    __syncthreads();
    const unsigned bin = s_bin, tau = s_tau;
    const float peak = s_peak;
    DevAcqResult out;
    out.index_time = tau;
    out.index_doppler = bin;
    out.peak = peak;
    out.input_power = 0.0f;
    out.second_peak = 0.0f;
    out.test_statistics = 0.0f;
    if (use_cfar)
        {
            if (threadIdx.x == 0)
                {
                    // acq.cc:429-431: power of the bin half a grid away, / effective / 2 / dwells
                    const unsigned opp = (bin + static_cast<unsigned>(n_bins) / 2u) % static_cast<unsigned>(n_bins);
                    const float per_sample = rs[opp].sum / static_cast<float>(static_cast<unsigned>(effective));
                    const float power = static_cast<float>(static_cast<double>(per_sample) / 2.0 / static_cast<double>(dwell_count));
                    out.input_power = power;
                    out.test_statistics = (power < 1.1920928955078125e-07f) ? 0.0f : peak / power;  // acq.cc:438-445
                    results[prn] = out;
                }
            return;
        }
    // acq.cc:485-516: blank [tau - spc, tau + spc) cyclically in the winning row, take the maximum of what is left
    int e1 = static_cast<int>(tau) - samples_per_chip;
    int e2 = static_cast<int>(tau) + samples_per_chip;
    if (e1 < 0)
        e1 += effective;
    else if (e2 >= effective)
        e2 -= effective;
    const float* __restrict__ g = grid + (static_cast<size_t>(prn) * n_bins + bin) * effective;
    float second = 0.0f;  // blanked cells hold 0.0 and the do-while of :498-509 always blanks at least one
    for (int i = threadIdx.x; i < effective; i += 256)
        {
            bool blank;
            if (e1 < e2)
                blank = (i >= e1 && i < e2);
            else if (e1 > e2)
                blank = (i >= e1 || i < e2);
            else
                blank = true;
            if (!blank) second = fmaxf(second, g[i]);
        }
    for (int off = 32; off > 0; off >>= 1) second = fmaxf(second, __shfl_down(second, off, 64));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = second;
    __syncthreads();
    if (threadIdx.x == 0)
        {
            second = fmaxf(fmaxf(s_w[0], s_w[1]), fmaxf(s_w[2], s_w[3]));
            out.second_peak = second;
            out.test_statistics = peak / second;  // acq.cc:516
            results[prn] = out;
        }
}

bool is_prime_small(int v)
{
    for (int d = 2; d * d <= v; d++)
        if (v % d == 0) return false;
    return v >= 2;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------
// host side: planning
// ---------------------------------------------------------------------------------------------------------
bool factor_length(int len, SubPlan* out)
{
    SubPlan sp;
    sp.len = len;
    sp.n_pass = 0;
    int rem = len;
    auto push = [&](int r) {
        if (sp.n_pass >= FFT_MAX_PASSES) return false;
        sp.radix[sp.n_pass++] = r;
        return true;
    };
    // powers of two as 8s then a 4 or 2; the larger butterflies do more work per LDS round trip
    while (rem % 8 == 0)
        {
            if (!push(8)) return false;
            rem /= 8;
        }
    if (rem % 4 == 0)
        {
            if (!push(4)) return false;
            rem /= 4;
        }
    if (rem % 2 == 0)
        {
            if (!push(2)) return false;
            rem /= 2;
        }
    for (int r : {5, 3})
        while (rem % r == 0)
            {
                if (!push(r)) return false;
                rem /= r;
            }
    for (int r = 7; r <= MAX_GENERIC_RADIX && rem > 1; r += 2)
        {
            if (!is_prime_small(r)) continue;
            while (rem % r == 0)
                {
                    if (!push(r)) return false;
                    rem /= r;
                }
        }
    if (rem != 1) return false;
    if (out) *out = sp;
    return true;
}

bool choose_split(int n, int* n1_out, int* n2_out)
{
    // n1 = column-pass length, n2 = row-pass length; prefer the most balanced split with n1 <= n2
    int best1 = 0, best2 = 0;
    double best_score = 1e300;
    for (int a = 1; static_cast<long long>(a) * a <= n; a++)
        {
            if (n % a) continue;
            const int b = n / a;
            if (a > MAX_SUB_LEN_COLS || b > MAX_SUB_LEN_ROWS) continue;
            if (!factor_length(a, nullptr) || !factor_length(b, nullptr)) continue;
            const double score = std::log(static_cast<double>(b) / a);
            if (score < best_score)
                {
                    best_score = score;
                    best1 = a;
                    best2 = b;
                }
        }
    if (best1 == 0) return false;
    *n1_out = best1;
    *n2_out = best2;
    return true;
}

namespace
{
int make_table(int len, float2** d_out)
{
    std::vector<float2> h(static_cast<size_t>(len));
    const double w = -2.0 * 3.14159265358979323846264338327950288 / static_cast<double>(len);
    for (int j = 0; j < len; j++) h[j] = make_float2(static_cast<float>(std::cos(w * j)), static_cast<float>(std::sin(w * j)));
    GSH_HIP(hipMalloc(d_out, sizeof(float2) * static_cast<size_t>(len)));
    GSH_HIP(hipMemcpy(*d_out, h.data(), sizeof(float2) * static_cast<size_t>(len), hipMemcpyHostToDevice));
    return GSH_OK;
}
}  // namespace

int plan_create(int n, FftPlan* plan)
{
    GSH_REQUIRE(plan != nullptr, "null plan");
    GSH_REQUIRE(n >= 4, "fft_size %d too small", n);
    int n1 = 0, n2 = 0;
    if (!choose_split(n, &n1, &n2))
        return set_error(GSH_ERR_UNSUPPORTED,
            "fft_size %d has no n1*n2 split with n1 <= %d, n2 <= %d and prime factors <= %d", n, MAX_SUB_LEN_COLS, MAX_SUB_LEN_ROWS, MAX_GENERIC_RADIX);
    plan->n = n;
    plan->n1 = n1;
    plan->n2 = n2;
    factor_length(n1, &plan->p1);
    factor_length(n2, &plan->p2);
    // tile sizes from the LDS budget: two ping-pong buffers + the sub-transform's root table
    int tc = 16;
    while (tc > 1 && (2 * n1 * tc + n1) * static_cast<int>(sizeof(float2)) > LDS_BUDGET / 2) tc >>= 1;
    plan->tile_cols = tc;
    int tr = 8;
    while (tr > 1 && (2 * n2 * tr + n2) * static_cast<int>(sizeof(float2)) > LDS_BUDGET / 2) tr >>= 1;
    plan->tile_rows = tr;
    int rc = make_table(n, &plan->d_tw_n);
    if (rc == GSH_OK) rc = make_table(n1, &plan->d_tw_1);
    if (rc == GSH_OK) rc = make_table(n2, &plan->d_tw_2);
    if (rc != GSH_OK) plan_destroy(plan);
    return rc;
}

void plan_destroy(FftPlan* plan)
{
    if (!plan) return;
    if (plan->d_tw_n) (void)hipFree(plan->d_tw_n);
    if (plan->d_tw_1) (void)hipFree(plan->d_tw_1);
    if (plan->d_tw_2) (void)hipFree(plan->d_tw_2);
    plan->d_tw_n = plan->d_tw_1 = plan->d_tw_2 = nullptr;
}

// ---------------------------------------------------------------------------------------------------------
// host side: launches
// ---------------------------------------------------------------------------------------------------------
int fft_forward(const FftPlan& p, const float2* src, size_t src_stride, int n_in, int place_off, const float* wipe_hz, double fs,
    float2* tmp, float2* dst, int batch, hipStream_t s, int fold)
{
    if (batch <= 0) return GSH_OK;
    FwdColsArgs c;
    c.fold = fold;
    c.src = src;
    c.src_stride = src_stride;
    c.n_in = n_in;
    c.place_off = place_off;
    c.wipe_hz = wipe_hz;
    c.inv_fs = 1.0 / fs;
    c.dst = tmp;
    c.tw_n = p.d_tw_n;
    c.tw_1 = p.d_tw_1;
    c.n1 = p.n1;
    c.n2 = p.n2;
    c.tile = p.tile_cols;
    c.sp = p.p1;
    const size_t lds_c = (2 * static_cast<size_t>(p.n1) * p.tile_cols + p.n1) * sizeof(float2);
    hipLaunchKernelGGL(fwd_cols_kernel, dim3((p.n2 + p.tile_cols - 1) / p.tile_cols, batch), dim3(FFT_THREADS), lds_c, s, c);
    GSH_HIP(hipGetLastError());
    RowsArgs r;
    r.src = tmp;
    r.codes = nullptr;
    r.dst = dst;
    r.tw_n = p.d_tw_n;
    r.tw_2 = p.d_tw_2;
    r.n1 = p.n1;
    r.n2 = p.n2;
    r.rows_per_wg = p.tile_rows;
    r.n_bins = 1;
    r.sp = p.p2;
    const size_t lds_r = (2 * static_cast<size_t>(p.n2) * p.tile_rows + p.n2) * sizeof(float2);
    hipLaunchKernelGGL((rows_kernel<0>), dim3((p.n1 + p.tile_rows - 1) / p.tile_rows, batch), dim3(FFT_THREADS), lds_r, s, r);
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}

int correlate_grid(const FftPlan& p, const float2* spectra, const float2* codes, float2* tmp, float* grid, int n_prn, int n_bins,
    int grid_off, int effective, int accumulate, float weight, hipStream_t s)
{
    const int cells = n_prn * n_bins;
    if (cells <= 0) return GSH_OK;
    GSH_REQUIRE(cells <= 65535, "%d PRN x %d bins exceeds the 65535-cell launch limit; search fewer PRNs per dwell", n_prn, n_bins);
    RowsArgs r;
    r.src = spectra;
    r.codes = codes;
    r.dst = tmp;
    r.tw_n = p.d_tw_n;
    r.tw_2 = p.d_tw_2;
    r.n1 = p.n1;
    r.n2 = p.n2;
    r.rows_per_wg = p.tile_rows;
    r.n_bins = n_bins;
    r.sp = p.p2;
    const size_t lds_r = (2 * static_cast<size_t>(p.n2) * p.tile_rows + p.n2) * sizeof(float2);
    hipLaunchKernelGGL((rows_kernel<1>), dim3((p.n1 + p.tile_rows - 1) / p.tile_rows, cells), dim3(FFT_THREADS), lds_r, s, r);
    GSH_HIP(hipGetLastError());
    InvColsArgs c;
    c.src = tmp;
    c.grid = grid;
    c.tw_1 = p.d_tw_1;
    c.n1 = p.n1;
    c.n2 = p.n2;
    c.tile = p.tile_cols;
    c.grid_off = grid_off;
    c.effective = effective;
    c.accumulate = accumulate;
    c.weight = weight;
    c.sp = p.p1;
    const size_t lds_c = (2 * static_cast<size_t>(p.n1) * p.tile_cols + p.n1) * sizeof(float2);
    hipLaunchKernelGGL(inv_cols_kernel, dim3((p.n2 + p.tile_cols - 1) / p.tile_cols, cells), dim3(FFT_THREADS), lds_c, s, c);
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}

int grid_statistics(const float* grid, RowStat* rows, DevAcqResult* results, int n_prn, int n_bins, int effective,
    int samples_per_chip, int use_cfar, unsigned dwell_count, hipStream_t s)
{
    if (n_prn <= 0) return GSH_OK;
    hipLaunchKernelGGL(row_stats_kernel, dim3(n_prn * n_bins), dim3(256), 0, s, grid, rows, effective);
    GSH_HIP(hipGetLastError());
    hipLaunchKernelGGL(final_stats_kernel, dim3(n_prn), dim3(256), 0, s, grid, rows, results, n_bins, effective, samples_per_chip, use_cfar, dwell_count);
    GSH_HIP(hipGetLastError());
    return GSH_OK;
}
}  // namespace gsh
