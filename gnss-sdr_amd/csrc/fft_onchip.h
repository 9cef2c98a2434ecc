// Whole-transform-on-one-CU FFT building blocks for MI355X (gfx950): the length-N transform of one acquisition
// cell (pcps_acquisition.cc:538-541, the IFFT of X_bin * conj(FFT(code))) never leaves the compute unit.
//
//   N = R1 * R2 * R3 (25 000 = 25 * 25 * 40).  Every thread keeps R elements of the current stage in REGISTERS,
//   runs one radix-R DFT on them (composite radices built at compile time from radix 2/3/4/5/8 butterflies with
//   literal twiddles), multiplies by the inter-stage twiddles (one sincospi seed + a power tree per thread),
//   and the two re-distributions between the three stages go through LDS one float component at a time
//   (N * 4 bytes <= 160 KiB although N * 8 is not).  Complex values are clang ext-vectors so the arithmetic maps
//   to gfx950's packed FP32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).
//
// Index maps (decimation in frequency, natural order in AND out):
//   n = n1*(R2*R3) + n2*R3 + n3          k = k1 + R1*k2 + R1*R2*k3
//   stage 1: thread t1 = n2*R3 + n3  holds n1 = 0..R1-1 -> DFT_R1 -> * W_N^{k1*t1}
//   stage 2: thread t2 = k1*R3 + n3  holds n2 = 0..R2-1 -> DFT_R2 -> * W_{R2*R3}^{k2*n3}
//   stage 3: thread t3 = k1 + R1*k2  holds n3 = 0..R3-1 -> DFT_R3 -> X[t3 + R1*R2*k3]
// so both the first load and the last store are unit-stride across lanes.
//
// The per-thread phase functions are __host__ __device__: tests/host/fft_onchip_host.cc runs the very same code
// thread by thread on the CPU (no GPU needed) against numpy.fft.  Nothing here derives from FFTW, GNU Radio or
// VOLK (the libraries behind the reference's transform).
#ifndef GSH_FFT_ONCHIP_H
#define GSH_FFT_ONCHIP_H

#include <type_traits>
#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GSH_HD __host__ __device__ __forceinline__
#else
#define GSH_HD inline __attribute__((always_inline))
#endif
#define GSH_AI __attribute__((always_inline))
// The exchange reads fill ONE component of R complex registers.  Left alone, the compiler pairs neighbouring reads into
// ds_read2_b32, whose two results land in a register pair = the same component of two different elements, and then spends
// a v_mov per value to take the pair apart again.  Volatile keeps them single ds_read_b32 straight into place.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GSH_OC_NO_PK_ASM)
typedef const volatile __attribute__((address_space(3))) float* gsh_lds_rd_ptr;
#define GSH_LDS_RD_PTR(p) ((gsh_lds_rd_ptr)(p))
#else
typedef const float* gsh_lds_rd_ptr;
#define GSH_LDS_RD_PTR(p) (p)
#endif

#pragma clang fp contract(fast)

namespace gsh
{
namespace oc
{
typedef float cf __attribute__((ext_vector_type(2)));  // complex: .x = re, .y = im

// ------------------------------------------------------------------------------------------ compile-time loops
template <int... Is, class F>
GSH_HD void static_for_impl(std::integer_sequence<int, Is...>, F&& f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
GSH_HD void static_for(F&& f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// ------------------------------------------------------------------------------------------ compile-time trig
constexpr double PI_D = 3.141592653589793238462643383279502884;

constexpr double cx_sin_small(double x)  // |x| <= pi/4
{
    const double x2 = x * x;
    double term = x, sum = x;
    for (int i = 1; i < 14; i++)
        {
            term *= -x2 / static_cast<double>((2 * i) * (2 * i + 1));
            sum += term;
        }
    return sum;
}
constexpr double cx_cos_small(double x)  // |x| <= pi/4
{
    const double x2 = x * x;
    double term = 1.0, sum = 1.0;
    for (int i = 1; i < 14; i++)
        {
            term *= -x2 / static_cast<double>((2 * i - 1) * (2 * i));
            sum += term;
        }
    return sum;
}
// cos / sin of 2*pi*m/r with the octant picked in integer arithmetic (exact 0, +-1 and equal +-sqrt(1/2) pairs)
constexpr double cx_trig_turn(int m, int r, bool want_sin)
{
    m %= r;
    if (m < 0) m += r;
    const long long num = 8LL * m;
    const int oct = static_cast<int>(num / r);
    const long long rem = num - static_cast<long long>(oct) * r;
    const double th = (PI_D / 4.0) * static_cast<double>(rem) / static_cast<double>(r);          // [0, pi/4)
    const double tc = (PI_D / 4.0) * static_cast<double>(r - rem) / static_cast<double>(r);      // pi/4 - th
    double c = 0.0, s = 0.0;
    switch (oct)
        {
        case 0: c = cx_cos_small(th); s = cx_sin_small(th); break;
        case 1: c = cx_sin_small(tc); s = cx_cos_small(tc); break;
        case 2: c = -cx_sin_small(th); s = cx_cos_small(th); break;
        case 3: c = -cx_cos_small(tc); s = cx_sin_small(tc); break;
        case 4: c = -cx_cos_small(th); s = -cx_sin_small(th); break;
        case 5: c = -cx_sin_small(tc); s = -cx_cos_small(tc); break;
        case 6: c = cx_sin_small(th); s = -cx_cos_small(th); break;
        default: c = cx_cos_small(tc); s = -cx_sin_small(tc); break;
        }
    return want_sin ? s : c;
}

// ------------------------------------------------------------------------------------------ complex helpers
GSH_HD cf mulmj(cf a) { return cf{a.y, -a.x}; }  // * (-j)
GSH_HD cf mulpj(cf a) { return cf{-a.y, a.x}; }  // * (+j)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GSH_OC_NO_PK_ASM)
// gfx950's packed-FP32 instructions pick each source half (op_sel / op_sel_hi) and negate it (neg_lo / neg_hi) for free, so
// a complex product is two instructions and "+- j * d" folds into the add.  The compiler only finds the broadcast forms on its
// own (every swap costs it a v_mov + v_xor), hence the instruction selection is spelled out here.
//   lo result: src halves op_sel[i];  hi result: src halves op_sel_hi[i];  0 = .x, 1 = .y
GSH_HD cf cmul(cf a, cf b)
{
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));  // (ax bx, ax by)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));  // (-ay by, ay bx) + t
    return r;
}
GSH_HD cf csqr(cf a) { return cmul(a, a); }
// conj(a) * b
GSH_HD cf cmul_conj(cf a, cf b)
{
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));  // (ax bx, ax by)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));  // (ay by, -ay bx) + t
    return r;
}
// u + (-j) d  and  u - (-j) d
GSH_HD cf add_mj(cf u, cf d)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(u), "v"(d));  // (ux + dy, uy - dx)
    return r;
}
GSH_HD cf sub_mj(cf u, cf d)
{
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(u), "v"(d));  // (ux - dy, uy + dx)
    return r;
}
// v * (c - j s), cs = (c, s) a uniform constant
GSH_HD cf mul_cs(cf v, cf cs)
{
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(v), "s"(cs));  // (x c, y c)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(v), "s"(cs), "v"(t));  // (y s, -x s) + t
    return r;
}
// |v|^2 as two scalar instructions (left to itself the vectoriser pairs two elements' sums and shuffles registers to do so)
GSH_HD float norm2(cf v)
{
    float t, m;
    asm("v_mul_f32 %0, %1, %1" : "=v"(t) : "v"(v.y));
    asm("v_fma_f32 %0, %1, %1, %2" : "=v"(m) : "v"(v.x), "v"(t));
    return m;
}
#else
GSH_HD float norm2(cf v) { return v.x * v.x + v.y * v.y; }
GSH_HD cf cmul(cf a, cf b) { return a.xx * b + a.yy * cf{-b.y, b.x}; }
GSH_HD cf csqr(cf a) { return cf{a.x * a.x - a.y * a.y, 2.0f * a.x * a.y}; }
// conj(a) * b
GSH_HD cf cmul_conj(cf a, cf b) { return a.xx * b + a.yy * cf{b.y, -b.x}; }
GSH_HD cf add_mj(cf u, cf d) { return u + mulmj(d); }
GSH_HD cf sub_mj(cf u, cf d) { return u - mulmj(d); }
GSH_HD cf mul_cs(cf v, cf cs) { return v * cs.x + cf{v.y, -v.x} * cs.y; }
#endif

// v * W_R^M, W_R = exp(-2 pi i / R), with the trivial rotations specialised
template <int M, int R>
GSH_HD cf mul_w(cf v)
{
    constexpr int m = ((M % R) + R) % R;
    constexpr float H = 0.70710678118654752440f;
    if constexpr (m == 0)
        return v;
    else if constexpr (4 * m == R)
        return mulmj(v);
    else if constexpr (2 * m == R)
        return -v;
    else if constexpr (4 * m == 3 * R)
        return mulpj(v);
    else if constexpr (8 * m == R)
        return add_mj(v, v) * H;  // (x + y, y - x) / sqrt 2
    else if constexpr (8 * m == 3 * R)
        return mulmj(add_mj(v, v)) * H;
    else if constexpr (8 * m == 5 * R)
        return add_mj(v, v) * -H;
    else if constexpr (8 * m == 7 * R)
        return sub_mj(v, v) * H;  // (x - y, y + x) / sqrt 2
    else
        {
            constexpr float c = static_cast<float>(cx_trig_turn(m, R, false));
            constexpr float s = static_cast<float>(cx_trig_turn(m, R, true));
            // (x + jy)(c - js) = (xc + ys) + j(yc - xs)
            return mul_cs(v, cf{c, s});
        }
}

// ------------------------------------------------------------------------------------------ register DFTs
// Dft<R>::run(a): a[k] <- sum_j a[j] exp(-2 pi i j k / R), natural order in and out, everything in registers.
template <int R>
struct Dft;

template <>
struct Dft<1>
{
    static GSH_HD void run(cf (&)[1]) {}
};

template <>
struct Dft<2>
{
    static GSH_HD void run(cf (&a)[2])
    {
        const cf t = a[0];
        a[0] = t + a[1];
        a[1] = t - a[1];
    }
};

template <>
struct Dft<3>
{
    static GSH_HD void run(cf (&a)[3])
    {
        constexpr float S = 0.86602540378443864676f;  // sin(2 pi / 3)
        const cf t = a[1] + a[2];
        const cf d = (a[1] - a[2]) * S;
        const cf u = a[0] - t * 0.5f;
        a[0] = a[0] + t;
        a[1] = add_mj(u, d);
        a[2] = sub_mj(u, d);
    }
};

template <>
struct Dft<4>
{
    static GSH_HD void run(cf (&a)[4])
    {
        const cf s02 = a[0] + a[2], d02 = a[0] - a[2];
        const cf s13 = a[1] + a[3], d13 = a[1] - a[3];
        a[0] = s02 + s13;
        a[2] = s02 - s13;
        a[1] = add_mj(d02, d13);
        a[3] = sub_mj(d02, d13);
    }
};

template <>
struct Dft<5>
{
    static GSH_HD void run(cf (&a)[5])
    {
        constexpr float C1 = 0.30901699437494742410f;   // cos(2 pi / 5)
        constexpr float C2 = -0.80901699437494742410f;  // cos(4 pi / 5)
        constexpr float S1 = 0.95105651629515357212f;   // sin(2 pi / 5)
        constexpr float S2 = 0.58778525229247312917f;   // sin(4 pi / 5)
        const cf t1 = a[1] + a[4], t2 = a[2] + a[3];
        const cf t3 = a[1] - a[4], t4 = a[2] - a[3];
        const cf m1 = a[0] + t1 * C1 + t2 * C2;
        const cf m2 = a[0] + t1 * C2 + t2 * C1;
        const cf n1 = t3 * S1 + t4 * S2;  // times -j below
        const cf n2 = t3 * S2 - t4 * S1;
        a[0] = a[0] + t1 + t2;
        a[1] = add_mj(m1, n1);
        a[4] = sub_mj(m1, n1);
        a[2] = add_mj(m2, n2);
        a[3] = sub_mj(m2, n2);
    }
};

template <>
struct Dft<8>
{
    static GSH_HD void run(cf (&a)[8])
    {
        cf e[4] = {a[0], a[2], a[4], a[6]};
        cf o[4] = {a[1], a[3], a[5], a[7]};
        Dft<4>::run(e);
        Dft<4>::run(o);
        const cf o1 = mul_w<1, 8>(o[1]);
        const cf q3 = mul_w<1, 8>(o[3]);  // o[3] W_8^3 = -j q3, folded into the adds below
        a[0] = e[0] + o[0];
        a[4] = e[0] - o[0];
        a[1] = e[1] + o1;
        a[5] = e[1] - o1;
        a[2] = add_mj(e[2], o[2]);
        a[6] = sub_mj(e[2], o[2]);
        a[3] = add_mj(e[3], q3);
        a[7] = sub_mj(e[3], q3);
    }
};

// Odd prime radix R (7, 11, 13, 31: the 2.046 Msps family is 2 * 3 * 11 * 31).  With s_j = a_j + a_{R-j}, d_j = a_j - a_{R-j}:
//   X[k], X[R-k] = (a_0 + sum_j s_j cos(2 pi j k / R))  -+  i (sum_j d_j sin(2 pi j k / R)),   j, k = 1 .. (R-1)/2
// i.e. (R-1)^2 / 2 real-coefficient multiply-adds on complex values instead of (R-1)^2 complex products.
template <int R>
struct DftOddPrime
{
    static constexpr int H = (R - 1) / 2;
    static GSH_HD void run(cf (&a)[R])
    {
        cf s[H], d[H];
        static_for<H>([&](auto J) GSH_AI {
            constexpr int j = decltype(J)::value + 1;
            s[j - 1] = a[j] + a[R - j];
            d[j - 1] = a[j] - a[R - j];
        });
        const cf a0 = a[0];
        cf sum = a0;
        static_for<H>([&](auto J) GSH_AI { sum = sum + s[decltype(J)::value]; });
        a[0] = sum;
        static_for<H>([&](auto K) GSH_AI {
            constexpr int k = decltype(K)::value + 1;
            cf p = a0, q = cf{0.0f, 0.0f};
            static_for<H>([&](auto J) GSH_AI {
                constexpr int j = decltype(J)::value + 1;
                constexpr float c = static_cast<float>(cx_trig_turn(j * k, R, false));
                constexpr float sn = static_cast<float>(cx_trig_turn(j * k, R, true));
                p = p + s[j - 1] * c;
                q = q + d[j - 1] * sn;
            });
            a[k] = add_mj(p, q);      // p - i q
            a[R - k] = sub_mj(p, q);  // p + i q
        });
    }
};

template <>
struct Dft<7> : DftOddPrime<7>
{
};
template <>
struct Dft<11> : DftOddPrime<11>
{
};
template <>
struct Dft<13> : DftOddPrime<13>
{
};
template <>
struct Dft<31> : DftOddPrime<31>
{
};

// first factor of a composite radix: the largest hand-written butterfly that divides it
constexpr int first_factor(int r)
{
    for (int f : {8, 5, 4, 3, 2})
        if (r % f == 0 && r / f >= 1) return f;
    return r;
}

// R = Ra * Rb:  j = j1*Rb + j2,  k = k1 + Ra*k2,  W_R^{jk} = W_Ra^{j1 k1} * W_R^{j2 k1} * W_Rb^{j2 k2}
template <int Ra, int Rb>
struct DftComposite
{
    static constexpr int R = Ra * Rb;
    static GSH_HD void run(cf (&a)[R])
    {
        cf t[R];
        static_for<Rb>([&](auto J2) GSH_AI {
            constexpr int j2 = decltype(J2)::value;
            cf v[Ra];
            static_for<Ra>([&](auto J1) GSH_AI { v[decltype(J1)::value] = a[decltype(J1)::value * Rb + j2]; });
            Dft<Ra>::run(v);
            static_for<Ra>([&](auto K1) GSH_AI {
                constexpr int k1 = decltype(K1)::value;
                t[k1 * Rb + j2] = mul_w<j2 * k1, R>(v[k1]);
            });
        });
        static_for<Ra>([&](auto K1) GSH_AI {
            constexpr int k1 = decltype(K1)::value;
            cf v[Rb];
            static_for<Rb>([&](auto J2) GSH_AI { v[decltype(J2)::value] = t[k1 * Rb + decltype(J2)::value]; });
            Dft<Rb>::run(v);
            static_for<Rb>([&](auto K2) GSH_AI { a[k1 + Ra * decltype(K2)::value] = v[decltype(K2)::value]; });
        });
    }
};

template <int R>
struct Dft : DftComposite<first_factor(R), R / first_factor(R)>
{
    static_assert(first_factor(R) != R || R == 1, "radix has a prime factor without a butterfly");
};

// a[k] *= w^k, powers by a squaring tree (rounding-error depth log2 R, not R)
template <int R>
GSH_HD void mul_powers(cf (&a)[R], cf w)
{
    if constexpr (R >= 32)
        {
            // radix 32 and up: the full tree keeps R / 2 powers alive next to the R values (the sub-cell kernels, at the 128-register limit, spilled).
            // w^k = w^(k % 8) * (w^8)^(k / 8): eight low powers by the tree, the high power stepped block by block (depth 3 + R / 8).
            cf lo[8];
            lo[1] = w;
            lo[2] = csqr(w);
            lo[3] = cmul(lo[2], w);
            lo[4] = csqr(lo[2]);
            lo[5] = cmul(lo[4], w);
            lo[6] = csqr(lo[3]);
            lo[7] = cmul(lo[6], w);
            const cf w8 = csqr(lo[4]);
            cf hi = w8;
            static_for<R>([&](auto K) GSH_AI {
                constexpr int k = decltype(K)::value;
                if constexpr (k >= 8)
                    {
                        if constexpr (k % 8 == 0 && k > 8) hi = cmul(hi, w8);
                        a[k] = cmul(a[k], hi);
                    }
                if constexpr (k % 8 != 0) a[k] = cmul(a[k], lo[k % 8]);
            });
        }
    else
        {
            cf p[R];
            static_for<R>([&](auto K) GSH_AI {
                constexpr int k = decltype(K)::value;
                if constexpr (k == 1) p[1] = w;
                if constexpr (k >= 2)
                    {
                        if constexpr (k % 2 == 0)
                            p[k] = csqr(p[k / 2]);
                        else
                            p[k] = cmul(p[k - 1], w);
                    }
                if constexpr (k >= 1) a[k] = cmul(a[k], p[k]);
            });
        }
}

// exp(-2 pi i * num / den), 0 <= num < den, den < 2^23
GSH_HD cf unit_root(int num, int den)
{
    const float x = 2.0f * static_cast<float>(num) / static_cast<float>(den);
    float s, c;
#if defined(__HIP_DEVICE_COMPILE__)
    sincospif(x, &s, &c);
#else
    s = static_cast<float>(__builtin_sin(PI_D * static_cast<double>(x)));
    c = static_cast<float>(__builtin_cos(PI_D * static_cast<double>(x)));
#endif
    return cf{c, -s};
}

// ------------------------------------------------------------------------------------------ the three-stage plan
constexpr int round_up_congruent(int at_least, int minus, int mod)
{
    // smallest s >= at_least with (s - minus) % mod == 0
    int s = at_least;
    while (((s - minus) % mod + mod) % mod != 0) s++;
    return s;
}

template <int R1_, int R2_, int R3_>
struct Plan
{
    static constexpr int R1 = R1_, R2 = R2_, R3 = R3_;
    static constexpr int N = R1 * R2 * R3;
    static constexpr int T1 = R2 * R3;  // threads that own a stage-1 butterfly
    static constexpr int T2 = R1 * R3;
    static constexpr int T3 = R1 * R2;
    static constexpr int TMAX = T1 > T2 ? (T1 > T3 ? T1 : T3) : (T2 > T3 ? T2 : T3);
    static constexpr int THREADS = (TMAX + 63) / 64 * 64;
    // exchange 1 (stage 1 -> 2): float address k1*S1 + n2*R3 + n3.
    //   writer lanes t1 -> consecutive addresses; reader lanes t2 = k1*R3 + n3 -> t2 + k1*(S1 - R3) + const:
    //   conflict-free for ds_read_b32 (32 banks) when S1 - R3 is a multiple of 32.
    static constexpr int S1 = round_up_congruent(T1, R3, 32);
    // exchange 2 (stage 2 -> 3): float address k2*S2 + n3*P2 + k1, P2 odd so that the writer lanes (n3 fastest)
    //   spread over the banks; reader lanes t3 = k1 + R1*k2 -> t3 + k2*(S2 - R1) + const: conflict-free when
    //   S2 - R1 is a multiple of 32.
    static constexpr int P2 = R1 | 1;
    static constexpr int S2 = round_up_congruent(R3 * P2, R1, 32);
    // ---- exchanges one float component at a time (rounds 1-2): N * 4 bytes of LDS, four write / read passes, 8 barriers per transform.  Kept for the
    // plans whose N * 4 bytes leave room for several work-groups per compute unit (see EX64 below).
    static constexpr int LDS_FLOATS32 = (R1 * S1 > R2 * S2) ? R1 * S1 : R2 * S2;
    // ---- 64-bit phased exchanges (round 3).  A re-distribution moves whole complex values (ds_write_b64 / ds_read_b64: half the LDS instructions of
    // the component-at-a-time form, and a value lands in its register pair as it is), a few ROWS at a time: a reader needs one row only (k1 in
    // exchange 1, k2 in exchange 2), so the rows are cut into NP phases of RP rows, two LDS regions take the phases alternately, and a step
    // is "read phase p - 1, write phase p" between two barriers -- NP barriers per exchange, the reads of one phase overlap the writes of the
    // next, and exchange 2 starts in the region exchange 1 did not end in, so no barrier separates the two exchanges either.
    // N = 25 000: NP = 3 (9 + 9 + 7 rows), 2 x 73 KB of LDS, 6 barriers per transform (was 8 with 4 x 100 KB passes); N <= 9 000: NP = 1, 2 barriers.
    static constexpr int LDS_BUDGET_BYTES = 160000;  // of the CU's 163 840: the kernels keep ~200 bytes of reduction scratch next to it
    static constexpr int phases_for(int rows, int row_elems)
    {
        int np = 1;
        while (2 * ((rows + np - 1) / np) * row_elems * 8 > LDS_BUDGET_BYTES) np++;
        return np;
    }
    static constexpr int NP1 = phases_for(R1, S1), RP1 = (R1 + NP1 - 1) / NP1;
    static constexpr int NP2 = phases_for(R2, S2), RP2 = (R2 + NP2 - 1) / NP2;
    static constexpr int REGION = (RP1 * S1 > RP2 * S2) ? RP1 * S1 : RP2 * S2;  // complex elements per region
    static constexpr int LDS_CF = 2 * REGION;
    static constexpr int START1 = 0, START2 = NP1 % 2;  // region of phase 0; exchange 2 starts where exchange 1 did NOT end
    // Which form a plan uses (measured on MI355X, profiles/ab/r03/acq_exchange_ab.txt): the phased form wins where one work-group fills the compute
    // unit anyway (25 000: +2 %, 4 x 25 000: +6 %); below that the component form's smaller footprint keeps several work-groups resident and is up to
    // 25 % faster (8 000 points: 43 vs 53 us per batch).  The 16 000-point plan sits in between (+5 % on its own, -28 % as 8 x 16 000 = 128 000
    // points, where the sub-cells' load loop schedules worse around the larger LDS allocation) and stays with the component form.
    // The phased form also needs fewer registers (a value leaves its register pair when its row's phase is written: 16 384 points take 101 instead of
    // 128 and stop spilling), so a plan whose work-group has 1 024 threads -- alone on its compute unit anyway -- uses it too.
    // GSH_OC_EX32 / GSH_OC_EX64 force one form (A/B builds).
#if defined(GSH_OC_EX32)
    static constexpr bool EX64 = false;
#elif defined(GSH_OC_EX64)
    static constexpr bool EX64 = true;
#else
    static constexpr bool EX64 = LDS_FLOATS32 * 4 > 72 * 1024 || THREADS == 1024;  // (a 1 024-thread work-group is alone on its compute unit whatever its LDS)
#endif
    static constexpr int LDS_BYTES = EX64 ? LDS_CF * 8 : LDS_FLOATS32 * 4;
    static constexpr int LDS_FLOATS = LDS_BYTES / 4;
    // ---- stage 1: a[n1] = x[n1*T1 + t1]
    static GSH_HD void stage1(cf (&a)[R1], int t1)
    {
        Dft<R1>::run(a);
        mul_powers<R1>(a, unit_root(t1, N));
    }
    template <int COMP>
    static GSH_HD void ex1_write32(const cf (&a)[R1], int t1, float* lds)
    {
        static_for<R1>([&](auto K1) GSH_AI { lds[decltype(K1)::value * S1 + t1] = a[decltype(K1)::value][COMP]; });
    }
    template <int COMP>
    static GSH_HD void ex1_read32(cf (&b)[R2], int t2, const float* lds)
    {
        const int k1 = t2 / R3, n3 = t2 - k1 * R3;
        gsh_lds_rd_ptr p = GSH_LDS_RD_PTR(lds + k1 * S1 + n3);
        static_for<R2>([&](auto N2) GSH_AI { b[decltype(N2)::value][COMP] = p[decltype(N2)::value * R3]; });
    }
    // phase PH of exchange 1: rows k1 in [PH * RP1, (PH + 1) * RP1), complex address (k1 - PH * RP1) * S1 + n2 * R3 + n3 inside the phase's region
    template <int PH>
    static GSH_HD void ex1_write(const cf (&a)[R1], int t1, cf* lds)
    {
        cf* reg = lds + ((PH + START1) % 2) * REGION + t1;
        static_for<R1>([&](auto K1) GSH_AI {
            constexpr int k1 = decltype(K1)::value;
            if constexpr (k1 / RP1 == PH) reg[(k1 - PH * RP1) * S1] = a[k1];
        });
    }
    template <int PH>
    static GSH_HD void ex1_read(cf (&b)[R2], int t2, const cf* lds)
    {
        const int k1 = t2 / R3, n3 = t2 - k1 * R3;
        if (k1 / RP1 != PH) return;  // this thread's row travels in another phase
        const cf* p = lds + ((PH + START1) % 2) * REGION + (k1 - PH * RP1) * S1 + n3;
        static_for<R2>([&](auto N2) GSH_AI { b[decltype(N2)::value] = p[decltype(N2)::value * R3]; });
    }
    // ---- stage 2
    static GSH_HD void stage2(cf (&b)[R2], int t2)
    {
        const int n3 = t2 % R3;
        Dft<R2>::run(b);
        mul_powers<R2>(b, unit_root(n3, R2 * R3));
    }
    template <int COMP>
    static GSH_HD void ex2_write32(const cf (&b)[R2], int t2, float* lds)
    {
        const int k1 = t2 / R3, n3 = t2 - k1 * R3;
        float* p = lds + n3 * P2 + k1;
        static_for<R2>([&](auto K2) GSH_AI { p[decltype(K2)::value * S2] = b[decltype(K2)::value][COMP]; });
    }
    template <int COMP>
    static GSH_HD void ex2_read32(cf (&c)[R3], int t3, const float* lds)
    {
        const int k2 = t3 / R1, k1 = t3 - k2 * R1;
        gsh_lds_rd_ptr p = GSH_LDS_RD_PTR(lds + k2 * S2 + k1);
        static_for<R3>([&](auto N3) GSH_AI { c[decltype(N3)::value][COMP] = p[decltype(N3)::value * P2]; });
    }
    // phase PH of exchange 2: rows k2 in [PH * RP2, (PH + 1) * RP2), complex address (k2 - PH * RP2) * S2 + n3 * P2 + k1
    template <int PH>
    static GSH_HD void ex2_write(const cf (&b)[R2], int t2, cf* lds)
    {
        const int k1 = t2 / R3, n3 = t2 - k1 * R3;
        cf* reg = lds + ((PH + START2) % 2) * REGION + n3 * P2 + k1;
        static_for<R2>([&](auto K2) GSH_AI {
            constexpr int k2 = decltype(K2)::value;
            if constexpr (k2 / RP2 == PH) reg[(k2 - PH * RP2) * S2] = b[k2];
        });
    }
    template <int PH>
    static GSH_HD void ex2_read(cf (&c)[R3], int t3, const cf* lds)
    {
        const int k2 = t3 / R1, k1 = t3 - k2 * R1;
        if (k2 / RP2 != PH) return;
        const cf* p = lds + ((PH + START2) % 2) * REGION + (k2 - PH * RP2) * S2 + k1;
        static_for<R3>([&](auto N3) GSH_AI { c[decltype(N3)::value] = p[decltype(N3)::value * P2]; });
    }
    // ---- stage 3: c[k3] -> X[t3 + T3*k3]
    static GSH_HD void stage3(cf (&c)[R3]) { Dft<R3>::run(c); }
};
}  // namespace oc
}  // namespace gsh

// transform lengths with an on-chip plan: X(R1, R2, R3), N = R1*R2*R3 (N*4 bytes of LDS, <= 1024 threads, <= 40
// elements per thread).  1 ms (GPS L1 / L5) and 4 ms (Galileo E1) code periods at the usual front-end rates.
#ifndef GSH_OC_PLANS  /* (a tuning build may name a shorter list on the command line: one plan compiles in seconds) */
#define GSH_OC_PLANS(X) \
    X(25, 25, 40) /* 25 000: 25 Msps x 1 ms, 6.25 Msps x 4 ms */ \
    X(10, 20, 20) /*  4 000:  4 Msps x 1 ms */ \
    X(8, 16, 16)  /*  2 048:  2.048 Msps x 1 ms */ \
    X(16, 16, 16) /*  4 096 */ \
    X(10, 20, 25) /*  5 000:  5 Msps x 1 ms */ \
    X(20, 20, 20) /*  8 000:  8 Msps x 1 ms, 2 Msps x 4 ms */ \
    X(16, 16, 32) /*  8 192 */ \
    X(20, 20, 25) /* 10 000: 10 Msps x 1 ms, 2.5 Msps x 4 ms */ \
    X(20, 25, 25) /* 12 500: 12.5 Msps x 1 ms */ \
    X(20, 20, 40) /* 16 000: 16 Msps x 1 ms, 4 Msps x 4 ms */ \
    X(16, 32, 32) /* 16 384 */ \
    X(25, 25, 32) /* 20 000: 20 Msps x 1 ms, 5 Msps x 4 ms */ \
    X(32, 32, 32) /* 32 768 */ \
    X(10, 10, 10) /*  1 000: QuickSync folds of 4 000 (p = 4), 1 Msps x 1 ms */ \
    X(10, 10, 20) /*  2 000: 2 Msps x 1 ms (the flowgraph's acquisition resampler for L1 / E1), QuickSync folds of 8 000 */ \
    X(10, 10, 25) /*  2 500: 2.5 Msps x 1 ms (25 Msps decimated by 10) */ \
    X(10, 25, 25) /*  6 250: 6.25 Msps x 1 ms */ \
    X(6, 11, 31)  /*  2 046: 2.046 Msps x 1 ms */ \
    X(12, 11, 31) /*  4 092: 4.092 Msps x 1 ms */ \
    X(24, 11, 31) /*  8 184: 8.184 Msps x 1 ms (GN3S-class front ends), 2.046 Msps x 4 ms (Galileo E1) */ \
    X(22, 24, 31) /* 16 368: 16.368 Msps x 1 ms, 4.092 Msps x 4 ms */ \
    X(16, 11, 31) /*  5 456: 5.456 Msps x 1 ms */ \
    X(10, 16, 16) /*  2 560: 2.56 Msps x 1 ms */ \
    X(16, 16, 40) /* 10 240: 2.56 Msps x 4 ms */ \
    X(25, 32, 32) /* 25 600: 25.6 Msps x 1 ms, 6.4 Msps x 4 ms; the sub-transform of 128 000 = 5 x 25 600 */
#endif

// N = S * M: one radix-S decimation-in-frequency step in front of the plan of M (pcps_onchip.hip); X(S, R1, R2, R3)
// (no 2 x (32, 32, 32): with 32 running sums per thread next to a radix-2 front step the sub-cell kernels do not fit 128 registers without
// scratch; 65 536 points -- no front-end rate of the reference's configurations gives it -- take the four-step path)
#ifndef GSH_OC_SPLIT_PLANS
#define GSH_OC_SPLIT_PLANS(X) \
    X(2, 25, 25, 40) /*  50 000: 50 Msps x 1 ms; bit-transition search at 25 Msps; 12.5 Msps x 4 ms */ \
    X(4, 25, 25, 40) /* 100 000: 25 Msps x 4 ms (Galileo E1) */ \
    X(8, 25, 25, 40) /* 200 000: 50 Msps x 4 ms */ \
    X(2, 20, 20, 40) /*  32 000: 4 Msps x 8 ms (Galileo E1 8 ms block), 8 Msps x 4 ms, 32 Msps x 1 ms */ \
    X(4, 20, 20, 40) /*  64 000: 16 Msps x 4 ms */ \
    X(5, 25, 32, 32) /* 128 000: 32 Msps x 4 ms (Galileo E1); five sub-cells read the product spectrum 5 times, 8 x 16 000 (rounds 1-2) read it 8 times */ \
    X(2, 25, 25, 32) /*  40 000: 40 Msps x 1 ms, 10 Msps x 4 ms; bit-transition search at 20 Msps */ \
    X(4, 25, 25, 32) /*  80 000: 4 Msps x 20 ms (GPS L2C), 20 Msps x 4 ms */ \
    X(2, 22, 24, 31) /*  32 736: 8.184 Msps x 4 ms (Galileo E1) */
#endif

#endif
