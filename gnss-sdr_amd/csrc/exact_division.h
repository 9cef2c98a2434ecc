// Division by a launch constant and fmod by a constant, bit for bit the IEEE results, for the one-lane loop arithmetic of the closed tracking loop
// (tracking_loop.hip).  Plain C++ over compiler builtins: the same text runs on the device and -- for tests/test_exact_division.py, which compares it with the
// machine's own division and fmod over millions of operands -- on the host.
#ifndef GSH_EXACT_DIVISION_H
#define GSH_EXACT_DIVISION_H

#ifdef __HIPCC__
#define GSH_EXACT_FN __device__ __host__ inline __attribute__((always_inline))
#else
#define GSH_EXACT_FN inline
#endif

#ifndef GSH_TRK_FAST_DIV
#define GSH_TRK_FAST_DIV 1
#endif

namespace gsh
{
// a / b for a divisor that is constant over the launch, y = RN(1 / b) formed once (by the host, or by the compiler for a literal): the IEEE quotient in five
// dependent operations instead of the division's reciprocal estimate + refinement + scaling sequence (about twice as many; each of the loop's five such
// divisions sits on the one-lane critical path of a period: 8.11 -> 7.96 us per period, profiles/ab/r03/closed_loop_steps.txt).
//   q0 = RN(a y) is within 2 ulp of a / b; one residual step makes it faithful; and for a faithful q with the exact residual r = a - b q (one FMA),
//   RN(q + r y) is the correctly rounded quotient whenever y is the correctly rounded reciprocal and b's significand is not all ones (Markstein, "Computation
//   of elementary functions on the IBM RISC System/6000 processor", IBM J. Res. Dev. 34, 1990; Cornea, Harrison, Tang, "Scientific Computing on Itanium-based
//   Systems", 2002, ch. 8).
// Preconditions, vetted ON THE HOST for the whole configuration (fast_division_applies in tracking_loop.hip: every constant that enters a dividend or is a
// divisor lies in a sane range, so that no residual and no quotient can leave the normal numbers): b > 0 finite; the dividend is zero, not finite, or of
// magnitude in (1e-150, 1e150).  A zero, infinite or NaN dividend gives itself (what a / b gives for positive finite b, sign of zero included) through a
// select, not a branch: a data-dependent branch here costs more than the division saves (it splits the one lane's instruction stream: 8.29 us).
// y == 0.0 -- the host's "not vetted" -- selects the plain division with a wave-uniform branch; the loop kernel takes that decision once for a whole stretch
// of its arithmetic and calls div_by_constant_vetted inside.
GSH_EXACT_FN double div_by_constant_vetted(double a, double b, double y)  // y = RN(1 / b), known to apply
{
    const double m = __builtin_fabs(a);
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    q = __builtin_fma(r, y, q);
    r = __builtin_fma(-b, q, a);
    q = __builtin_fma(r, y, q);
    return (m > 0.0 && m < __builtin_inf()) ? q : a;
}
GSH_EXACT_FN double div_by_constant(double a, double b, double y)
{
#if GSH_TRK_FAST_DIV
    if (y != 0.0) return div_by_constant_vetted(a, b, y);
#endif
    return a / b;
}

// fmod(x, p) for a positive constant p with inv_p = RN(1 / p), |x| < 1e6 p: n = trunc(|x| / p) formed with the reciprocal is at most one off (the quotient's
// error is below 2^-32); |x| - n p is exact in one FMA for the right n and for n +- 1 (the remainder of a floating-point division is representable, and so is
// that remainder -+ the divisor: a multiple of the smaller operand's ulp, below the divisor in magnitude); one exact correction gives fmod's value, the sign
// is x's.  *slow is set, and nothing computed, for larger and for non-finite arguments: the caller takes the library's fmod.
GSH_EXACT_FN double fmod_by_constant(double x, double p, double inv_p, bool* slow)
{
    const double m = __builtin_fabs(x);
    *slow = !(m < 1.0e6 * p);
    const double n = __builtin_trunc(m * inv_p);
    const double r = __builtin_fma(-n, p, m);
    const double up = r + p, down = r - p;  // (both formed next to the comparisons: a shorter chain than correct-then-test-again)
    return __builtin_copysign((r < 0.0) ? up : ((r >= p) ? down : r), x);
}
}  // namespace gsh

#endif  // GSH_EXACT_DIVISION_H
