/*
 * TEST INFRASTRUCTURE ONLY: Boost is absent.  boost::math::gamma_p_inv(a, p) -- the x with P(a, x) = p, P the
 * regularised lower incomplete gamma function -- is evaluated here from its definition (series for x < a + 1,
 * Lentz continued fraction otherwise; safeguarded Newton on x).  pcps_acquisition.cc:52-56 rounds the result to
 * float, so double-precision agreement to ~1e-13 relative gives the same threshold as Boost.
 */
#ifndef ORACLE_SHIM_BOOST_GAMMA_HPP
#define ORACLE_SHIM_BOOST_GAMMA_HPP
#include <cmath>
#include <limits>
namespace boost
{
namespace math
{
namespace shim_detail
{
inline double gamma_p(double a, double x)
{
    if (x <= 0.0) return 0.0;
    const double lg = std::lgamma(a);
    if (x < a + 1.0)
        {
            double term = 1.0 / a, sum = term, ap = a;
            for (int i = 0; i < 100000; i++)
                {
                    ap += 1.0;
                    term *= x / ap;
                    sum += term;
                    if (std::fabs(term) < std::fabs(sum) * 1e-17) break;
                }
            return sum * std::exp(-x + a * std::log(x) - lg);
        }
    const double tiny = 1e-300;
    double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
    for (int i = 1; i < 100000; i++)
        {
            const double an = -i * (i - a);
            b += 2.0;
            d = an * d + b;
            if (std::fabs(d) < tiny) d = tiny;
            c = b + an / c;
            if (std::fabs(c) < tiny) c = tiny;
            d = 1.0 / d;
            const double del = d * c;
            h *= del;
            if (std::fabs(del - 1.0) < 1e-17) break;
        }
    return 1.0 - std::exp(-x + a * std::log(x) - lg) * h;
}
}  // namespace shim_detail

inline double gamma_p_inv(double a, double p)
{
    if (p <= 0.0) return 0.0;
    if (p >= 1.0) return std::numeric_limits<double>::infinity();
    double lo = 0.0, hi = a + 1.0;
    while (shim_detail::gamma_p(a, hi) < p) hi *= 2.0;
    double x = 0.5 * (lo + hi);
    const double lg = std::lgamma(a);
    for (int it = 0; it < 400; it++)
        {
            const double f = shim_detail::gamma_p(a, x) - p;
            if (f > 0.0) hi = x; else lo = x;
            const double dens = std::exp(-x + (a - 1.0) * std::log(x) - lg);
            double xn = (dens > 0.0) ? x - f / dens : 0.5 * (lo + hi);
            if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
            if (std::fabs(xn - x) <= 4e-16 * std::fabs(x)) { x = xn; break; }
            x = xn;
        }
    return x;
}
}  // namespace math
}  // namespace boost
#endif
