/*
 * TEST INFRASTRUCTURE ONLY: Boost is absent.  The input-filter blocks (pulse_blanking_cc.cc:51-52, notch_cc.cc:54-55, notch_lite_cc.cc:60-61) take their
 * threshold from  quantile(complement(chi_squared_distribution<float>(k), pfa))  -- the x with Q(k/2, x/2) = pfa, i.e. x = 2 * gamma_p_inv(k/2, 1 - pfa).
 */
#ifndef ORACLE_SHIM_BOOST_CHI_SQUARED_HPP
#define ORACLE_SHIM_BOOST_CHI_SQUARED_HPP
#include <boost/math/special_functions/gamma.hpp>
namespace boost
{
namespace math
{
template <class T>
struct chi_squared_distribution
{
    explicit chi_squared_distribution(T k) : k_(k) {}
    T degrees_of_freedom() const { return k_; }
    T k_;
};
template <class D, class P>
struct complemented2_type
{
    const D& dist;
    P param;
};
template <class T, class P>
complemented2_type<chi_squared_distribution<T>, P> complement(const chi_squared_distribution<T>& d, P p)
{
    return complemented2_type<chi_squared_distribution<T>, P>{d, p};
}
template <class T, class P>
T quantile(const complemented2_type<chi_squared_distribution<T>, P>& c)
{
    return static_cast<T>(2.0 * gamma_p_inv(0.5 * static_cast<double>(c.dist.k_), 1.0 - static_cast<double>(c.param)));
}
}  // namespace math
}  // namespace boost
#endif
