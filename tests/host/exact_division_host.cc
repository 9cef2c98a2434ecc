// Host build of gnss-sdr_amd/csrc/exact_division.h for tests/test_exact_division.py: the functions applied to arrays, next to the machine's own division / fmod.
#include "exact_division.h"
#include <cmath>
#include <cstdint>
#include <cstring>

extern "C"
{
    // returns the number of operands whose quotient differs in any bit from a[i] / b
    int64_t gsh_test_div_by_constant(const double* a, int64_t n, double b, double* first_bad)
    {
        const double y = 1.0 / b;
        int64_t bad = 0;
        for (int64_t i = 0; i < n; i++)
            {
                const double q = gsh::div_by_constant(a[i], b, y), ref = a[i] / b;
                if (std::memcmp(&q, &ref, sizeof(q)) != 0 && !(std::isnan(q) && std::isnan(ref)))
                    {
                        if (bad == 0 && first_bad != nullptr) *first_bad = a[i];
                        bad++;
                    }
            }
        return bad;
    }

    int64_t gsh_test_fmod_by_constant(const double* x, int64_t n, double p, double* first_bad, int64_t* n_slow)
    {
        const double inv = 1.0 / p;
        int64_t bad = 0, slow_count = 0;
        for (int64_t i = 0; i < n; i++)
            {
                bool slow = false;
                const double r = gsh::fmod_by_constant(x[i], p, inv, &slow), ref = std::fmod(x[i], p);
                if (slow)
                    {
                        slow_count++;
                        continue;
                    }
                if (std::memcmp(&r, &ref, sizeof(r)) != 0)
                    {
                        if (bad == 0 && first_bad != nullptr) *first_bad = x[i];
                        bad++;
                    }
            }
        if (n_slow != nullptr) *n_slow = slow_count;
        return bad;
    }
}
