#ifndef MOCK_PMT_H
#define MOCK_PMT_H
#include <any>
#include <memory>
#include <cstdint>
#include <string>
#include <typeinfo>
#include <utility>
#include <vector>
namespace pmt
{
struct pmt_base
{
    std::string symbol;
    long value{0};
    bool is_long{false};
    std::any any;
};
using pmt_t = std::shared_ptr<pmt_base>;
inline pmt_t mp(const std::string& s)
{
    auto p = std::make_shared<pmt_base>();
    p->symbol = s;
    return p;
}
inline pmt_t intern(const std::string& s) { return mp(s); }
inline pmt_t from_long(long v)
{
    auto p = std::make_shared<pmt_base>();
    p->value = v;
    p->is_long = true;
    return p;
}
inline long to_long(const pmt_t& p) { return p->value; }
inline std::string symbol_to_string(const pmt_t& p) { return p->symbol; }
inline pmt_t make_any(const std::any& a)
{
    auto p = std::make_shared<pmt_base>();
    p->any = a;
    return p;
}
inline std::any& any_ref(const pmt_t& p) { return p->any; }
inline bool eqv(const pmt_t& a, const pmt_t& b) { return a && b && a->symbol == b->symbol; }
// numbers and dictionaries as far as the reference's sensor-data tag helpers use them (src/algorithms/libs/sensor_data/: compiled in place beside the direct
// resampler blocks, oracle/Makefile); a dictionary is a vector of (key, value) pairs held in `any`
inline pmt_t from_uint64(uint64_t v) { return make_any(std::any(v)); }
inline uint64_t to_uint64(const pmt_t& p) { return std::any_cast<uint64_t>(p->any); }
inline pmt_t from_double(double v) { return make_any(std::any(v)); }
inline pmt_t from_float(double v) { return make_any(std::any(v)); }
inline double to_double(const pmt_t& p) { return p->is_long ? static_cast<double>(p->value) : std::any_cast<double>(p->any); }
inline double to_float(const pmt_t& p) { return to_double(p); }
using dict_t = std::vector<std::pair<pmt_t, pmt_t>>;
inline pmt_t make_dict() { return make_any(std::any(dict_t())); }
inline bool is_dict(const pmt_t& p) { return p && p->any.type() == typeid(dict_t); }
inline bool dict_has_key(const pmt_t& d, const pmt_t& key)
{
    if (!is_dict(d)) return false;
    for (const auto& kv : std::any_cast<const dict_t&>(d->any))
        if (eqv(kv.first, key)) return true;
    return false;
}
inline pmt_t dict_ref(const pmt_t& d, const pmt_t& key, const pmt_t& not_found)
{
    if (is_dict(d))
        for (const auto& kv : std::any_cast<const dict_t&>(d->any))
            if (eqv(kv.first, key)) return kv.second;
    return not_found;
}
inline pmt_t dict_add(const pmt_t& d, const pmt_t& key, const pmt_t& value)
{
    dict_t out = is_dict(d) ? std::any_cast<const dict_t&>(d->any) : dict_t();
    for (auto& kv : out)
        if (eqv(kv.first, key))
            {
                kv.second = value;
                return make_any(std::any(out));
            }
    out.emplace_back(key, value);
    return make_any(std::any(out));
}
}  // namespace pmt
#endif
