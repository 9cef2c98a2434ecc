#ifndef MOCK_PMT_H
#define MOCK_PMT_H
#include <any>
#include <memory>
#include <string>
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
}  // namespace pmt
#endif
