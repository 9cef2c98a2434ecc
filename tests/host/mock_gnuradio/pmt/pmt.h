#ifndef MOCK_PMT_H
#define MOCK_PMT_H
#include <memory>
#include <string>
namespace pmt
{
struct pmt_base
{
    std::string symbol;
    long value{0};
    bool is_long{false};
};
using pmt_t = std::shared_ptr<pmt_base>;
inline pmt_t mp(const std::string& s)
{
    auto p = std::make_shared<pmt_base>();
    p->symbol = s;
    return p;
}
inline pmt_t from_long(long v)
{
    auto p = std::make_shared<pmt_base>();
    p->value = v;
    p->is_long = true;
    return p;
}
inline long to_long(const pmt_t& p) { return p->value; }
inline std::string symbol_to_string(const pmt_t& p) { return p->symbol; }
}  // namespace pmt
#endif
