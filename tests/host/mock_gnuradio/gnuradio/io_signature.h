#ifndef MOCK_GR_IO_SIGNATURE_H
#define MOCK_GR_IO_SIGNATURE_H
#include <cstddef>
#include <memory>
namespace gr
{
class io_signature
{
public:
    using sptr = std::shared_ptr<io_signature>;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item) { return sptr(new io_signature(min_streams, max_streams, sizeof_stream_item)); }
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    int sizeof_stream_item(int) const { return d_size; }

private:
    io_signature(int mn, int mx, int sz) : d_min(mn), d_max(mx), d_size(sz) {}
    int d_min, d_max, d_size;
};
}  // namespace gr
#endif
