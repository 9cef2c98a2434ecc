/* TEST INFRASTRUCTURE ONLY: Boost is absent.  boost::circular_buffer as the tracking block uses it (fixed capacity, push_back
 * overwrites the oldest, random access from the oldest element, iteration, front/back, clear, set_capacity). */
#ifndef ORACLE_SHIM_BOOST_CIRCULAR_BUFFER_HPP
#define ORACLE_SHIM_BOOST_CIRCULAR_BUFFER_HPP
#include <cstddef>
#include <iterator>
#include <vector>
namespace boost
{
template <typename T>
class circular_buffer
{
public:
    typedef std::size_t size_type;
    typedef T value_type;
    circular_buffer() = default;
    explicit circular_buffer(size_type cap) : d(cap) {}
    void set_capacity(size_type cap)
    {
        std::vector<T> keep;
        for (size_type i = 0; i < n; i++) keep.push_back((*this)[i]);
        d.assign(cap, T());
        head = 0;
        n = 0;
        for (auto& v : keep) push_back(v);
    }
    size_type capacity() const { return d.size(); }
    size_type size() const { return n; }
    bool empty() const { return n == 0; }
    bool full() const { return n == d.size(); }
    void clear()
    {
        head = 0;
        n = 0;
    }
    void push_back(const T& v)
    {
        if (d.empty()) return;
        if (n < d.size())
            {
                d[(head + n) % d.size()] = v;
                n++;
            }
        else
            {
                d[head] = v;
                head = (head + 1) % d.size();
            }
    }
    void pop_front()
    {
        if (n == 0) return;
        head = (head + 1) % d.size();
        n--;
    }
    T& operator[](size_type i) { return d[(head + i) % d.size()]; }
    const T& operator[](size_type i) const { return d[(head + i) % d.size()]; }
    T& at(size_type i) { return (*this)[i]; }
    const T& at(size_type i) const { return (*this)[i]; }
    T& front() { return (*this)[0]; }
    T& back() { return (*this)[n - 1]; }
    const T& front() const { return (*this)[0]; }
    const T& back() const { return (*this)[n - 1]; }

    template <typename CB, typename R>
    class iter
    {
    public:
        typedef std::random_access_iterator_tag iterator_category;
        typedef T value_type;
        typedef std::ptrdiff_t difference_type;
        typedef R* pointer;
        typedef R& reference;
        iter() = default;
        iter(CB* c, size_type i) : cb(c), idx(i) {}
        reference operator*() const { return (*cb)[idx]; }
        pointer operator->() const { return &(*cb)[idx]; }
        reference operator[](difference_type k) const { return (*cb)[idx + k]; }
        iter& operator++() { ++idx; return *this; }
        iter operator++(int) { iter t = *this; ++idx; return t; }
        iter& operator--() { --idx; return *this; }
        iter operator--(int) { iter t = *this; --idx; return t; }
        iter& operator+=(difference_type k) { idx += k; return *this; }
        iter& operator-=(difference_type k) { idx -= k; return *this; }
        iter operator+(difference_type k) const { return iter(cb, idx + k); }
        iter operator-(difference_type k) const { return iter(cb, idx - k); }
        difference_type operator-(const iter& o) const { return static_cast<difference_type>(idx) - static_cast<difference_type>(o.idx); }
        bool operator==(const iter& o) const { return idx == o.idx; }
        bool operator!=(const iter& o) const { return idx != o.idx; }
        bool operator<(const iter& o) const { return idx < o.idx; }
        bool operator>(const iter& o) const { return idx > o.idx; }
        bool operator<=(const iter& o) const { return idx <= o.idx; }
        bool operator>=(const iter& o) const { return idx >= o.idx; }

    private:
        CB* cb{nullptr};
        size_type idx{0};
    };
    typedef iter<circular_buffer, T> iterator;
    typedef iter<const circular_buffer, const T> const_iterator;
    iterator begin() { return iterator(this, 0); }
    iterator end() { return iterator(this, n); }
    const_iterator begin() const { return const_iterator(this, 0); }
    const_iterator end() const { return const_iterator(this, n); }
    const_iterator cbegin() const { return begin(); }
    const_iterator cend() const { return end(); }

private:
    std::vector<T> d;
    size_type head{0}, n{0};
};
}  // namespace boost
#endif
