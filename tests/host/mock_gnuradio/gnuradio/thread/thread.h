#ifndef MOCK_GR_THREAD_H
#define MOCK_GR_THREAD_H
// gr::thread names gnss-sdr's blocks use (GNU Radio typedefs them to boost::thread / boost::mutex); std:: stands in
#include <mutex>
#include <thread>
namespace gr
{
namespace thread
{
typedef std::thread thread;
typedef std::recursive_mutex mutex;
class scoped_lock
{
public:
    explicit scoped_lock(mutex& m) : d_lock(m) {}
    void lock() { d_lock.lock(); }
    void unlock() { d_lock.unlock(); }

private:
    std::unique_lock<mutex> d_lock;
};
}  // namespace thread
}  // namespace gr
#endif
