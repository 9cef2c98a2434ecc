#ifndef MOCK_GLOG_LOGGING_H
#define MOCK_GLOG_LOGGING_H
// glog's LOG / DLOG / VLOG as far as gnss-sdr's sources use them.  Silent, unless GSH_TEST_LOG is set in the environment: then WARNING and ERROR lines go to stderr
// (GSH_TEST_LOG=2: INFO as well) -- how a test run shows what an adapter gave up on.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
struct MockLogSink
{
    explicit MockLogSink(const char* level)
    {
        static const int verbosity = [] {
            const char* e = std::getenv("GSH_TEST_LOG");
            return e != nullptr ? std::atoi(e) : 0;
        }();
        on = verbosity >= 2 || (verbosity >= 1 && (std::strcmp(level, "WARNING") == 0 || std::strcmp(level, "ERROR") == 0 || std::strcmp(level, "FATAL") == 0));
        if (on) ss << "[" << level << "] ";
    }
    ~MockLogSink()
    {
        if (on)
            {
                ss << '\n';
                std::cerr << ss.str();
            }
    }
    template <typename T>
    MockLogSink& operator<<(const T& v)
    {
        if (on) ss << v;
        return *this;
    }
    MockLogSink& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
    bool on{false};
    std::ostringstream ss;
};
#define LOG(level) MockLogSink(#level)
#define DLOG(level) MockLogSink("DEBUG")
#define VLOG(level) MockLogSink("DEBUG")
#endif
