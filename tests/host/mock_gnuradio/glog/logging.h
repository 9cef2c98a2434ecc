#ifndef MOCK_GLOG_LOGGING_H
#define MOCK_GLOG_LOGGING_H
#include <iostream>
struct MockLogSink
{
    template <typename T>
    MockLogSink& operator<<(const T&) { return *this; }
    MockLogSink& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
#define LOG(level) MockLogSink()
#define DLOG(level) MockLogSink()
#define VLOG(level) MockLogSink()
#endif
