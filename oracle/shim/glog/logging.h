/* TEST INFRASTRUCTURE: stand-in for <glog/logging.h>: the reference's tracking_loop_filter.cc only uses
 * LOG(WARNING) << ... in set_order (tracking_loop_filter.cc:247-253); the stream is discarded. */
#ifndef SHIM_GLOG_LOGGING_H
#define SHIM_GLOG_LOGGING_H
#include <iostream>
struct ShimNullLog
{
    template <class T>
    ShimNullLog& operator<<(const T&) { return *this; }
};
#define LOG(severity) ShimNullLog()
#define DLOG(severity) ShimNullLog()
#endif
