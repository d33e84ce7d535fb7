// common.h -- error plumbing shared by the host-side translation units of libscanfuse.so
#pragma once
#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/scanfuse.h"
#include "../../include/scanfuse_internal.h"

namespace sf {

std::string& last_error_ref();

// CPUs this process may really use: hardware concurrency capped by the cgroup CPU quota (params.cpp)
int usable_cpus();

inline int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}

}  // namespace sf

#define SF_API extern "C" __attribute__((visibility("default")))

#define SF_HIP_CHECK(call)                                                                       \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return sf::fail(SF_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
