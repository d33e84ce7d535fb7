// hip_util.h -- small RAII helpers shared by the mesh filters that run on the GPU (simplify_gpu.hip, clean_gpu.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace sf {

struct DevBuf {   // one hipMalloc'ed buffer, freed with its scope
  void* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) {
    if (p) { (void)hipFree(p); p = nullptr; }
    return hipMalloc(&p, bytes ? bytes : 16);
  }
  template <typename T> T* as() { return (T*)p; }
};

struct StreamGuard {   // a stream of the call's own: host threads finishing several meshes, and the fuser of the next scan, share the device
  hipStream_t s = nullptr;
  StreamGuard() = default;
  StreamGuard(const StreamGuard&) = delete;
  StreamGuard& operator=(const StreamGuard&) = delete;
  ~StreamGuard() { if (s) (void)hipStreamDestroy(s); }
};

}  // namespace sf
