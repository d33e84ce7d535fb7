// TEST INFRASTRUCTURE (tools/tsan): a stand-in for <hip/hip_runtime.h> that lets pipeline.hip (sf_fuse_run: decode pool, pinned ring, copy
// streams, reaper thread) be compiled as plain C++ and run under ThreadSanitizer without a GPU.  Streams are REAL asynchronous queues
// (one worker thread each, operations executed in order), events complete when their stream reaches them, "device memory" is host
// memory -- so a protocol error (a decode worker refilling a pinned slot before the copy that reads it has run, a copy overwriting device
// buffers a queued pass still reads) shows up as a data race between threads, which is what TSan reports.
#pragma once
#include <cstddef>
#include <cstdint>

#define __host__
#define __device__
#define __global__
#define __restrict__ __restrict

struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
struct FakeStream;
struct FakeEvent;
typedef FakeStream* hipStream_t;
typedef FakeEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0;

const char* hipGetErrorString(hipError_t);
hipError_t hipSetDevice(int);
hipError_t hipGetDeviceCount(int*);
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned);
hipError_t hipStreamCreateWithPriority(hipStream_t*, unsigned, int);
hipError_t hipDeviceGetStreamPriorityRange(int*, int*);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipEventCreateWithFlags(hipEvent_t*, unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t);
hipError_t hipHostMalloc(void**, size_t, unsigned);
hipError_t hipHostFree(void*);
hipError_t hipMalloc(void**, size_t);
hipError_t hipFree(void*);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind);
hipError_t hipMemset(void* dst, int value, size_t bytes);

// for the stubs of the device side (tools/tsan/harness.cpp): run `fn(arg)` on the stream, in order
void fake_stream_enqueue(hipStream_t, void (*fn)(void*), void* arg);
