// TEST INFRASTRUCTURE: the asynchronous fake HIP runtime behind tools/tsan/fake_hip/hip/hip_runtime.h
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

struct FakeEvent {
  std::mutex m;
  std::condition_variable cv;
  uint64_t recorded = 0, done = 0;   // generation counters: record() bumps `recorded`, the stream bumps `done` when it gets there
};

struct FakeStream {
  std::mutex m;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool busy = false, stop = false;
  std::thread th;
  FakeStream() : th([this] { run(); }) {}
  void run() {
    for (;;) {
      std::function<void()> op;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        op = std::move(q.front());
        q.pop_front();
        busy = true;
      }
      op();
      {
        std::lock_guard<std::mutex> lk(m);
        busy = false;
        if (q.empty()) idle.notify_all();
      }
    }
  }
  void push(std::function<void()> f) {
    { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(f)); }
    cv.notify_one();
  }
  void drain() {
    std::unique_lock<std::mutex> lk(m);
    idle.wait(lk, [&] { return q.empty() && !busy; });
  }
};

const char* hipGetErrorString(hipError_t) { return "fake hip error"; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new FakeStream(); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new FakeStream(); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) {
  if (!s) return hipSuccess;
  s->drain();
  { std::lock_guard<std::mutex> lk(s->m); s->stop = true; }
  s->cv.notify_one();
  s->th.join();
  delete s;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { s->drain(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new FakeEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  uint64_t gen;
  { std::lock_guard<std::mutex> lk(e->m); gen = ++e->recorded; }
  s->push([e, gen] { { std::lock_guard<std::mutex> lk(e->m); if (e->done < gen) e->done = gen; } e->cv.notify_all(); });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  uint64_t gen;
  { std::lock_guard<std::mutex> lk(e->m); gen = e->recorded; }   // the most recent record at the time of the call, as HIP defines it
  s->push([e, gen] { std::unique_lock<std::mutex> lk(e->m); e->cv.wait(lk, [&] { return e->done >= gen; }); });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  std::unique_lock<std::mutex> lk(e->m);
  const uint64_t gen = e->recorded;
  e->cv.wait(lk, [&] { return e->done >= gen; });
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
  s->push([dst, src, bytes] { std::memcpy(dst, src, bytes); });
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n); return *p ? hipSuccess : 1; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : 1; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { std::memcpy(dst, src, bytes); return hipSuccess; }
hipError_t hipMemset(void* dst, int value, size_t bytes) { std::memset(dst, value, bytes); return hipSuccess; }
void fake_stream_enqueue(hipStream_t s, void (*fn)(void*), void* arg) { s->push([fn, arg] { fn(arg); }); }
