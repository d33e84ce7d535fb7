// launch_cost.hip -- where does the host time of a frame go?  Enqueue cost of the HIP calls run_batch makes, measured on an idle stream:
// kernels with small / 2.6 KB by-value arguments, event records (timing / no timing), cross-stream waits.  hipcc -O2 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float f[650]; };   // 2.6 KB like BatchFrames
__global__ void k_small(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 9999) *p = (int)b.f[3]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t et, en;
  hipEventCreate(&et);
  hipEventCreateWithFlags(&en, hipEventDisableTiming);
  Big b = {};
  const int N = 2000;
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, nullptr);
    double t1 = now(); hipStreamSynchronize(s); double t1b = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, b, nullptr);
    double t2 = now(); hipStreamSynchronize(s); double t2b = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, nullptr); hipEventRecord(et, s); }
    double t3 = now(); hipStreamSynchronize(s); double t3b = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, nullptr); hipEventRecord(en, s); }
    double t4 = now(); hipStreamSynchronize(s); double t4b = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, nullptr); hipEventRecord(en, s); hipStreamWaitEvent(s2, en, 0);
                                  hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s2, nullptr); }
    double t5 = now(); hipStreamSynchronize(s); hipStreamSynchronize(s2); double t5b = now();
    if (rep == 1)
      printf("{\"per_call_us\": {\"launch_small\": %.2f, \"launch_2600B_arg\": %.2f, \"launch_plus_timing_event\": %.2f, \"launch_plus_notiming_event\": %.2f, "
             "\"launch_event_wait_launch_2streams\": %.2f}, \"drain_us\": {\"small\": %.2f, \"big\": %.2f, \"timing_event\": %.2f, \"notiming_event\": %.2f, \"two_streams\": %.2f}}\n",
             (t1 - t0) / N * 1e6, (t2 - t1b) / N * 1e6, (t3 - t2b) / N * 1e6, (t4 - t3b) / N * 1e6, (t5 - t4b) / N * 1e6,
             (t1b - t0) / N * 1e6, (t2b - t1b) / N * 1e6, (t3b - t2b) / N * 1e6, (t4b - t3b) / N * 1e6, (t5b - t4b) / N * 1e6);
  }
  return 0;
}
