// struct_buffer_probe.hip -- what does gfx950 range-check on a buffer load with idxen + offen through a V# with stride != 0 and swizzle off?
// If both fields are checked (index >= num_records -> 0, offset >= stride -> 0) a 2-D image gather needs neither the row multiply nor the
// "inside the image" compares.
//   hipcc -O2 --offload-arch=gfx950 tools/gpu/struct_buffer_probe.hip -o /tmp/sbp && /tmp/sbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* img, int W, int H, const int* xs, const int* ys, float* out, int n) {
  const unsigned long long base = (unsigned long long)img;
  u32x4 r;
  r.x = (unsigned)base;
  r.y = (unsigned)(base >> 32) | ((unsigned)(W * 4) << 16);   // stride in bits [61:48]
  r.z = (unsigned)H;                                          // num_records (records of `stride` bytes)
  r.w = 0x00020000u;
  r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y); r.z = __builtin_amdgcn_readfirstlane(r.z); r.w = __builtin_amdgcn_readfirstlane(r.w);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32x2 a = {(unsigned)ys[i], (unsigned)xs[i] << 2};   // {index, byte offset}
  float v;
  asm volatile("buffer_load_dword %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(a), "s"(r) : "memory");
  out[i] = v;
}
int main() {
  const int W = 8, H = 4, PAD = 64;
  std::vector<float> h(W * H + PAD);
  for (int i = 0; i < W * H + PAD; i++) h[i] = 100.0f + i;
  float* d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int xs[] = {3, 7, 8, 9, -1, 2, 2, 2, 1 << 30, (1 << 30) + 3, 0, 15, 8};
  const int ys[] = {2, 3, 1, 0, 1, 4, -1, 5, 1, 1, 0, 3, 3};
  const int n = sizeof(xs) / 4;
  int *dx, *dy; float* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice); hipMemcpy(dy, ys, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, W, H, dx, dy, dout, n);
  std::vector<float> o(n); hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; i++) {
    const bool inside = xs[i] >= 0 && xs[i] < W && ys[i] >= 0 && ys[i] < H;
    printf("x=%11d y=%3d -> %8.1f   (%s; in-image value would be %.1f, linear address value %.1f)\n", xs[i], ys[i], o[i], inside ? "inside" : "OUTSIDE",
           inside ? 100.0f + ys[i] * W + xs[i] : 0.0f, (ys[i] >= 0 && (long)ys[i] * W + (long)(unsigned)(xs[i] << 2) / 4 < W * H + PAD) ? 100.0f + ys[i] * W + (unsigned)(xs[i] << 2) / 4 : -1.0f);
  }
  return 0;
}
