// valu_peak.hip -- what is the VALU issue ceiling of a gfx950 SIMD, in wave64 instructions per cycle?
//
// bench.py's roofline.frac for the batched integrate kernel is SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE x 1024 SIMDs): it assumes that a
// wave64 VALU instruction owns its SIMD for 4 cycles (a 16-lane datapath), i.e. that 0.25 wave-instructions per cycle and SIMD is the ceiling.
// MI355X_MICROARCH.md's wave-scheduling paragraph says 2 cycles (SIMD-32).  This tool measures it: streams of INDEPENDENT instructions of one
// class (8 accumulators, 64 instructions per loop trip), W = 1 / 2 / 4 / 8 waves per SIMD on every CU, timed by s_memtime inside each wave and by
// HIP events outside.  Every (class, W) is its own kernel symbol, so one `rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE`
// pass over this binary yields the very ratio the bench computes, on a stream whose true issue rate is known.
//
//   hipcc -O2 --offload-arch=gfx950 tools/gpu/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak [iters]
//
// Output: one JSON object; per (class, W): wave-instructions per cycle and SIMD from the waves' own clocks (mean over waves) and from the
// wall clock x the measured shader clock, the waves-per-SIMD census read from HW_ID, and the shader clock (s_memtime ticks per wall second).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

enum Op { FMA = 0, PK_FMA, ADD_U32, MUL_F32, CVT_F32_I32, RCP_F32, LERP_U8, PERM_B32, CNDMASK, MIN_F32, MAD_U32_U24, FMA_DEP, MIX_INTEGRATE, NUM_OPS };
static const char* kOpName[NUM_OPS] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_mul_f32", "v_cvt_f32_i32", "v_rcp_f32", "v_lerp_u8", "v_perm_b32", "v_cndmask_b32",
                                       "v_min_f32", "v_mad_u32_u24", "v_fma_f32 (one dependent chain)", "mix per 8: 2 v_fma_f32 + 2 v_pk_fma_f32 + 2 v_add_u32 + 1 v_lerp_u8 + 1 v_cvt_f32_i32"};

struct WaveRec { unsigned long long t0, t1; unsigned hw_id, xcc_id; };

template <int OP>
__global__ void __launch_bounds__(256) k_stream(WaveRec* rec, float* sink, int iters) {
  extern __shared__ char lds_pad[];   // only sizes the workgroups-per-CU limit
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  f2 pb = {0.999f, 1.001f}, pc = {1e-6f, -1e-6f};
  float b = 0.9999f, c = 1e-7f;
  unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7, ub = 0x01020304u, uc = 0x00070503u;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
  // one asm statement of 64 instructions per trip: nothing of the compiler's (hazard s_nops between statements, moves) sits inside the stream
#define X8(s) s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s
#define ROW3(INS) INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9"
#define ROW2(INS) INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8"
#define ROW1(INS) INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7"
#define FREGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
#define UREGS : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ub), "v"(uc)
#define PREGS : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc)
  for (int i = 0; i < iters; i++) {
    if (OP == FMA) asm volatile(X8(ROW3("v_fma_f32")) FREGS);
    else if (OP == PK_FMA) asm volatile(X8(ROW3("v_pk_fma_f32")) PREGS);
    else if (OP == ADD_U32) asm volatile(X8(ROW2("v_add_u32")) UREGS);
    else if (OP == LERP_U8) asm volatile(X8(ROW3("v_lerp_u8")) UREGS);
    else if (OP == PERM_B32) asm volatile(X8(ROW3("v_perm_b32")) UREGS);
    else if (OP == MAD_U32_U24) asm volatile(X8(ROW3("v_mad_u32_u24")) UREGS);
    else if (OP == MUL_F32) asm volatile(X8(ROW2("v_mul_f32")) FREGS);
    else if (OP == MIN_F32) asm volatile(X8(ROW2("v_min_f32")) FREGS);
    else if (OP == CVT_F32_I32) asm volatile(X8(ROW1("v_cvt_f32_i32")) FREGS);
    else if (OP == RCP_F32) asm volatile(X8(ROW1("v_rcp_f32")) FREGS);
    else if (OP == CNDMASK) asm volatile(X8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                                            "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc") FREGS : "vcc");
    else if (OP == FMA_DEP) asm volatile(X8(X8("v_fma_f32 %0, %0, %1, %2")) : "+v"(a0) : "v"(b), "v"(c));
    else if (OP == MIX_INTEGRATE)
      // the class shares of profiles/r04_pmc_valu_mix.txt (FMA 65.6 M, INT32 57.3 M, CVT 24.7 M, ADD 16.2 M, MUL 12.4 M of the 176 M full-rate ones) approximated by an
      // 8-instruction pattern: 2 fma + 2 pk_fma + 3 int + 1 cvt; the quarter-rate v_rcp / v_rsq (8.4 M of 270 M) are measured on their own
      asm volatile(X8("v_fma_f32 %0, %0, %8, %9\n v_add_u32 %3, %3, %10\n v_pk_fma_f32 %6, %6, %12, %13\n v_fma_f32 %1, %1, %8, %9\n v_add_u32 %4, %4, %10\n v_cvt_f32_i32 %2, %2\n"
                      "v_pk_fma_f32 %7, %7, %12, %13\n v_lerp_u8 %5, %5, %10, %11")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(p0), "+v"(p1) : "v"(b), "v"(c), "v"(ub), "v"(uc), "v"(pb), "v"(pc));
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if ((threadIdx.x & 63) == 0) {
    WaveRec r = {t0, t1, hw, xcc};
    rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
  if (s == 12345.678f) sink[0] = s;   // keeps the accumulators alive
}

typedef void (*Kern)(WaveRec*, float*, int);
template <int OP> static Kern kern() { return k_stream<OP>; }

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20000;   // 64 instructions per trip: 1.28 M instructions per wave, ~2 ms at 4 cycles each
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int wall_khz = 0;
  (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  WaveRec* d_rec;
  float* d_sink;
  const int max_blocks = cus * 8;
  CK(hipMalloc(&d_rec, sizeof(WaveRec) * max_blocks * 4));
  CK(hipMalloc(&d_sink, 64));
  std::vector<WaveRec> rec(max_blocks * 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  Kern ks[NUM_OPS] = {kern<FMA>(), kern<PK_FMA>(), kern<ADD_U32>(), kern<MUL_F32>(), kern<CVT_F32_I32>(), kern<RCP_F32>(), kern<LERP_U8>(), kern<PERM_B32>(), kern<CNDMASK>(),
                      kern<MIN_F32>(), kern<MAD_U32_U24>(), kern<FMA_DEP>(), kern<MIX_INTEGRATE>()};
  printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d, \"wall_clock_rate_khz\": %d, \"iters\": %d, \"instructions_per_wave\": %lld,\n \"streams\": [\n",
         prop.name, prop.gcnArchName, cus, prop.clockRate, wall_khz, iters, (long long)iters * 64);
  bool first = true;
  for (int op = 0; op < NUM_OPS; op++) {
    for (int W : {1, 2, 4, 8}) {
      // W workgroups of 4 waves per CU: the dynamic-LDS request caps the workgroups a CU can hold at W, the grid is W x CUs, so every SIMD carries W waves
      const int blocks = cus * W;
      const size_t lds = W == 1 ? 96 * 1024 : W == 2 ? 64 * 1024 : W == 4 ? 36 * 1024 : 18 * 1024;   // 160 KB per CU: W fit, W + 1 do not (8 x 4 waves is the wave limit too)
      CK(hipFuncSetAttribute((const void*)ks[op], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int it = (op == RCP_F32) ? iters / 4 : iters;
      hipLaunchKernelGGL(ks[op], dim3(blocks), dim3(256), lds, 0, d_rec, d_sink, 200);   // warm-up (clocks, code)
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(ks[op], dim3(blocks), dim3(256), lds, 0, d_rec, d_sink, it);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(rec.data(), d_rec, sizeof(WaveRec) * blocks * 4, hipMemcpyDeviceToHost));
      double sum = 0;
      unsigned long long tmin = ~0ull, tmax = 0, dmax = 0;
      std::map<unsigned long long, int> census;
      for (int w = 0; w < blocks * 4; w++) {
        const unsigned long long d = rec[w].t1 - rec[w].t0;
        sum += (double)d;
        dmax = std::max(dmax, d);
        tmin = std::min(tmin, rec[w].t0);
        tmax = std::max(tmax, rec[w].t1);
        // gfx9 HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13] (the wave-slot, pipe, queue, vm fields are masked out); XCC_ID [3:0]
        census[((unsigned long long)(rec[w].xcc_id & 0xf) << 32) | (rec[w].hw_id & 0xff30u)]++;
      }
      int cmin = 1 << 30, cmax = 0;
      for (auto& kv : census) { cmin = std::min(cmin, kv.second); cmax = std::max(cmax, kv.second); }
      const double mean_cyc = sum / (blocks * 4);
      const double insts = (double)it * 64;
      const double span = (double)(tmax - tmin);                       // first wave in to last wave out, s_memtime ticks
      const double ticks_per_s = span / (ms * 1e-3);                   // s_memtime ticks per wall second (event time includes launch latency: a lower bound)
      const double per_simd_waveclock = insts * W / mean_cyc;          // a SIMD issues W waves' streams in mean_cyc
      const double per_simd_span = insts * W / span;
      printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"event_ms\": %.4f, \"mean_wave_ticks\": %.0f, \"max_wave_ticks\": %llu, \"span_ticks\": %.0f, "
             "\"memtime_ticks_per_s\": %.4g, \"inst_per_tick_per_simd\": %.4f, \"inst_per_tick_per_simd_span\": %.4f, \"ticks_per_inst\": %.3f, "
             "\"simd_slots_seen\": %zu, \"waves_per_simd_census\": [%d, %d]}",
             first ? "" : ",\n", kOpName[op], W, ms, mean_cyc, dmax, span, ticks_per_s, per_simd_waveclock, per_simd_span, mean_cyc / (insts * W), census.size(), cmin, cmax);
      first = false;
    }
  }
  printf("\n ]}\n");
  return 0;
}
