// valu_peak.hip -- what is the VALU issue ceiling of a gfx950 SIMD, in wave64 instructions per cycle?
//
// bench.py's roofline.frac for the batched integrate kernel is SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE x 1024 SIMDs): it assumes that a
// wave64 VALU instruction owns its SIMD for 4 cycles (a 16-lane datapath), i.e. that 0.25 wave-instructions per cycle and SIMD is the ceiling.
// MI355X_MICROARCH.md's wave-scheduling paragraph says 2 cycles (SIMD-32).  This tool measures it: streams of INDEPENDENT instructions of one
// class (8 accumulators, 64 instructions per loop trip), W = 1 / 2 / 4 / 8 waves per SIMD on every CU, timed by s_memtime inside each wave and by
// HIP events outside.  Every (class, W) is its own kernel symbol, so one `rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE`
// pass over this binary yields the very ratio the bench computes, on a stream whose true issue rate is known.
//
//   hipcc -O2 -std=c++20 --offload-arch=gfx950 tools/gpu/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak [iters]
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

struct WaveRec { unsigned long long t0, t1; unsigned hw_id, xcc_id; };

// one asm statement of 64 instructions per trip: nothing of the compiler's (hazard s_nops between statements, moves) sits inside the stream
#define X8(s) s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s
#define ROW3(INS) INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9"
#define ROW2(INS) INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8"
#define ROW2R(INS) INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n" INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7"
#define ROW1(INS) INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7"
#define ROWT(INS, TAIL) INS " %0, %0, %8" TAIL "\n" INS " %1, %1, %8" TAIL "\n" INS " %2, %2, %8" TAIL "\n" INS " %3, %3, %8" TAIL "\n" INS " %4, %4, %8" TAIL "\n" INS " %5, %5, %8" TAIL "\n" INS " %6, %6, %8" TAIL "\n" INS " %7, %7, %8" TAIL
#define FREGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(m64)
#define UREGS : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ub), "v"(uc), "s"(m64)
#define PREGS : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc), "s"(m64)
#define LREGS : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(l4), "+v"(l5), "+v"(l6), "+v"(l7) : "v"(ub), "v"(uc), "s"(m64)

// STREAM(name, the 64 instructions, their registers [, clobbers]): a kernel of its own per stream, so that a counter pass reports each apart
#define STREAM(NAME, ASM, ...)                                                                                                                    \
  __global__ void __launch_bounds__(256) k_##NAME(WaveRec* rec, float* sink, int iters) {                                                        \
    extern __shared__ char lds_pad[]; /* only sizes the workgroups-per-CU limit */                                                               \
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;             \
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};                    \
    f2 pb = {0.999f, 1.001f}, pc = {1e-6f, -1e-6f};                                                                                              \
    float b = 0.9999f, c = 1e-7f;                                                                                                                \
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7, ub = 0x01020304u, uc = 3u; \
    unsigned long long l0 = u0, l1 = u1, l2 = u2, l3 = u3, l4 = u4, l5 = u5, l6 = u6, l7 = u7;                                                    \
    unsigned long long m64 = 0x5555aaaa3333ccccull + (unsigned long long)iters; /* a lane mask in an SGPR pair */                                 \
    unsigned long long t0, t1;                                                                                                                   \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));                                                                              \
    for (int i = 0; i < iters; i++) asm volatile(ASM __VA_OPT__(,) __VA_ARGS__);                                                                             \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));                                                                              \
    unsigned hw, xcc;                                                                                                                            \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                                                             \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                                                           \
    if ((threadIdx.x & 63) == 0) {                                                                                                               \
      WaveRec r = {t0, t1, hw, xcc};                                                                                                             \
      rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;                                                                                              \
    }                                                                                                                                            \
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + \
              (float)(l0 ^ l1 ^ l2 ^ l3 ^ l4 ^ l5 ^ l6 ^ l7);                                                                                     \
    if (s == 12345.678f) sink[0] = s; /* keeps the accumulators alive */                                                                          \
  }

// ---- fp32
STREAM(v_fma_f32, X8(ROW3("v_fma_f32")) FREGS)
STREAM(v_fmac_f32, X8(ROW2R("v_fmac_f32")) FREGS)
STREAM(v_mul_f32, X8(ROW2("v_mul_f32")) FREGS)
STREAM(v_add_f32, X8(ROW2("v_add_f32")) FREGS)
STREAM(v_sub_f32, X8(ROW2("v_sub_f32")) FREGS)
STREAM(v_min_f32, X8(ROW2("v_min_f32")) FREGS)
STREAM(v_max_f32, X8(ROW2("v_max_f32")) FREGS)
STREAM(v_min3_f32, X8(ROW3("v_min3_f32")) FREGS)
STREAM(v_pk_fma_f32, X8(ROW3("v_pk_fma_f32")) PREGS)
STREAM(v_pk_mul_f32, X8(ROW2("v_pk_mul_f32")) PREGS)
STREAM(v_pk_add_f32, X8(ROW2("v_pk_add_f32")) PREGS)
STREAM(v_rcp_f32, X8(ROW1("v_rcp_f32")) FREGS)
STREAM(v_fma_f32_dependent_chain, X8(X8("v_fma_f32 %0, %0, %8, %9")) FREGS)
// ---- conversions
STREAM(v_cvt_f32_i32, X8(ROW1("v_cvt_f32_i32")) FREGS)
STREAM(v_cvt_i32_f32, X8(ROW1("v_cvt_i32_f32")) FREGS)
STREAM(v_cvt_f32_ubyte3, X8(ROW1("v_cvt_f32_ubyte3")) FREGS)
STREAM(v_floor_f32, X8(ROW1("v_floor_f32")) FREGS)
// ---- moves, logic, shifts, integer
STREAM(v_mov_b32, X8("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8") UREGS)
STREAM(v_add_u32, X8(ROW2("v_add_u32")) UREGS)
STREAM(v_add_u32_clamp, X8(ROWT("v_add_u32_e64", " clamp")) UREGS)
STREAM(v_sub_u32, X8(ROW2("v_sub_u32")) UREGS)
STREAM(v_and_b32, X8(ROW2("v_and_b32")) UREGS)
STREAM(v_or_b32, X8(ROW2("v_or_b32")) UREGS)
STREAM(v_lshlrev_b32, X8(ROW2R("v_lshlrev_b32")) UREGS)
STREAM(v_ashrrev_i32, X8(ROW2R("v_ashrrev_i32")) UREGS)
STREAM(v_lshlrev_b32_sdwa, X8(ROWT("v_lshlrev_b32_sdwa", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3")) UREGS)
STREAM(v_mul_u32_u24, X8(ROW2("v_mul_u32_u24")) UREGS)
STREAM(v_mul_lo_u32, X8(ROW2("v_mul_lo_u32")) UREGS)
STREAM(v_mad_u32_u24, X8(ROW3("v_mad_u32_u24")) UREGS)
STREAM(v_add_lshl_u32, X8(ROW3("v_add_lshl_u32")) UREGS)
STREAM(v_lshl_add_u32, X8(ROW3("v_lshl_add_u32")) UREGS)
STREAM(v_add3_u32, X8(ROW3("v_add3_u32")) UREGS)
STREAM(v_and_or_b32, X8(ROW3("v_and_or_b32")) UREGS)
STREAM(v_bfe_u32, X8(ROW3("v_bfe_u32")) UREGS)
STREAM(v_alignbit_b32, X8(ROW3("v_alignbit_b32")) UREGS)
STREAM(v_lerp_u8, X8(ROW3("v_lerp_u8")) UREGS)
STREAM(v_perm_b32, X8(ROW3("v_perm_b32")) UREGS)
STREAM(v_lshl_add_u64, X8("v_lshl_add_u64 %0, %0, 1, %0\n v_lshl_add_u64 %1, %1, 1, %1\n v_lshl_add_u64 %2, %2, 1, %2\n v_lshl_add_u64 %3, %3, 1, %3\n"
                          "v_lshl_add_u64 %4, %4, 1, %4\n v_lshl_add_u64 %5, %5, 1, %5\n v_lshl_add_u64 %6, %6, 1, %6\n v_lshl_add_u64 %7, %7, 1, %7") LREGS)
// ---- compares and selects
STREAM(v_cmp_gt_f32_to_vcc, X8("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
                               "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8") FREGS : "vcc")
STREAM(v_cmp_gt_u32_to_sgprs, X8("v_cmp_gt_u32_e64 s[20:21], %0, %8\n v_cmp_gt_u32_e64 s[22:23], %1, %8\n v_cmp_gt_u32_e64 s[24:25], %2, %8\n v_cmp_gt_u32_e64 s[26:27], %3, %8\n"
                                 "v_cmp_gt_u32_e64 s[20:21], %4, %8\n v_cmp_gt_u32_e64 s[22:23], %5, %8\n v_cmp_gt_u32_e64 s[24:25], %6, %8\n v_cmp_gt_u32_e64 s[26:27], %7, %8")
       UREGS : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
STREAM(v_cndmask_b32_vcc, X8(ROWT("v_cndmask_b32", ", vcc")) FREGS : "vcc")
STREAM(v_cndmask_b32_vcc_set_once, "s_mov_b64 vcc, %10\n" X8(ROWT("v_cndmask_b32", ", vcc")) FREGS : "vcc")
STREAM(v_cndmask_b32_sgpr_mask, X8(ROWT("v_cndmask_b32_e64", ", %10")) FREGS)
STREAM(v_cmp_then_cndmask, X8("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc") FREGS : "vcc")
STREAM(v_cmp_sgprs_then_cndmask, X8("v_cmp_gt_f32_e64 s[20:21], %0, %8\n v_cmp_gt_f32_e64 s[22:23], %2, %8\n v_cmp_gt_f32_e64 s[24:25], %4, %8\n v_cmp_gt_f32_e64 s[26:27], %6, %8\n"
                                    "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n v_cndmask_b32_e64 %5, %5, %8, s[24:25]\n v_cndmask_b32_e64 %7, %7, %8, s[26:27]")
       FREGS : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
// ---- mixes
STREAM(mix_2fma_2pkfma_2add_1lerp_1cvt, X8("v_fma_f32 %0, %0, %8, %9\n v_add_u32 %3, %3, %10\n v_pk_fma_f32 %6, %6, %12, %13\n v_fma_f32 %1, %1, %8, %9\n v_add_u32 %4, %4, %10\n v_cvt_f32_i32 %2, %2\n"
                                           "v_pk_fma_f32 %7, %7, %12, %13\n v_lerp_u8 %5, %5, %10, %11")
       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(p0), "+v"(p1) : "v"(b), "v"(c), "v"(ub), "v"(uc), "v"(pb), "v"(pc))
STREAM(mix_fma_alternating_with_pk_fma, X8("v_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %4, %4, %10, %11\n v_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %5, %5, %10, %11\n"
                                           "v_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %6, %6, %10, %11\n v_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %7, %7, %10, %11")
       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(b), "v"(c), "v"(pb), "v"(pc))
STREAM(mix_fma_alternating_with_cvt, X8("v_fma_f32 %0, %0, %8, %9\n v_cvt_f32_i32 %4, %4\n v_fma_f32 %1, %1, %8, %9\n v_cvt_f32_i32 %5, %5\n"
                                        "v_fma_f32 %2, %2, %8, %9\n v_cvt_f32_i32 %6, %6\n v_fma_f32 %3, %3, %8, %9\n v_cvt_f32_i32 %7, %7") FREGS)

// ---- round two: which classes overlap?  (an alternation that runs at the faster class's rate means the two issue side by side)
#define ALT(A, B) A "\n" B "\n" A "\n" B "\n" A "\n" B "\n" A "\n" B
#define ALT4(I0, I1, I2, I3) I0 "\n" I1 "\n" I2 "\n" I3 "\n" I0 "\n" I1 "\n" I2 "\n" I3
#define MIXREGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u0), "+v"(u1), "+v"(p0), "+v"(p1) : "v"(b), "v"(c), "v"(ub), "v"(uc), "v"(pb), "v"(pc), "s"(m64)
STREAM(alt_fma_cmp, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_cmp_gt_f32 vcc, %2, %8", "v_fma_f32 %1, %1, %8, %9", "v_cmp_gt_f32 vcc, %3, %8")) MIXREGS : "vcc")
STREAM(alt_fma_cndmask_sgpr, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_cndmask_b32_e64 %2, %2, %8, %14", "v_fma_f32 %1, %1, %8, %9", "v_cndmask_b32_e64 %3, %3, %8, %14")) MIXREGS)
STREAM(alt_fma_min, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_min_f32 %2, %2, %8", "v_fma_f32 %1, %1, %8, %9", "v_min_f32 %3, %3, %8")) MIXREGS)
STREAM(alt_fma_perm, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_perm_b32 %4, %4, %10, %11", "v_fma_f32 %1, %1, %8, %9", "v_perm_b32 %5, %5, %10, %11")) MIXREGS)
STREAM(alt_fma_lerp, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_lerp_u8 %4, %4, %10, %11", "v_fma_f32 %1, %1, %8, %9", "v_lerp_u8 %5, %5, %10, %11")) MIXREGS)
STREAM(alt_fma_lshl, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_lshlrev_b32 %4, %11, %4", "v_fma_f32 %1, %1, %8, %9", "v_lshlrev_b32 %5, %11, %5")) MIXREGS)
STREAM(alt_fma_rcp, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_rcp_f32 %2, %2", "v_fma_f32 %1, %1, %8, %9", "v_rcp_f32 %3, %3")) MIXREGS)
STREAM(alt_add_u32_cvt, X8(ALT4("v_add_u32 %4, %4, %10", "v_cvt_f32_i32 %2, %2", "v_add_u32 %5, %5, %10", "v_cvt_f32_i32 %3, %3")) MIXREGS)
STREAM(alt_pk_fma_cvt, X8(ALT4("v_pk_fma_f32 %6, %6, %12, %13", "v_cvt_f32_i32 %2, %2", "v_pk_fma_f32 %7, %7, %12, %13", "v_cvt_f32_i32 %3, %3")) MIXREGS)
STREAM(alt_pk_fma_add_u32, X8(ALT4("v_pk_fma_f32 %6, %6, %12, %13", "v_add_u32 %4, %4, %10", "v_pk_fma_f32 %7, %7, %12, %13", "v_add_u32 %5, %5, %10")) MIXREGS)
STREAM(alt_cvt_cmp, X8(ALT4("v_cvt_f32_i32 %0, %0", "v_cmp_gt_f32 vcc, %2, %8", "v_cvt_f32_i32 %1, %1", "v_cmp_gt_f32 vcc, %3, %8")) MIXREGS : "vcc")
STREAM(ratio_2fma_1cvt, X8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_cvt_f32_i32 %2, %2\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_cvt_f32_i32 %3, %3\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9") MIXREGS)
STREAM(ratio_3fma_1cvt, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9", "v_fma_f32 %3, %3, %8, %9", "v_cvt_f32_i32 %2, %2")) MIXREGS)
STREAM(ratio_1fma_3cvt, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_cvt_f32_i32 %1, %1", "v_cvt_f32_i32 %3, %3", "v_cvt_f32_i32 %2, %2")) MIXREGS)
// ---- more single classes
STREAM(v_lshrrev_b32, X8(ROW2R("v_lshrrev_b32")) UREGS)
STREAM(v_xor_b32, X8(ROW2("v_xor_b32")) UREGS)
STREAM(v_bfi_b32, X8(ROW3("v_bfi_b32")) UREGS)
STREAM(v_max_u32, X8(ROW2("v_max_u32")) UREGS)
STREAM(v_mul_hi_u32, X8(ROW2("v_mul_hi_u32")) UREGS)
STREAM(v_mad_u64_u32_skipped_v_add_co_u32, X8("v_add_co_u32 %0, vcc, %0, %8\n v_add_co_u32 %1, vcc, %1, %8\n v_add_co_u32 %2, vcc, %2, %8\n v_add_co_u32 %3, vcc, %3, %8\n"
                                              "v_add_co_u32 %4, vcc, %4, %8\n v_add_co_u32 %5, vcc, %5, %8\n v_add_co_u32 %6, vcc, %6, %8\n v_add_co_u32 %7, vcc, %7, %8") UREGS : "vcc")
STREAM(v_cvt_u32_f32, X8(ROW1("v_cvt_u32_f32")) FREGS)
STREAM(v_trunc_f32, X8(ROW1("v_trunc_f32")) FREGS)
STREAM(v_med3_f32, X8(ROW3("v_med3_f32")) FREGS)
STREAM(v_add_f32_e64_abs, X8("v_add_f32_e64 %0, |%0|, %8\n v_add_f32_e64 %1, |%1|, %8\n v_add_f32_e64 %2, |%2|, %8\n v_add_f32_e64 %3, |%3|, %8\n"
                             "v_add_f32_e64 %4, |%4|, %8\n v_add_f32_e64 %5, |%5|, %8\n v_add_f32_e64 %6, |%6|, %8\n v_add_f32_e64 %7, |%7|, %8") FREGS)
STREAM(v_mul_f32_e64_clamp, X8(ROWT("v_mul_f32_e64", " clamp")) FREGS)
STREAM(v_fma_f32_neg, X8("v_fma_f32 %0, -%0, %8, %9\n v_fma_f32 %1, -%1, %8, %9\n v_fma_f32 %2, -%2, %8, %9\n v_fma_f32 %3, -%3, %8, %9\n"
                         "v_fma_f32 %4, -%4, %8, %9\n v_fma_f32 %5, -%5, %8, %9\n v_fma_f32 %6, -%6, %8, %9\n v_fma_f32 %7, -%7, %8, %9") FREGS)
STREAM(v_fma_f32_sgpr_operand, X8("v_fma_f32 %0, %0, s10, %9\n v_fma_f32 %1, %1, s10, %9\n v_fma_f32 %2, %2, s10, %9\n v_fma_f32 %3, %3, s10, %9\n"
                                  "v_fma_f32 %4, %4, s10, %9\n v_fma_f32 %5, %5, s10, %9\n v_fma_f32 %6, %6, s10, %9\n v_fma_f32 %7, %7, s10, %9") FREGS)
STREAM(v_mul_f32_literal, X8("v_mul_f32 %0, 0x3f7fff00, %0\n v_mul_f32 %1, 0x3f7fff00, %1\n v_mul_f32 %2, 0x3f7fff00, %2\n v_mul_f32 %3, 0x3f7fff00, %3\n"
                             "v_mul_f32 %4, 0x3f7fff00, %4\n v_mul_f32 %5, 0x3f7fff00, %5\n v_mul_f32 %6, 0x3f7fff00, %6\n v_mul_f32 %7, 0x3f7fff00, %7") FREGS)
STREAM(v_mov_b32_dpp_row_shr, X8("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                 "v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                 "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf") UREGS)

// ---- round three: is the overlap between neighbouring instructions of ONE wave, or between waves?
#define FMA4 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9"
#define CVT4 "v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3"
STREAM(clumps_4fma_4cvt, X8(FMA4 "\n" CVT4) MIXREGS)
STREAM(clumps_32fma_32cvt, FMA4 "\n" FMA4 "\n" FMA4 "\n" FMA4 "\n" FMA4 "\n" FMA4 "\n" FMA4 "\n" FMA4 "\n" CVT4 "\n" CVT4 "\n" CVT4 "\n" CVT4 "\n" CVT4 "\n" CVT4 "\n" CVT4 "\n" CVT4 MIXREGS)
STREAM(clumps_2fma_2cvt, X8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3") MIXREGS)
STREAM(alt_fma_cvt_dependent_pairs, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_cvt_f32_i32 %0, %0", "v_fma_f32 %1, %1, %8, %9", "v_cvt_f32_i32 %1, %1")) MIXREGS)
STREAM(alt_mul_cmp, X8(ALT4("v_mul_f32 %0, %0, %8", "v_cmp_gt_u32 vcc, %4, %10", "v_mul_f32 %1, %1, %8", "v_cmp_gt_u32 vcc, %5, %10")) MIXREGS : "vcc")
STREAM(alt_add_f32_mul24, X8(ALT4("v_add_f32 %0, %0, %8", "v_mul_u32_u24 %4, %4, %10", "v_add_f32 %1, %1, %8", "v_mul_u32_u24 %5, %5, %10")) MIXREGS)
STREAM(alt_mov_cvt, X8(ALT4("v_mov_b32 %4, %10", "v_cvt_f32_i32 %2, %2", "v_mov_b32 %5, %10", "v_cvt_f32_i32 %3, %3")) MIXREGS)
STREAM(alt_and_cvt, X8(ALT4("v_and_b32 %4, %4, %10", "v_cvt_f32_i32 %2, %2", "v_and_b32 %5, %5, %10", "v_cvt_f32_i32 %3, %3")) MIXREGS)
STREAM(alt_fma_add_u32, X8(ALT4("v_fma_f32 %0, %0, %8, %9", "v_add_u32 %4, %4, %10", "v_fma_f32 %1, %1, %8, %9", "v_add_u32 %5, %5, %10")) MIXREGS)
STREAM(alt_fma_rcp_1_to_3, X8(ALT4("v_rcp_f32 %2, %2", "v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9", "v_fma_f32 %3, %3, %8, %9")) MIXREGS)

// ---- round four: operands.  An SGPR source made v_fma_f32 a 4-cycle instruction above; which sources keep the 2-cycle rate?
#define SG3(INS, A, B) INS " %0, %0, " A ", " B "\n" INS " %1, %1, " A ", " B "\n" INS " %2, %2, " A ", " B "\n" INS " %3, %3, " A ", " B "\n" INS " %4, %4, " A ", " B "\n" INS " %5, %5, " A ", " B "\n" INS " %6, %6, " A ", " B "\n" INS " %7, %7, " A ", " B
#define SG2(INS, A) INS " %0, " A ", %0\n" INS " %1, " A ", %1\n" INS " %2, " A ", %2\n" INS " %3, " A ", %3\n" INS " %4, " A ", %4\n" INS " %5, " A ", %5\n" INS " %6, " A ", %6\n" INS " %7, " A ", %7"
STREAM(v_mul_f32_sgpr, X8(SG2("v_mul_f32", "s10")) FREGS)
STREAM(v_add_f32_sgpr, X8(SG2("v_add_f32", "s10")) FREGS)
STREAM(v_add_u32_sgpr, X8(SG2("v_add_u32", "s10")) UREGS)
STREAM(v_and_b32_sgpr, X8(SG2("v_and_b32", "s10")) UREGS)
STREAM(v_mov_b32_from_sgpr, X8("v_mov_b32 %0, s10\n v_mov_b32 %1, s11\n v_mov_b32 %2, s10\n v_mov_b32 %3, s11\n v_mov_b32 %4, s10\n v_mov_b32 %5, s11\n v_mov_b32 %6, s10\n v_mov_b32 %7, s11") UREGS)
STREAM(v_fma_f32_inline_constant, X8(SG3("v_fma_f32", "%8", "1.0")) FREGS)
STREAM(v_fma_f32_two_sgprs_same, X8(SG3("v_fma_f32", "s10", "s10")) FREGS)
STREAM(v_fmac_f32_sgpr, X8(SG2("v_fmac_f32", "s10")) FREGS)
STREAM(v_mul_f32_inline_constant, X8(SG2("v_mul_f32", "0.5")) FREGS)
STREAM(v_fma_f32_rotating_vgprs, X8("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %1, %2, %3, %4\n v_fma_f32 %2, %3, %4, %5\n v_fma_f32 %3, %4, %5, %6\n"
                                    "v_fma_f32 %4, %5, %6, %7\n v_fma_f32 %5, %6, %7, %0\n v_fma_f32 %6, %7, %0, %1\n v_fma_f32 %7, %0, %1, %2") FREGS)
STREAM(v_fma_f32_same_source_twice, X8("v_fma_f32 %0, %0, %0, %9\n v_fma_f32 %1, %1, %1, %9\n v_fma_f32 %2, %2, %2, %9\n v_fma_f32 %3, %3, %3, %9\n"
                                       "v_fma_f32 %4, %4, %4, %9\n v_fma_f32 %5, %5, %5, %9\n v_fma_f32 %6, %6, %6, %9\n v_fma_f32 %7, %7, %7, %9") FREGS)
STREAM(v_cvt_then_fma_sgpr, X8(ALT4("v_fma_f32 %0, %0, s10, %9", "v_cvt_f32_i32 %2, %2", "v_fma_f32 %1, %1, s10, %9", "v_cvt_f32_i32 %3, %3")) MIXREGS)
STREAM(v_cmp_gt_f32_sgpr_to_sgprs, X8("v_cmp_gt_f32_e64 s[20:21], s10, %0\n v_cmp_gt_f32_e64 s[22:23], s10, %1\n v_cmp_gt_f32_e64 s[24:25], s10, %2\n v_cmp_gt_f32_e64 s[26:27], s10, %3\n"
                                      "v_cmp_gt_f32_e64 s[20:21], s10, %4\n v_cmp_gt_f32_e64 s[22:23], s10, %5\n v_cmp_gt_f32_e64 s[24:25], s10, %6\n v_cmp_gt_f32_e64 s[26:27], s10, %7")
       FREGS : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
STREAM(v_sub_f32_then_min, X8(ALT4("v_sub_f32 %0, %0, %8", "v_min_f32 %2, %2, %9", "v_sub_f32 %1, %1, %8", "v_min_f32 %3, %3, %9")) MIXREGS)

typedef void (*Kern)(WaveRec*, float*, int);
struct Stream { const char* name; Kern k; int quarter; };
#define S_(NAME) {#NAME, k_##NAME, 0}
static const Stream kStreams[] = {
    S_(v_fma_f32), S_(v_fmac_f32), S_(v_mul_f32), S_(v_add_f32), S_(v_sub_f32), S_(v_min_f32), S_(v_max_f32), S_(v_min3_f32), S_(v_pk_fma_f32), S_(v_pk_mul_f32), S_(v_pk_add_f32),
    {"v_rcp_f32", k_v_rcp_f32, 1}, S_(v_fma_f32_dependent_chain), S_(v_cvt_f32_i32), S_(v_cvt_i32_f32), S_(v_cvt_f32_ubyte3), S_(v_floor_f32), S_(v_mov_b32), S_(v_add_u32),
    S_(v_add_u32_clamp), S_(v_sub_u32), S_(v_and_b32), S_(v_or_b32), S_(v_lshlrev_b32), S_(v_ashrrev_i32), S_(v_lshlrev_b32_sdwa), S_(v_mul_u32_u24), S_(v_mul_lo_u32), S_(v_mad_u32_u24),
    S_(v_add_lshl_u32), S_(v_lshl_add_u32), S_(v_add3_u32), S_(v_and_or_b32), S_(v_bfe_u32), S_(v_alignbit_b32), S_(v_lerp_u8), S_(v_perm_b32), S_(v_lshl_add_u64),
    S_(v_cmp_gt_f32_to_vcc), S_(v_cmp_gt_u32_to_sgprs), S_(v_cndmask_b32_vcc), S_(v_cndmask_b32_vcc_set_once), S_(v_cndmask_b32_sgpr_mask), S_(v_cmp_then_cndmask),
    S_(v_cmp_sgprs_then_cndmask), S_(mix_2fma_2pkfma_2add_1lerp_1cvt), S_(mix_fma_alternating_with_pk_fma), S_(mix_fma_alternating_with_cvt),
    S_(alt_fma_cmp), S_(alt_fma_cndmask_sgpr), S_(alt_fma_min), S_(alt_fma_perm), S_(alt_fma_lerp), S_(alt_fma_lshl), S_(alt_fma_rcp), S_(alt_add_u32_cvt), S_(alt_pk_fma_cvt),
    S_(alt_pk_fma_add_u32), S_(alt_cvt_cmp), S_(ratio_2fma_1cvt), S_(ratio_3fma_1cvt), S_(ratio_1fma_3cvt), S_(v_lshrrev_b32), S_(v_xor_b32), S_(v_bfi_b32), S_(v_max_u32), S_(v_mul_hi_u32),
    S_(v_mad_u64_u32_skipped_v_add_co_u32), S_(v_cvt_u32_f32), S_(v_trunc_f32), S_(v_med3_f32), S_(v_add_f32_e64_abs), S_(v_mul_f32_e64_clamp), S_(v_fma_f32_neg), S_(v_fma_f32_sgpr_operand),
    S_(v_mul_f32_literal), S_(v_mov_b32_dpp_row_shr),
    S_(clumps_4fma_4cvt), S_(clumps_32fma_32cvt), S_(clumps_2fma_2cvt), S_(alt_fma_cvt_dependent_pairs), S_(alt_mul_cmp), S_(alt_add_f32_mul24), S_(alt_mov_cvt), S_(alt_and_cvt), S_(alt_fma_add_u32),
    S_(alt_fma_rcp_1_to_3), S_(v_mul_f32_sgpr), S_(v_add_f32_sgpr), S_(v_add_u32_sgpr), S_(v_and_b32_sgpr), S_(v_mov_b32_from_sgpr), S_(v_fma_f32_inline_constant),
    S_(v_fma_f32_two_sgprs_same), S_(v_fmac_f32_sgpr), S_(v_mul_f32_inline_constant), S_(v_fma_f32_rotating_vgprs), S_(v_fma_f32_same_source_twice), S_(v_cvt_then_fma_sgpr),
    S_(v_cmp_gt_f32_sgpr_to_sgprs), S_(v_sub_f32_then_min)};
constexpr int NUM_OPS = (int)(sizeof(kStreams) / sizeof(kStreams[0]));

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
  printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d, \"wall_clock_rate_khz\": %d, \"iters\": %d, \"instructions_per_wave\": %lld,\n \"streams\": [\n",
         prop.name, prop.gcnArchName, cus, prop.clockRate, wall_khz, iters, (long long)iters * 64);
  bool first = true;
  for (int op = 0; op < NUM_OPS; op++) {
    for (int W : {1, 2, 4, 8}) {
      // W workgroups of 4 waves per CU: the dynamic-LDS request caps the workgroups a CU can hold at W, the grid is W x CUs, so every SIMD carries W waves
      const int blocks = cus * W;
      const size_t lds = W == 1 ? 96 * 1024 : W == 2 ? 64 * 1024 : W == 4 ? 36 * 1024 : 18 * 1024;   // 160 KB per CU: W fit, W + 1 do not (8 x 4 waves is the wave limit too)
      CK(hipFuncSetAttribute((const void*)kStreams[op].k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int it = kStreams[op].quarter ? iters / 4 : iters;
      hipLaunchKernelGGL(kStreams[op].k, dim3(blocks), dim3(256), lds, 0, d_rec, d_sink, 200);   // warm-up (clocks, code)
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(kStreams[op].k, dim3(blocks), dim3(256), lds, 0, d_rec, d_sink, it);
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
             first ? "" : ",\n", kStreams[op].name, W, ms, mean_cyc, dmax, span, ticks_per_s, per_simd_waveclock, per_simd_span, mean_cyc / (insts * W), census.size(), cmin, cmax);
      first = false;
    }
  }
  printf("\n ]}\n");
  return 0;
}
