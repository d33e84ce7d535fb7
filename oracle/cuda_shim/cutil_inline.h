// oracle/cuda_shim/cutil_inline.h -- TEST INFRASTRUCTURE.  What the reference's cuda_SimpleMatrixUtil.h / cudaUtil.h expect from the NVIDIA SDK's
// cutil headers (absent here) when its __device__ __host__ matrix classes are compiled as plain host C++ (oracle/Makefile: _ref/libref_unproject.so):
// the execution-space keywords as empty macros, the CUDA vector types it names, make_*, __int_as_float.  Nothing here is reference code.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#define __device__
#define __host__
#define __global__
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
#define cudaAssert(condition)   /* cudaUtil.h defines it as a printf; the matrix classes' asserts are not exercised here */
#define __CONDITIONAL_UNROLL__
