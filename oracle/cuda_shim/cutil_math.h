// oracle/cuda_shim/cutil_math.h -- TEST INFRASTRUCTURE: the three cutil_math helpers the reference's matrix header uses (see cutil_inline.h)
#pragma once
static inline float length(const float3& v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
static inline float3& operator/=(float3& a, float s) { a.x /= s; a.y /= s; a.z /= s; return a; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
