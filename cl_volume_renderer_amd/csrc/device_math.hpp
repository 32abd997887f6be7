// device_math.hpp -- scalar/vector float helpers with FIXED evaluation order for gfx950.
//
// The render path's results are integer state (voxel cache, RGBA8) derived from float geometry,
// so the float semantics are part of the contract (DESIGN.md "Semantics"): every operation is a
// single IEEE-754 binary32 op in the order written (the library is compiled with
// -ffp-contract=off and correctly rounded division / sqrt), conversions truncate and saturate with
// NaN -> 0, and the three transcendental built-ins the path uses (atan2, asin, pow) return the
// correctly rounded binary32 value (evaluated in binary64, rounded once).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clvr {

struct f3 {
  float x, y, z;
};

__device__ __forceinline__ f3 make_f3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }

__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
  return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float length3(f3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ f3 normalize3(f3 a) {
  const float l = length3(a);
  return f3{a.x / l, a.y / l, a.z / l};
}

// OpenCL min/max on floats: min(a,b) = b < a ? b : a ; max(a,b) = a < b ? b : a
__device__ __forceinline__ float cl_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float cl_max(float a, float b) { return (a < b) ? b : a; }

// float -> int32 / uint32: truncate toward zero, saturate, NaN -> 0.  That is exactly what V_CVT_I32_F32 / V_CVT_U32_F32 do on
// this hardware (out-of-range values and infinities saturate, NaN converts to 0, negative values convert to 0u), so the contract's
// conversion is ONE instruction; written out in C++ it compiled to three compares, three selects and a branch around the convert --
// two dozen instructions, ten times per event phase of k_bounce.  tests/test_gpu_device_math.py checks the instruction against the
// written-out definition on every special value and on random bit patterns.
__device__ __forceinline__ int32_t f2i(float v) {
  int32_t r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ uint32_t f2u(float v) {
  uint32_t r;
  asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
// the definition, written out (the oracle's form): what the instructions above must equal
__device__ __forceinline__ int32_t f2i_reference(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int32_t)v;
}
__device__ __forceinline__ uint32_t f2u_reference(float v) {
  if (v != v) return 0u;
  if (v >= 4294967296.0f) return 0xFFFFFFFFu;
  if (v <= 0.0f) return 0u;
  return (uint32_t)v;
}

// correctly rounded binary32 transcendental results via binary64
__device__ __forceinline__ float cr_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ __forceinline__ float cr_asinf(float v) { return (float)asin((double)v); }
__device__ __forceinline__ float cr_powf(float a, float b) { return (float)pow((double)a, (double)b); }

// utility_sampling.cl:13-21
__device__ __forceinline__ uint32_t hash_u32(uint32_t seed) {
  seed = (seed ^ 61u) ^ (seed >> 16);
  seed <<= 3;
  seed ^= (seed >> 4);
  seed *= 0xDEADBEEFu;
  seed ^= (seed >> 15);
  return seed;
}

}  // namespace clvr
