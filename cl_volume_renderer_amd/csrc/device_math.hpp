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

// the smallest value of the wave (the same in every lane): four DPP steps inside the rows of 16 lanes (pairs and quads by quad permutes,
// eights by the half-row mirror, sixteens by the row mirror -- every source lane exists), then the four rows through scalar registers; six
// __shfl_xor steps are six LDS permutes with their address arithmetic.  Checked against that loop by clwh_debug_wave_min.
template <int CTRL>
__device__ __forceinline__ uint32_t wave_min_dpp_step(uint32_t t) {
  const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)t, CTRL, 0xF, 0xF, false);
  return other < t ? other : t;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  uint32_t t = wave_min_dpp_step<0xB1>(v);  // quad_perm [1,0,3,2]
  t = wave_min_dpp_step<0x4E>(t);           // quad_perm [2,3,0,1]
  t = wave_min_dpp_step<0x141>(t);          // row_half_mirror
  t = wave_min_dpp_step<0x140>(t);          // row_mirror
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)t, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)t, 16),
                 r2 = (uint32_t)__builtin_amdgcn_readlane((int)t, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)t, 48);
  const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
  return a < b ? a : b;
}
__device__ __forceinline__ uint32_t wave_min_u32_reference(uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, off);
    v = o < v ? o : v;
  }
  return v;
}

}  // namespace clvr
