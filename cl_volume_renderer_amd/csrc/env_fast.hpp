// env_fast.hpp -- certified fast path of the environment-map texel lookup.
//
// The contract (DESIGN.md "Semantics") defines the texel through correctly rounded binary32 atan2 /
// asin followed by a chain of float operations that is monotone in the angle:
//     i = clamp(int(floor(((theta * 0.1591549431f) + 0.5f) * w)), 0, w-1),   theta = RN(atan2(d.x, d.z))
// The fast path evaluates a cheap polynomial approximation t of the angle whose absolute error against
// the true angle is far below kAngleBracket (checked exhaustively-by-sampling on the CPU in
// tests/test_env_fast.py: the functions below use IEEE +,*,/,sqrt,fma only, so host and gfx950
// evaluate them identically) and feeds BOTH ends of [t - kAngleBracket, t + kAngleBracket] through the
// exact chain.  RN(true angle) lies in that interval, the chain is monotone, so when both ends give the
// same texel that texel IS the contract's.  Otherwise (about one lookup in a thousand at 4096x2048) the
// caller takes the exact binary64 route.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CLVR_HD __host__ __device__ __forceinline__
#else
#define CLVR_HD inline
#endif

namespace clvr {

constexpr float kAngleBracket = 2.0e-6f;  // > 4x the measured worst error (3.0e-7 / 2.0e-7) plus half an ulp of pi

// atan(a)/a on [0,1] as a degree-9 polynomial in a^2 (Chebyshev fit, truncation error 1.8e-9)
CLVR_HD float atan_unit_approx(float a) {
  const float s = a * a;
  float p = -0.0017437011472916245f;
  p = fmaf(p, s, 0.010680719461477509f);
  p = fmaf(p, s, -0.030717508932651568f);
  p = fmaf(p, s, 0.057463557876651285f);
  p = fmaf(p, s, -0.0837206394969292f);
  p = fmaf(p, s, 0.10940198965715174f);
  p = fmaf(p, s, -0.14261573680400244f);
  p = fmaf(p, s, 0.19998230640387646f);
  p = fmaf(p, s, -0.3333328229551216f);
  p = fmaf(p, s, 0.9999999975460196f);
  return a * p;
}

// approximate atan2(y, x); NaN when x == y == 0 or an input is NaN (the caller then goes exact)
CLVR_HD float atan2_approx(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
  float r = atan_unit_approx(mn / mx);
  if (ay > ax) r = 1.57079637f - r;
  if (x < 0.0f) r = 3.14159274f - r;
  return copysignf(r, y);
}

// approximate asin(v), |v| <= 1; NaN outside (as the exact function)
CLVR_HD float asin_approx(float v) {
  const float c = sqrtf((1.0f - v) * (1.0f + v));
  return atan2_approx(v, c);
}

CLVR_HD int32_t env_f2i(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  int32_t r;  // saturating, NaN -> 0: the definition below in one instruction (device_math.hpp f2i)
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
#endif
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int32_t)v;
}

// the contract's chain from an angle to a texel coordinate (monotone non-decreasing in `angle`)
CLVR_HD int32_t env_coord(float angle, float inv_pi_factor, int32_t n) {
  float u = angle * inv_pi_factor;
  u = u + 0.5f;
  int32_t i = env_f2i(floorf(u * (float)n));
  i = i < 0 ? 0 : i;
  i = i > n - 1 ? n - 1 : i;
  return i;
}

// utility_environment_map.cl:3-13 texel of direction d, fast path.  Returns false when the bracket
// straddles a texel boundary (or the approximation is NaN): the caller must use the exact route.
CLVR_HD bool env_texel_fast(float dx, float dy, float dz, int32_t w, int32_t h, int32_t &oi, int32_t &oj) {
  const float t = atan2_approx(dx, dz);
  const float p = asin_approx(-dy);
  if (!(t == t) || !(p == p)) return false;
  const int32_t i0 = env_coord(t - kAngleBracket, 0.1591549431f, w), i1 = env_coord(t + kAngleBracket, 0.1591549431f, w);
  const int32_t j0 = env_coord(p - kAngleBracket, 0.318309886f, h), j1 = env_coord(p + kAngleBracket, 0.318309886f, h);
  oi = i0;
  oj = j0;
  return i0 == i1 && j0 == j1;
}

}  // namespace clvr
