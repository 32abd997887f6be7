// Host build of csrc/env_fast.hpp for tests/test_env_fast.py: the approximations use IEEE operations
// only, so what is measured here on the CPU is what gfx950 computes.
#include "../../cl_volume_renderer_amd/csrc/env_fast.hpp"

extern "C" {
float probe_atan2_approx(float y, float x) { return clvr::atan2_approx(y, x); }
float probe_asin_approx(float v) { return clvr::asin_approx(v); }
float probe_bracket(void) { return clvr::kAngleBracket; }
int probe_env_texel_fast(float dx, float dy, float dz, int w, int h, int *ij) {
  int32_t i = 0, j = 0;
  const bool ok = clvr::env_texel_fast(dx, dy, dz, w, h, i, j);
  ij[0] = i;
  ij[1] = j;
  return ok ? 1 : 0;
}
// worst absolute error of the two approximations against binary64 libm over n samples of a
// deterministic low-discrepancy sweep (plus the caller's special values)
void probe_sweep(long n, double *max_err_atan2, double *max_err_asin) {
  double ea = 0.0, es = 0.0;
  for (long k = 0; k < n; ++k) {
    const double u = (k + 0.5) / (double)n;
    const double ang = (2.0 * u - 1.0) * 3.14159265358979323846;
    const double rad = 0.001 + 1000.0 * (double)((k * 2654435761u) & 0xFFFF) / 65536.0;
    const float y = (float)(rad * sin(ang)), x = (float)(rad * cos(ang));
    const double d = fabs((double)clvr::atan2_approx(y, x) - atan2((double)y, (double)x));
    if (d > ea) ea = d;
    const float v = (float)(2.0 * u - 1.0);
    const double e = fabs((double)clvr::asin_approx(v) - asin((double)v));
    if (e > es) es = e;
    // cluster near +-1 where asin is ill-conditioned
    const float v2 = (float)(1.0 - u * 1e-4);
    const double e2 = fabs((double)clvr::asin_approx(v2) - asin((double)v2));
    if (e2 > es) es = e2;
    const double e3 = fabs((double)clvr::asin_approx(-v2) - asin((double)-v2));
    if (e3 > es) es = e3;
  }
  *max_err_atan2 = ea;
  *max_err_asin = es;
}
}
