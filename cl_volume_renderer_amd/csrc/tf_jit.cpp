// tf_jit.cpp -- general fallback for transfer-function source outside the rule grammar.
//
// The reference JIT-compiles WHATEVER `is_event_gen` the caller prepends (clw_function.hpp:74-111).  The
// rule parser (tf_parse.cpp) covers everything the application generates; anything else is compiled here
// with hiprtc -- not into the render kernels, but into one tiny classification kernel that evaluates the
// user's function once per voxel (value, |gradient| at the voxel) and writes a class byte: 0 = no event,
// k = event with the k-th distinct colour the function produced.  From there on the precompiled kernels
// run unchanged on class bytes and a colour palette (at most CLWH_TF_MAX_RULES distinct colours).
// Limitation: positions whose gradient taps round across an integer (render_device.hpp, classify_step)
// use the voxel's class instead of the reference's literal 7-fetch evaluation.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstring>
#include <string>
#include <vector>

#include "clwh_internal.hpp"

namespace clvr {

static const char *kPrelude = R"(
struct tf_int4 { int x, y, z, w; };
#define int4 tf_int4
#define uint4 tf_int4
#define inline __device__ inline
)";

static const char *kClassifier = R"(
#undef inline
#undef int4
#undef uint4
__device__ unsigned char clvr_tf_class_of(short value, short gradient, unsigned long long *palette, int max_colors, int *error) {
  tf_int4 c = {-1, -1, -1, -1};
  if (!is_event_gen(value, gradient, &c)) return 0;
  const bool wrote = !(c.x == -1 && c.y == -1 && c.z == -1 && c.w == -1);
  const unsigned long long key = wrote ? (0x100000000ull | (unsigned)(c.x & 255) | ((unsigned)(c.y & 255) << 8) |
                                          ((unsigned)(c.z & 255) << 16) | ((unsigned)(c.w & 255) << 24))
                                       : 0ull;
  for (int k = 0; k < max_colors; ++k) {
    const unsigned long long prev = atomicCAS(&palette[k], ~0ull, key);
    if (prev == ~0ull || prev == key) return (unsigned char)(k + 1);
  }
  *error = 1;
  return 0;
}

// `border_class` receives the class of the border texel: value 0 with gradient 0 (what a position with a NaN coordinate
// reads; a position with a coordinate == dimension reads value 0 too, with whatever gradient its in-range taps give --
// approximated by this class)
extern "C" __global__ void clvr_tf_classify(const short *vol, int X, int Y, int Z, unsigned char *cls,
                                            unsigned long long *palette, int max_colors, int *error, int *border_class) {
  const size_t n = (size_t)X * (size_t)Y * (size_t)Z;
  if (blockIdx.x == 0 && threadIdx.x == 0) *border_class = clvr_tf_class_of((short)0, (short)0, palette, max_colors, error);
  // grid-stride: a launch may not exceed 2^32 work-items (2048^3 voxels = 2^33)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
  const int x = (int)(i % (size_t)X), y = (int)((i / (size_t)X) % (size_t)Y), z = (int)(i / ((size_t)X * (size_t)Y));
  auto at = [&](int px, int py, int pz) -> int {
    if ((unsigned)px >= (unsigned)X || (unsigned)py >= (unsigned)Y || (unsigned)pz >= (unsigned)Z) return 0;
    return vol[((size_t)pz * (size_t)Y + (size_t)py) * (size_t)X + (size_t)px];
  };
  const float gx = (float)(at(x + 1, y, z) - at(x - 1, y, z));
  const float gy = (float)(at(x, y + 1, z) - at(x, y - 1, z));
  const float gz = (float)(at(x, y, z + 1) - at(x, y, z - 1));
  float len = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz)));
  int gi = (len != len) ? 0 : (len >= 2147483648.0f ? 2147483647 : (int)len);
  const unsigned char out = clvr_tf_class_of((short)vol[i], (short)gi, palette, max_colors, error);
  cls[i] = out;
  }
}
)";

// compile prelude + user source + classifier for gfx950; `log` receives the compiler's messages
int tf_jit_compile(const char *user_source, std::vector<char> &code, std::string &log) {
  std::string src = std::string(kPrelude) + user_source + "\n" + kClassifier;
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "is_event_gen.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return CLWH_ERR_TF_UNSUPPORTED;
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off"};
  const hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  size_t ls = 0;
  if (hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
    log.resize(ls);
    (void)hiprtcGetProgramLog(prog, &log[0]);
  }
  int rc = CLWH_ERR_TF_UNSUPPORTED;
  if (r == HIPRTC_SUCCESS) {
    size_t cs = 0;
    if (hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs > 0) {
      code.resize(cs);
      if (hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS) rc = CLWH_OK;
    }
  }
  (void)hiprtcDestroyProgram(&prog);
  return rc;
}

}  // namespace clvr
