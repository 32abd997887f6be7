// volume_kernels.hip -- the volume pre-processing kernels next to the hot path:
//   fetch_stats  opencl_kernels/reference_volume_figures.cl:10-26  min/max of value and of |gradient|
//   apply_clip   opencl_kernels/reference_volume_clip.cl:4-15      copy of a sub-box
// Both are single HBM streams.  The reference issues four global atomics per voxel; here every wave
// reduces with cross-lane operations, every block through LDS, and only one atomic per block and
// statistic reaches memory.
#include "clwh_internal.hpp"
#include "device_math.hpp"

namespace clvr {

__device__ __forceinline__ int wave_min(int v) {
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
  return v;
}

__global__ __launch_bounds__(256) void k_fetch_stats(const int16_t *__restrict__ vol, int X, int Y, int Z, int32_t *stats) {
  __shared__ int s_red[4][4];
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, z = blockIdx.z;
  int vmin = 2147483647, vmax = -2147483647 - 1, gmin = 2147483647, gmax = -2147483647 - 1;
  if (x < X) {
    auto at = [&](int px, int py, int pz) -> int {  // border texel = 0
      if ((unsigned)px >= (unsigned)X || (unsigned)py >= (unsigned)Y || (unsigned)pz >= (unsigned)Z) return 0;
      return vol[((size_t)pz * (size_t)Y + (size_t)py) * (size_t)X + (size_t)px];
    };
    const int v = at(x, y, z);
    const float gx = (float)(at(x + 1, y, z) - at(x - 1, y, z));
    const float gy = (float)(at(x, y + 1, z) - at(x, y - 1, z));
    const float gz = (float)(at(x, y, z + 1) - at(x, y, z - 1));
    const int g = f2i(sqrtf((gx * gx + gy * gy) + gz * gz));  // atomic_min(int*, float): the float converts to int
    vmin = vmax = v;
    gmin = gmax = g;
  }
  vmin = wave_min(vmin); vmax = wave_max(vmax); gmin = wave_min(gmin); gmax = wave_max(gmax);
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (lane == 0u) { s_red[wave][0] = vmin; s_red[wave][1] = vmax; s_red[wave][2] = gmin; s_red[wave][3] = gmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned waves = (blockDim.x + 63u) >> 6;
    for (unsigned w = 1; w < waves; ++w) {
      vmin = min(vmin, s_red[w][0]); vmax = max(vmax, s_red[w][1]);
      gmin = min(gmin, s_red[w][2]); gmax = max(gmax, s_red[w][3]);
    }
    if (vmin <= vmax) {  // the block held at least one voxel
      atomicMin(&stats[0], vmin); atomicMax(&stats[1], vmax);
      atomicMin(&stats[2], gmin); atomicMax(&stats[3], gmax);
    }
  }
}

__global__ __launch_bounds__(256) void k_apply_clip(const int16_t *__restrict__ src, int SX, int SY, int SZ, int16_t *dst,
                                                    int DX, int DY, int DZ, const uint32_t *start, const uint32_t *len) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, z = blockIdx.z;
  // clipped_size = {end[0], end[1], end[2]}: the caller passes lengths (app/reference_volume.cpp:63)
  if (x >= (int)len[0] || y >= (int)len[1] || z >= (int)len[2] || x >= DX || y >= DY || z >= DZ) return;
  const int sx = (int)start[0] + x, sy = (int)start[1] + y, sz = (int)start[2] + z;
  int16_t v = 0;  // read_imagei outside the image: border 0
  if ((unsigned)sx < (unsigned)SX && (unsigned)sy < (unsigned)SY && (unsigned)sz < (unsigned)SZ)
    v = src[((size_t)sz * (size_t)SY + (size_t)sy) * (size_t)SX + (size_t)sx];
  dst[((size_t)z * (size_t)DY + (size_t)y) * (size_t)DX + (size_t)x] = v;
}

static unsigned row_block(int X) { return X <= 64 ? 64u : (X <= 128 ? 128u : 256u); }

hipError_t launch_fetch_stats(const int16_t *vol, int X, int Y, int Z, int32_t *stats, hipStream_t s) {
  const unsigned b = row_block(X);
  hipLaunchKernelGGL(k_fetch_stats, dim3(((unsigned)X + b - 1u) / b, (unsigned)Y, (unsigned)Z), dim3(b), 0, s, vol, X, Y, Z, stats);
  return hipGetLastError();
}

hipError_t launch_apply_clip(const int16_t *src, int SX, int SY, int SZ, int16_t *dst, int DX, int DY, int DZ,
                             const uint32_t *start, const uint32_t *len, hipStream_t s) {
  const unsigned b = row_block(DX);
  hipLaunchKernelGGL(k_apply_clip, dim3(((unsigned)DX + b - 1u) / b, (unsigned)DY, (unsigned)DZ), dim3(b), 0, s, src, SX, SY,
                     SZ, dst, DX, DY, DZ, start, len);
  return hipGetLastError();
}

}  // namespace clvr
