// volume_kernels.hip -- the volume pre-processing kernels next to the hot path:
//   fetch_stats  opencl_kernels/reference_volume_figures.cl:10-26  min/max of value and of |gradient|
//   apply_clip   opencl_kernels/reference_volume_clip.cl:4-15      copy of a sub-box
//   bilateral_filter  opencl_kernels/volume_filter.cl:5-11         5x5x5 bilateral filter (LDS tile, host-built weight table)
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

// ------------------------------------------------------------------------------------------------
// bilateral_filter  opencl_kernels/volume_filter.cl:5-11 with bilateral_kernel utility_filter.cl:38-62:
// 5x5x5 bilateral filter of the short volume (sigma_s 0.6, sigma_r 1).  The weight of a tap is
// exp(-r2/(2*0.6^2) - d^2/2) with r2 the squared integer offset (13 values) and d the integer difference
// to the centre voxel; it rounds to zero for |d| >= 16, so all weights the kernel can ever use are a
// 13 x 17 table the host evaluates once in binary64 (`weights`, clwh_runtime.hip) -- no transcendental on
// the device.  A block filters an 8x8x8 brick out of a 12^3 LDS tile (3.4 KB); the 125 taps are
// accumulated in the reference's z, y, x order, un-contracted, so the two float sums round identically.
// HBM traffic is one read and one write of the volume; the kernel is LDS/VALU bound (125 taps per voxel).
constexpr int kBfR = 2, kBfB = 8, kBfT = kBfB + 2 * kBfR, kBfD = 17;
__global__ __launch_bounds__(kBfB * kBfB * kBfB) void k_bilateral_filter(const int16_t *__restrict__ src, int X, int Y, int Z,
                                                                          int16_t *__restrict__ dst,
                                                                          const float *__restrict__ weights) {
  __shared__ int16_t tile[kBfT][kBfT][kBfT];
  __shared__ float wtab[13][kBfD];
  const int tid = threadIdx.x;
  const int bx = blockIdx.x * kBfB, by = blockIdx.y * kBfB, bz = blockIdx.z * kBfB;
  for (int i = tid; i < 13 * kBfD; i += kBfB * kBfB * kBfB) (&wtab[0][0])[i] = weights[i];
  for (int i = tid; i < kBfT * kBfT * kBfT; i += kBfB * kBfB * kBfB) {
    const int lx = i % kBfT, ly = (i / kBfT) % kBfT, lz = i / (kBfT * kBfT);
    const int gx = bx + lx - kBfR, gy = by + ly - kBfR, gz = bz + lz - kBfR;
    int16_t v = 0;  // read_imagei outside the image: border 0
    if ((unsigned)gx < (unsigned)X && (unsigned)gy < (unsigned)Y && (unsigned)gz < (unsigned)Z)
      v = src[((size_t)gz * (size_t)Y + (size_t)gy) * (size_t)X + (size_t)gx];
    tile[lz][ly][lx] = v;
  }
  __syncthreads();
  const int lx = tid % kBfB, ly = (tid / kBfB) % kBfB, lz = tid / (kBfB * kBfB);
  const int px = bx + lx, py = by + ly, pz = bz + lz;
  if (px >= X || py >= Y || pz >= Z) return;
  const int mid = tile[lz + kBfR][ly + kBfR][lx + kBfR];
  float out_colour = 0.0f, wp = 0.0f;
#pragma unroll
  for (int z = -kBfR; z <= kBfR; ++z)
#pragma unroll
    for (int y = -kBfR; y <= kBfR; ++y)
#pragma unroll
      for (int x = -kBfR; x <= kBfR; ++x) {
        const int local = tile[lz + kBfR + z][ly + kBfR + y][lx + kBfR + x];
        const int d = min(abs(mid - local), kBfD - 1);
        const float w = wtab[x * x + y * y + z * z][d];
        wp = wp + w;
        out_colour = out_colour + (float)local * w;
      }
  const float q = out_colour / wp;  // wp >= 1: the centre tap weighs exp(0)
  int32_t r = f2i(q);
  r = max(-32768, min(32767, r));
  dst[((size_t)pz * (size_t)Y + (size_t)py) * (size_t)X + (size_t)px] = (int16_t)r;
}

// ------------------------------------------------------------------------------------------------
// tf_sort_values  opencl_kernels/histogram.cl:4-32: 2-D histogram over (value, |gradient|) for the
// transfer-function editor.  The reference does one global atomic per voxel; CT data puts most voxels
// into a handful of bins, so here a thread walks 16 consecutive voxels merging equal bins, and a wave
// merges equal bins across its lanes before anything reaches memory.  Bins outside the width x height
// frame (value == max_value rounds to column `width`; values below min_value go negative -- both write
// out of bounds in the reference) are dropped.
__global__ __launch_bounds__(256) void k_tf_sort_values(const int16_t *__restrict__ vol, int X, int Y, int Z, uint32_t *frame,
                                                        int width, int height, float min_value, float max_value,
                                                        float min_gradient, float max_gradient) {
  constexpr int kRun = 16;
  const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * kRun;
  const int y = blockIdx.y, z = blockIdx.z;
  auto at = [&](int px, int py, int pz) -> int {
    if ((unsigned)px >= (unsigned)X || (unsigned)py >= (unsigned)Y || (unsigned)pz >= (unsigned)Z) return 0;
    return vol[((size_t)pz * (size_t)Y + (size_t)py) * (size_t)X + (size_t)px];
  };
  const float value_range = max_value - min_value;
  const float gradient_range = max_gradient - min_gradient;
  int run_bin = -1;
  uint32_t run_count = 0u;
  auto flush = [&]() {
    // merge equal bins across the wave: one atomic per distinct bin
    int bin = run_count ? run_bin : -1;
    unsigned long long todo = __ballot(bin >= 0);
    while (todo != 0ull) {
      const int leader = __ffsll((long long)todo) - 1;
      const int b = __shfl(bin, leader);
      const unsigned long long same = __ballot(bin == b);
      uint32_t c = (bin == b) ? run_count : 0u;
      for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
      if ((int)(threadIdx.x & 63u) == leader) atomicAdd(&frame[b], c);
      todo &= ~same;
    }
    run_bin = -1;
    run_count = 0u;
  };
  for (int k = 0; k < kRun; ++k) {
    const int x = x0 + k;
    int bin = -1;
    if (x < X) {
      const int ref_value = at(x, y, z);
      const float gx = (float)(at(x + 1, y, z) - at(x - 1, y, z));
      const float gy = (float)(at(x, y + 1, z) - at(x, y - 1, z));
      const float gz = (float)(at(x, y, z + 1) - at(x, y, z - 1));
      const float grad_length = sqrtf((gx * gx + gy * gy) + gz * gz);
      if (!(grad_length > max_gradient) && !((float)ref_value > max_value)) {
        const int px = f2i(roundf((((float)ref_value - min_value) / value_range) * (float)width));
        const int py = f2i(roundf(((grad_length - min_gradient) / gradient_range) * (float)height));
        if (px >= 0 && px < width && py >= 0 && py < height) bin = px * height + py;
      }
    }
    // the wave must reach flush() together: flush when any lane's bin changes
    const bool changes = bin != run_bin && run_count != 0u;
    if (__ballot(changes) != 0ull) flush();
    if (bin >= 0) {
      run_bin = bin;
      run_count += 1u;
    }
  }
  flush();
}

// tf_flush_color_frame  opencl_kernels/histogram.cl:34-69: bin count -> rank among the distinct counts ->
// grey level 20..255 (0 for empty bins), rows flipped
__global__ __launch_bounds__(256) void k_tf_flush_color_frame(uint32_t *color_frame, int fw, int fh, const int32_t *frame,
                                                              const int32_t *lookup, int lookup_len) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= fw || y >= fh) return;
  const int value = frame[x * fh + (fh - y - 1)];
  // `lookup` is sorted ascending (a std::set on the host): binary search instead of the linear scan
  int lo = 0, hi = lookup_len - 1, local_value = -1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int v = lookup[mid];
    if (v == value) { local_value = mid; break; }
    if (v < value) lo = mid + 1; else hi = mid - 1;
  }
  int result = 0;
  if (local_value > -1) result = (int)(20.0f + (((float)local_value) / (float)lookup_len) * (255.0f - 20.0f));
  const uint32_t c = (uint32_t)min(max(result, 0), 255);  // write_imagei on an UNSIGNED_INT8 image saturates
  color_frame[(size_t)y * fw + x] = c | (c << 8) | (c << 16) | (255u << 24);
}

static unsigned row_block(int X) { return X <= 64 ? 64u : (X <= 128 ? 128u : 256u); }

hipError_t launch_bilateral_filter(const int16_t *src, int X, int Y, int Z, int16_t *dst, const float *weights, hipStream_t s) {
  const dim3 grid(((unsigned)X + kBfB - 1u) / kBfB, ((unsigned)Y + kBfB - 1u) / kBfB, ((unsigned)Z + kBfB - 1u) / kBfB);
  hipLaunchKernelGGL(k_bilateral_filter, grid, dim3(kBfB * kBfB * kBfB), 0, s, src, X, Y, Z, dst, weights);
  return hipGetLastError();
}

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

hipError_t launch_tf_sort_values(const int16_t *vol, int X, int Y, int Z, uint32_t *frame, int width, int height,
                                 float min_v, float max_v, float min_g, float max_g, hipStream_t s) {
  const unsigned per_block = 64u * 16u;  // one wave per block: the flush is a wave-wide rendezvous
  hipLaunchKernelGGL(k_tf_sort_values, dim3(((unsigned)X + per_block - 1u) / per_block, (unsigned)Y, (unsigned)Z), dim3(64), 0, s,
                     vol, X, Y, Z, frame, width, height, min_v, max_v, min_g, max_g);
  return hipGetLastError();
}

hipError_t launch_tf_flush_color_frame(uint32_t *color_frame, int fw, int fh, const int32_t *frame, const int32_t *lookup,
                                       int lookup_len, hipStream_t s) {
  hipLaunchKernelGGL(k_tf_flush_color_frame, dim3((unsigned)(fw + 15) / 16u, (unsigned)(fh + 15) / 16u), dim3(256), 0, s,
                     color_frame, fw, fh, frame, lookup, lookup_len);
  return hipGetLastError();
}

}  // namespace clvr
