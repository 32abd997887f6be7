// volume_kernels.hip -- the volume pre-processing kernels next to the hot path:
//   fetch_stats  opencl_kernels/reference_volume_figures.cl:10-26  min/max of value and of |gradient|
//   apply_clip   opencl_kernels/reference_volume_clip.cl:4-15      copy of a sub-box
//   bilateral_filter  opencl_kernels/volume_filter.cl:5-11         5x5x5 bilateral filter (LDS tile, host-built weight table)
//   tf_sort_values / tf_flush_color_frame  opencl_kernels/histogram.cl   the transfer-function editor's 2-D histogram
// All are single HBM streams.  The reference issues four global atomics per voxel (statistics) or one (histogram); here
// persistent grids walk the volume, reduce in registers / LDS, and a block touches memory once per statistic or bin it met.
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

// Eight consecutive voxels of a row and, for the central differences of utility_filter.cl:2-35, their six neighbour taps (border texel 0).
// Rows of a multiple of 8 voxels are read with 16-byte loads; any other size voxel by voxel.
struct Voxels8 {
  int v[8], gx[8], gy[8], gz[8];  // value, v(x+1) - v(x-1), v(y+1) - v(y-1), v(z+1) - v(z-1)
};
__device__ __forceinline__ void load_voxels8(const int16_t *__restrict__ vol, int X, int Y, int Z, int x0, int y, int z, Voxels8 &o) {
  auto at = [&](int px, int py, int pz) -> int {  // border texel = 0
    if ((unsigned)px >= (unsigned)X || (unsigned)py >= (unsigned)Y || (unsigned)pz >= (unsigned)Z) return 0;
    return vol[((size_t)pz * (size_t)Y + (size_t)py) * (size_t)X + (size_t)px];
  };
  if ((X & 7) == 0) {  // x0 is a multiple of 8: the eight voxels exist and are 16-byte aligned
    const int16_t *own = vol + ((size_t)z * (size_t)Y + (size_t)y) * (size_t)X + (size_t)x0;
    auto load8 = [](const int16_t *p, int (&v)[8]) {
      const uint4 q = *reinterpret_cast<const uint4 *>(p);
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int h = 0; h < 8; ++h) v[h] = (int)(int16_t)(w[h >> 1] >> (16 * (h & 1)));
    };
    int ym[8], yp[8], zm[8], zp[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) ym[h] = yp[h] = zm[h] = zp[h] = 0;
    load8(own, o.v);
    if (y > 0) load8(own - X, ym);
    if (y + 1 < Y) load8(own + X, yp);
    if (z > 0) load8(own - (size_t)X * (size_t)Y, zm);
    if (z + 1 < Z) load8(own + (size_t)X * (size_t)Y, zp);
    const int left = x0 > 0 ? (int)own[-1] : 0, right = x0 + 8 < X ? (int)own[8] : 0;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      o.gx[h] = (h < 7 ? o.v[h < 7 ? h + 1 : 7] : right) - (h > 0 ? o.v[h > 0 ? h - 1 : 0] : left);
      o.gy[h] = yp[h] - ym[h];
      o.gz[h] = zp[h] - zm[h];
    }
  } else {
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      const int x = x0 + h;  // x >= X: the caller ignores the entry
      o.v[h] = at(x, y, z);
      o.gx[h] = at(x + 1, y, z) - at(x - 1, y, z);
      o.gy[h] = at(x, y + 1, z) - at(x, y - 1, z);
      o.gz[h] = at(x, y, z + 1) - at(x, y, z - 1);
    }
  }
}
__device__ __forceinline__ float gradient_length8(const Voxels8 &o, int h) {
  const float gx = (float)o.gx[h], gy = (float)o.gy[h], gz = (float)o.gz[h];
  return sqrtf((gx * gx + gy * gy) + gz * gz);
}

// ------------------------------------------------------------------------------------------------
// Column walk: every voxel with its six central-difference taps, each voxel of the volume fetched from memory about once.
// (load_voxels8 above issues five 16-byte loads per eight voxels -- the row and its four neighbour rows: a 5x read amplification that
// the L2 has to serve; fetch_stats ran at 1.15 TB/s of unique bytes on it.)  A block of 256 lanes owns a tile of 256 (x) x 8 (y) voxels
// and walks kColZ slices along z: a lane keeps the eight voxels of its (chunk, row) for the slices z - 1, z, z + 1 in registers (the z
// taps), and slice z goes through LDS once, with a one-voxel rim, for the y and x taps.  Rim rows / voxels are loaded one slice ahead
// like the core.  Needs rows of a multiple of 8 voxels (16-byte loads); border texel 0 (CLK_ADDRESS_CLAMP) everywhere outside.
// A barrier that orders LDS accesses only.  __syncthreads() is also a release fence for GLOBAL memory: it waits for every outstanding
// load of the wave (vmcnt(0)) -- which would force the next slice's prefetch below to land before slice z is even written to LDS.  The
// loaded values are consumed through registers (the compiler waits for exactly the load it needs), nothing here communicates through
// global memory inside a launch.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
constexpr int kColX = 256, kColY = 8, kColZ = 16, kColPitch = kColX + 16;  // LDS row: 7 unused shorts, x0 - 1, 256 voxels, x0 + 256, padding
struct ColumnTile {
  __attribute__((aligned(16))) int16_t rows[kColY + 2][kColPitch];
};
__device__ __forceinline__ void unpack8(uint4 q, int (&v)[8]) {
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int h = 0; h < 8; ++h) v[h] = (int)(int16_t)(w[h >> 1] >> (16 * (h & 1)));
}
// f(x0, y, z, value[8], gx[8], gy[8], gz[8]) for every group of eight voxels of the tile (tx, ty, tz); all 256 lanes of the block call it
template <class F>
__device__ __forceinline__ void walk_column_tile(const int16_t *__restrict__ vol, int X, int Y, int Z, int tx, int ty, int tz, ColumnTile &t, F &&f) {
  const int cx = (int)(threadIdx.x & 31u), ry = (int)(threadIdx.x >> 5);
  const int x0 = tx * kColX + cx * 8, y = ty * kColY + ry;
  const int z_begin = tz * kColZ, z_end = min(z_begin + kColZ, Z);
  const bool col_in = x0 < X && y < Y;  // the lane's eight voxels exist (X is a multiple of 8)
  // rim duties: row y - 1 (lanes of row 0), row y + 8 (lanes of row 7); voxel x - 1 (chunk 0), voxel x + 256 (chunk 31) of the lane's own row
  const int rim_y = ry == 0 ? y - 1 : (ry == kColY - 1 ? y + 1 : -1);
  const bool rim_row = (ry == 0 || ry == kColY - 1) && x0 < X && rim_y >= 0 && rim_y < Y;
  const int rim_x = cx == 0 ? x0 - 1 : (cx == 31 ? x0 + 8 : -1);
  const bool rim_vox = (cx == 0 || cx == 31) && y < Y && rim_x >= 0 && rim_x < X;
  auto slice = [&](int z, uint4 &core, uint4 &rrow, int &rvox) {
    core = uint4{0u, 0u, 0u, 0u}; rrow = uint4{0u, 0u, 0u, 0u}; rvox = 0;
    if ((unsigned)z >= (unsigned)Z) return;
    const int16_t *zs = vol + (size_t)z * (size_t)Y * (size_t)X;
    if (col_in) core = *reinterpret_cast<const uint4 *>(zs + (size_t)y * (size_t)X + (size_t)x0);
    if (rim_row) rrow = *reinterpret_cast<const uint4 *>(zs + (size_t)rim_y * (size_t)X + (size_t)x0);
    if (rim_vox) rvox = zs[(size_t)y * (size_t)X + (size_t)rim_x];
  };
  uint4 prev, cur, nxt, rim_cur, rim_nxt, unused_row;
  int rv_cur, rv_nxt, unused_vox;
  slice(z_begin - 1, prev, unused_row, unused_vox);
  slice(z_begin, cur, rim_cur, rv_cur);
  for (int z = z_begin; z < z_end; ++z) {
    slice(z + 1, nxt, rim_nxt, rv_nxt);  // in flight while slice z is processed
    lds_barrier();                        // the previous slice's taps have been read
    *reinterpret_cast<uint4 *>(&t.rows[ry + 1][8 + cx * 8]) = cur;
    if (ry == 0) *reinterpret_cast<uint4 *>(&t.rows[0][8 + cx * 8]) = rim_cur;
    if (ry == kColY - 1) *reinterpret_cast<uint4 *>(&t.rows[kColY + 1][8 + cx * 8]) = rim_cur;
    if (cx == 0) t.rows[ry + 1][7] = (int16_t)rv_cur;
    if (cx == 31) t.rows[ry + 1][8 + kColX] = (int16_t)rv_cur;
    lds_barrier();
    if (col_in) {
      int v[8], up[8], dn[8], zm[8], zp[8], gx[8], gy[8], gz[8];
      unpack8(cur, v);
      unpack8(*reinterpret_cast<const uint4 *>(&t.rows[ry][8 + cx * 8]), up);        // row y - 1
      unpack8(*reinterpret_cast<const uint4 *>(&t.rows[ry + 2][8 + cx * 8]), dn);    // row y + 1
      unpack8(prev, zm);
      unpack8(nxt, zp);
      const int left = t.rows[ry + 1][8 + cx * 8 - 1], right = t.rows[ry + 1][8 + cx * 8 + 8];
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        gx[h] = (h < 7 ? v[h < 7 ? h + 1 : 7] : right) - (h > 0 ? v[h > 0 ? h - 1 : 0] : left);
        gy[h] = dn[h] - up[h];
        gz[h] = zp[h] - zm[h];
      }
      f(x0, y, z, v, gx, gy, gz);
    }
    prev = cur; cur = nxt; rim_cur = rim_nxt; rv_cur = rv_nxt;
  }
}
// tiles of the volume for the walk; a persistent grid takes them round-robin
__device__ __forceinline__ size_t column_tiles(int X, int Y, int Z, int &ntx, int &nty) {
  ntx = (X + kColX - 1) / kColX;
  nty = (Y + kColY - 1) / kColY;
  return (size_t)ntx * (size_t)nty * (size_t)((Z + kColZ - 1) / kColZ);
}

// fetch_stats on the column walk (rows of a multiple of 8 voxels); the gradient statistic as in k_fetch_stats below
__global__ __launch_bounds__(256) void k_fetch_stats_columns(const int16_t *__restrict__ vol, int X, int Y, int Z, int32_t *stats) {
  __shared__ ColumnTile tile;
  __shared__ int s_red[4][4];
  int vmin = 2147483647, vmax = -2147483647 - 1, smin = 2147483647, smax = -1;
  int ntx, nty;
  const size_t n_tiles = column_tiles(X, Y, Z, ntx, nty);
  for (size_t ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
    const int tx = (int)(ti % (size_t)ntx), ty = (int)((ti / (size_t)ntx) % (size_t)nty), tz = (int)(ti / ((size_t)ntx * (size_t)nty));
    walk_column_tile(vol, X, Y, Z, tx, ty, tz, tile, [&](int, int, int, const int (&v)[8], const int (&gx)[8], const int (&gy)[8], const int (&gz)[8]) {
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const float fx = (float)gx[h], fy = (float)gy[h], fz = (float)gz[h];
        const int sb = (int)__float_as_uint((fx * fx + fy * fy) + fz * fz);  // >= +0: orders like an integer
        vmin = min(vmin, v[h]); vmax = max(vmax, v[h]);
        smin = min(smin, sb); smax = max(smax, sb);
      }
    });
  }
  int gmin = smax >= 0 ? f2i(sqrtf(__uint_as_float((uint32_t)smin))) : 2147483647;
  int gmax = smax >= 0 ? f2i(sqrtf(__uint_as_float((uint32_t)smax))) : -2147483647 - 1;
  vmin = wave_min(vmin); vmax = wave_max(vmax); gmin = wave_min(gmin); gmax = wave_max(gmax);
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  __syncthreads();
  if (lane == 0u) { s_red[wave][0] = vmin; s_red[wave][1] = vmax; s_red[wave][2] = gmin; s_red[wave][3] = gmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (unsigned w = 1; w < 4u; ++w) {
      vmin = min(vmin, s_red[w][0]); vmax = max(vmax, s_red[w][1]);
      gmin = min(gmin, s_red[w][2]); gmax = max(gmax, s_red[w][3]);
    }
    if (vmin <= vmax) {
      if (vmin < __hip_atomic_load(&stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&stats[0], vmin);
      if (vmax > __hip_atomic_load(&stats[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats[1], vmax);
      if (gmin < __hip_atomic_load(&stats[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&stats[2], gmin);
      if (gmax > __hip_atomic_load(&stats[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats[3], gmax);
    }
  }
}

// A persistent grid walks the volume, a lane eight consecutive voxels of a row at a time; a block reduces through cross-lane
// operations and LDS and touches the four statistics only if it improves them.  (The first version ran one block per 256 voxels with
// four same-address atomics each: 2 M atomics serialised at one L2 channel, 24 ms at 512^3 for a 0.27 GB stream.)
__global__ __launch_bounds__(256) void k_fetch_stats(const int16_t *__restrict__ vol, int X, int Y, int Z, int32_t *stats) {
  __shared__ int s_red[4][4];
  // The gradient statistic is min / max over the voxels of (int)sqrt(s), s = (gx^2 + gy^2) + gz^2 in binary32.  Correctly rounded
  // sqrt and the truncating conversion are both monotone, so min and max commute with them: the loop keeps the extremes of s (as
  // the bit patterns of non-negative floats, which order like integers) and the square root -- fifteen instructions with the
  // contract's rounding -- is taken twice per lane at the end instead of once per voxel.
  int vmin = 2147483647, vmax = -2147483647 - 1, smin = 2147483647, smax = -1;
  // work item = eight consecutive voxels of a row; a block takes 256 consecutive items (several rows when rows are short)
  const size_t per_row = ((size_t)X + 7u) / 8u, n_items = (size_t)Y * (size_t)Z * per_row;
  for (size_t item = (size_t)blockIdx.x * 256u + threadIdx.x; item < n_items; item += (size_t)gridDim.x * 256u) {
    const size_t row = item / per_row;
    const int x0 = (int)(item - row * per_row) * 8;
    const int z = (int)(row / (size_t)Y), y = (int)(row - (size_t)z * (size_t)Y);
    Voxels8 o;
    load_voxels8(vol, X, Y, Z, x0, y, z, o);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      if (x0 + h >= X) break;
      const float gx = (float)o.gx[h], gy = (float)o.gy[h], gz = (float)o.gz[h];
      const int sb = (int)__float_as_uint((gx * gx + gy * gy) + gz * gz);  // >= +0: no sign bit
      vmin = min(vmin, o.v[h]); vmax = max(vmax, o.v[h]);
      smin = min(smin, sb); smax = max(smax, sb);
    }
  }
  // atomic_min(int*, float): the float converts to int (reference_volume_figures.cl:20-23)
  int gmin = smax >= 0 ? f2i(sqrtf(__uint_as_float((uint32_t)smin))) : 2147483647;
  int gmax = smax >= 0 ? f2i(sqrtf(__uint_as_float((uint32_t)smax))) : -2147483647 - 1;
  vmin = wave_min(vmin); vmax = wave_max(vmax); gmin = wave_min(gmin); gmax = wave_max(gmax);
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (lane == 0u) { s_red[wave][0] = vmin; s_red[wave][1] = vmax; s_red[wave][2] = gmin; s_red[wave][3] = gmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (unsigned w = 1; w < 4u; ++w) {
      vmin = min(vmin, s_red[w][0]); vmax = max(vmax, s_red[w][1]);
      gmin = min(gmin, s_red[w][2]); gmax = max(gmax, s_red[w][3]);
    }
    if (vmin <= vmax) {  // the block held at least one voxel; a stale read below only costs an atomic that changes nothing
      if (vmin < __hip_atomic_load(&stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&stats[0], vmin);
      if (vmax > __hip_atomic_load(&stats[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats[1], vmax);
      if (gmin < __hip_atomic_load(&stats[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&stats[2], gmin);
      if (gmax > __hip_atomic_load(&stats[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats[3], gmax);
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

// the same copy, eight voxels (one 16-byte store) per lane: destination rows of a multiple of 8 voxels.  The source run starts at an
// arbitrary voxel (2-byte aligned): one unaligned 16-byte load; a run that crosses the image's x range or the clip length goes voxel by voxel.
struct __attribute__((packed, aligned(2))) Short8Unaligned { uint4 v; };
__global__ __launch_bounds__(256) void k_apply_clip8(const int16_t *__restrict__ src, int SX, int SY, int SZ, int16_t *dst,
                                                     int DX, int DY, int DZ, const uint32_t *start, const uint32_t *len) {
  const int per_row = DX >> 3;
  const int item = blockIdx.x * 256 + threadIdx.x;
  const int y = item / per_row, x = (item - y * per_row) * 8, z = blockIdx.y;
  const int lx = (int)len[0], ly = (int)len[1], lz = (int)len[2];
  if (y >= DY || y >= ly || z >= lz || x >= lx) return;
  const int sx = (int)start[0] + x, sy = (int)start[1] + y, sz = (int)start[2] + z;
  int16_t *out = dst + ((size_t)z * (size_t)DY + (size_t)y) * (size_t)DX + (size_t)x;
  const bool row_in = (unsigned)sy < (unsigned)SY && (unsigned)sz < (unsigned)SZ;
  const int16_t *in = src + ((size_t)(row_in ? sz : 0) * (size_t)SY + (size_t)(row_in ? sy : 0)) * (size_t)SX;
  if (x + 8 <= lx && sx >= 0 && sx + 8 <= SX) {
    uint4 v = uint4{0u, 0u, 0u, 0u};  // read_imagei outside the image: border 0
    if (row_in) v = reinterpret_cast<const Short8Unaligned *>(in + sx)->v;
    *reinterpret_cast<uint4 *>(out) = v;
    return;
  }
  for (int h = 0; h < 8 && x + h < lx; ++h) out[h] = (row_in && (unsigned)(sx + h) < (unsigned)SX) ? in[sx + h] : (int16_t)0;
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

// The same filter, two voxels per lane (x = 2 lx, 2 lx + 1): the kernel above is bound by VALU issue -- about ten instructions per tap
// and voxel -- so this one does the integer part of a tap for two taps and the float part for two voxels per instruction:
//   * the six tile values a row of taps needs (x - 2 .. x + 3 of the even voxel) arrive as three dwords instead of ten 16-bit reads;
//   * |centre - tap| for the two halves of a dword at once: a packed 16-bit subtract that SATURATES (a difference of two shorts does
//     not fit 16 bits, but every |d| >= 16 means the same weight 0), negate, max, min 16: four instructions for two taps;
//   * the two voxels' running sums as float2: v_pk_add_f32 / v_pk_mul_f32 keep each voxel's own sum, tap by tap in the reference's
//     z, y, x order, with a separate multiply and add exactly like the scalar code -- bit-identical results at 1.5 instead of 3
//     instructions per tap and voxel.
typedef short bf_short2 __attribute__((ext_vector_type(2)));
typedef float bf_float2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf_short2 bf_absdiff16(bf_short2 a, bf_short2 b) {
  const bf_short2 d = __builtin_elementwise_sub_sat(a, b);                 // clamps at +-32767 / -32768
  const bf_short2 n = __builtin_elementwise_sub_sat(bf_short2{0, 0}, d);   // -(-32768) clamps to 32767
  const bf_short2 m = __builtin_elementwise_max(d, n);
  return __builtin_elementwise_min(m, bf_short2{(short)(kBfD - 1), (short)(kBfD - 1)});
}
__global__ __launch_bounds__(kBfB * kBfB * kBfB / 2) void k_bilateral_filter2(const int16_t *__restrict__ src, int X, int Y, int Z,
                                                                              int16_t *__restrict__ dst, const float *__restrict__ weights) {
  __shared__ __attribute__((aligned(8))) int16_t tile[kBfT][kBfT][kBfT];
  __shared__ float wtab[13][kBfD];
  constexpr int kThreads = kBfB * kBfB * kBfB / 2;
  const int tid = threadIdx.x;
  const int bx = blockIdx.x * kBfB, by = blockIdx.y * kBfB, bz = blockIdx.z * kBfB;
  for (int i = tid; i < 13 * kBfD; i += kThreads) (&wtab[0][0])[i] = weights[i];
  for (int i = tid; i < kBfT * kBfT * kBfT; i += kThreads) {
    const int lx = i % kBfT, ly = (i / kBfT) % kBfT, lz = i / (kBfT * kBfT);
    const int gx = bx + lx - kBfR, gy = by + ly - kBfR, gz = bz + lz - kBfR;
    int16_t v = 0;  // read_imagei outside the image: border 0
    if ((unsigned)gx < (unsigned)X && (unsigned)gy < (unsigned)Y && (unsigned)gz < (unsigned)Z)
      v = src[((size_t)gz * (size_t)Y + (size_t)gy) * (size_t)X + (size_t)gx];
    tile[lz][ly][lx] = v;
  }
  __syncthreads();
  const int lx2 = (tid % (kBfB / 2)) * 2, ly = (tid / (kBfB / 2)) % kBfB, lz = tid / (kBfB * kBfB / 2);
  const int px = bx + lx2, py = by + ly, pz = bz + lz;
  if (px >= X || py >= Y || pz >= Z) return;
  const short mid_a = tile[lz + kBfR][ly + kBfR][lx2 + kBfR], mid_b = tile[lz + kBfR][ly + kBfR][lx2 + kBfR + 1];
  const bf_short2 ma = bf_short2{mid_a, mid_a}, mb = bf_short2{mid_b, mid_b};
  bf_float2 out_colour = bf_float2{0.0f, 0.0f}, wp = bf_float2{0.0f, 0.0f};  // .x: the even voxel, .y: its right neighbour
#pragma unroll
  for (int z = -kBfR; z <= kBfR; ++z)
#pragma unroll
    for (int y = -kBfR; y <= kBfR; ++y) {
      // tile x positions lx2 .. lx2 + 5 of this row: the even voxel's taps are positions 0..4, the odd voxel's 1..5
      const bf_short2 *row = reinterpret_cast<const bf_short2 *>(&tile[lz + kBfR + z][ly + kBfR + y][lx2]);
      const bf_short2 q0 = row[0], q1 = row[1], q2 = row[2];
      const bf_short2 a0 = bf_absdiff16(ma, q0), a1 = bf_absdiff16(ma, q1), a2 = bf_absdiff16(ma, q2);
      const bf_short2 b0 = bf_absdiff16(mb, q0), b1 = bf_absdiff16(mb, q1), b2 = bf_absdiff16(mb, q2);
      const float f[6] = {(float)q0.x, (float)q0.y, (float)q1.x, (float)q1.y, (float)q2.x, (float)q2.y};
      const int da[5] = {a0.x, a0.y, a1.x, a1.y, a2.x}, db[5] = {b0.y, b1.x, b1.y, b2.x, b2.y};
#pragma unroll
      for (int x = -kBfR; x <= kBfR; ++x) {
        const int k = x + kBfR;
        const float *wrow = wtab[x * x + y * y + z * z];
        const bf_float2 w = bf_float2{wrow[da[k]], wrow[db[k]]};
        const bf_float2 local = bf_float2{f[k], f[k + 1]};
        wp = wp + w;
        out_colour = out_colour + local * w;
      }
    }
  {
    const float q = out_colour.x / wp.x;  // wp >= 1: the centre tap weighs exp(0)
    const int32_t r = max(-32768, min(32767, f2i(q)));
    dst[((size_t)pz * (size_t)Y + (size_t)py) * (size_t)X + (size_t)px] = (int16_t)r;
  }
  if (px + 1 < X) {
    const float q = out_colour.y / wp.y;
    const int32_t r = max(-32768, min(32767, f2i(q)));
    dst[((size_t)pz * (size_t)Y + (size_t)py) * (size_t)X + (size_t)px + 1u] = (int16_t)r;
  }
}

// ------------------------------------------------------------------------------------------------
// tf_sort_values  opencl_kernels/histogram.cl:4-32: 2-D histogram over (value, |gradient|) for the
// transfer-function editor.  The reference does one global atomic per voxel; CT data puts most voxels
// into a handful of bins.  A persistent grid walks the volume (a lane eight voxels at a time, merging runs of equal bins)
// and every block counts into a hash table in LDS, flushed to memory once at the end.  (The first version merged equal bins
// across a wave and sent one global atomic per wave and distinct bin: still 94 ms at 512^3 on the few hot bins.)  Bins outside the width x height
// frame (value == max_value rounds to column `width`; values below min_value go negative -- both write
// out of bounds in the reference) are dropped.
// 4096 slots (32 KB): four blocks per CU.  8192 slots left two (0.68 ms at 512^3 against 0.54); 2048: 0.55 ms; 1024: 0.81 ms (bins that find
// no slot go to memory).  Also measured, without gain (profiles/r03_volume_kernels_512.txt): the first probes of a lane's eight updates in
// flight together; both bin coordinates from exact tables instead of sqrt / divide / round (VALU 69 % -> 35 % busy, same time: the walk's
// slice takes 9 us here against 2.3 us in k_fetch_stats_columns, and it is not issue, LDS cycles or same-bin atomics that it waits for).
constexpr int kHistSlotsLog2 = 12, kHistSlots = 1 << kHistSlotsLog2;
__global__ __launch_bounds__(256) void k_tf_sort_values(const int16_t *__restrict__ vol, int X, int Y, int Z, uint32_t *frame,
                                                        int width, int height, float min_value, float max_value,
                                                        float min_gradient, float max_gradient) {
  // the block's private histogram of the bins it meets: open addressing, keys never leave once set, counts by LDS atomics;
  // a bin that finds its probe window taken goes to memory directly
  __shared__ int s_key[kHistSlots];
  __shared__ uint32_t s_cnt[kHistSlots];
  for (int i = (int)threadIdx.x; i < kHistSlots; i += 256) { s_key[i] = -1; s_cnt[i] = 0u; }
  __syncthreads();
  auto add = [&](int bin, uint32_t c) {
    unsigned slot = ((unsigned)bin * 2654435761u) >> (32 - kHistSlotsLog2);
    for (int probe = 0; probe < 4; ++probe) {
      int k = s_key[slot];
      if (k == -1) {
        const int old = atomicCAS(&s_key[slot], -1, bin);
        k = old == -1 ? bin : old;
      }
      if (k == bin) {
        atomicAdd(&s_cnt[slot], c);
        return;
      }
      slot = (slot + 1u) & (unsigned)(kHistSlots - 1);
    }
    atomicAdd(&frame[bin], c);
  };
  const float value_range = max_value - min_value;
  const float gradient_range = max_gradient - min_gradient;
  // work item = eight consecutive voxels of a row; a block takes 256 consecutive items (several rows when rows are short)
  const size_t per_row = ((size_t)X + 7u) / 8u, n_items = (size_t)Y * (size_t)Z * per_row;
  for (size_t item = (size_t)blockIdx.x * 256u + threadIdx.x; item < n_items; item += (size_t)gridDim.x * 256u) {
    const size_t row = item / per_row;
    const int x0 = (int)(item - row * per_row) * 8;
    const int z = (int)(row / (size_t)Y), y = (int)(row - (size_t)z * (size_t)Y);
    Voxels8 o;
    load_voxels8(vol, X, Y, Z, x0, y, z, o);
    int run_bin = -1;
    uint32_t run_count = 0u;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      if (x0 + h >= X) break;
      const int ref_value = o.v[h];
      const float grad_length = gradient_length8(o, h);
      int bin = -1;
      if (!(grad_length > max_gradient) && !((float)ref_value > max_value)) {
        const int px = f2i(roundf((((float)ref_value - min_value) / value_range) * (float)width));
        const int py = f2i(roundf(((grad_length - min_gradient) / gradient_range) * (float)height));
        if (px >= 0 && px < width && py >= 0 && py < height) bin = px * height + py;
      }
      if (bin != run_bin) {  // neighbouring voxels mostly share a bin: one table update per run
        if (run_count) add(run_bin, run_count);
        run_bin = bin;
        run_count = 0u;
      }
      if (bin >= 0) run_count += 1u;
    }
    if (run_count) add(run_bin, run_count);
  }
  __syncthreads();
  for (int i = (int)threadIdx.x; i < kHistSlots; i += 256)
    if (s_key[i] >= 0 && s_cnt[i] != 0u) atomicAdd(&frame[s_key[i]], s_cnt[i]);
}

// the same histogram on the column walk (rows of a multiple of 8 voxels): every voxel fetched about once instead of five times
__global__ __launch_bounds__(256) void k_tf_sort_values_columns(const int16_t *__restrict__ vol, int X, int Y, int Z, uint32_t *frame,
                                                                int width, int height, float min_value, float max_value,
                                                                float min_gradient, float max_gradient) {
  __shared__ ColumnTile tile;
  __shared__ int s_key[kHistSlots];
  __shared__ uint32_t s_cnt[kHistSlots];
  for (int i = (int)threadIdx.x; i < kHistSlots; i += 256) { s_key[i] = -1; s_cnt[i] = 0u; }
  __syncthreads();
  auto add = [&](int bin, uint32_t c) {
    unsigned slot = ((unsigned)bin * 2654435761u) >> (32 - kHistSlotsLog2);
    for (int probe = 0; probe < 4; ++probe) {
      int k = s_key[slot];
      if (k == -1) {
        const int old = atomicCAS(&s_key[slot], -1, bin);
        k = old == -1 ? bin : old;
      }
      if (k == bin) {
        atomicAdd(&s_cnt[slot], c);
        return;
      }
      slot = (slot + 1u) & (unsigned)(kHistSlots - 1);
    }
    atomicAdd(&frame[bin], c);
  };
  const float value_range = max_value - min_value;
  const float gradient_range = max_gradient - min_gradient;
  int ntx, nty;
  const size_t n_tiles = column_tiles(X, Y, Z, ntx, nty);
  for (size_t ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
    const int tx = (int)(ti % (size_t)ntx), ty = (int)((ti / (size_t)ntx) % (size_t)nty), tz = (int)(ti / ((size_t)ntx * (size_t)nty));
    walk_column_tile(vol, X, Y, Z, tx, ty, tz, tile, [&](int, int, int, const int (&v)[8], const int (&gx)[8], const int (&gy)[8], const int (&gz)[8]) {
      int run_bin = -1;
      uint32_t run_count = 0u;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int ref_value = v[h];
        const float fx = (float)gx[h], fy = (float)gy[h], fz = (float)gz[h];
        const float grad_length = sqrtf((fx * fx + fy * fy) + fz * fz);
        int bin = -1;
        if (!(grad_length > max_gradient) && !((float)ref_value > max_value)) {
          const int px = f2i(roundf((((float)ref_value - min_value) / value_range) * (float)width));
          const int py = f2i(roundf(((grad_length - min_gradient) / gradient_range) * (float)height));
          if (px >= 0 && px < width && py >= 0 && py < height) bin = px * height + py;
        }
        if (bin != run_bin) {  // neighbouring voxels mostly share a bin: one table update per run
          if (run_count) add(run_bin, run_count);
          run_bin = bin;
          run_count = 0u;
        }
        if (bin >= 0) run_count += 1u;
      }
      if (run_count) add(run_bin, run_count);
    });
  }
  __syncthreads();
  for (int i = (int)threadIdx.x; i < kHistSlots; i += 256)
    if (s_key[i] >= 0 && s_cnt[i] != 0u) atomicAdd(&frame[s_key[i]], s_cnt[i]);
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
  hipLaunchKernelGGL(k_bilateral_filter2, grid, dim3(kBfB * kBfB * kBfB / 2), 0, s, src, X, Y, Z, dst, weights);
  return hipGetLastError();
}

static unsigned persistent_grid(int X, int Y, int Z) {
  const size_t n_items = (size_t)Y * (size_t)Z * (((size_t)X + 7u) / 8u);
  return (unsigned)std::min<size_t>((n_items + 255u) / 256u, 2048u);
}

static unsigned column_grid(int X, int Y, int Z) {
  const size_t n_tiles = (size_t)((X + kColX - 1) / kColX) * (size_t)((Y + kColY - 1) / kColY) * (size_t)((Z + kColZ - 1) / kColZ);
  return (unsigned)std::min<size_t>(n_tiles, 2048u);
}

hipError_t launch_fetch_stats(const int16_t *vol, int X, int Y, int Z, int32_t *stats, hipStream_t s) {
  if ((X & 7) == 0 && (reinterpret_cast<uintptr_t>(vol) & 15u) == 0u) {
    hipLaunchKernelGGL(k_fetch_stats_columns, dim3(column_grid(X, Y, Z)), dim3(256), 0, s, vol, X, Y, Z, stats);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_fetch_stats, dim3(persistent_grid(X, Y, Z)), dim3(256), 0, s, vol, X, Y, Z, stats);
  return hipGetLastError();
}

hipError_t launch_apply_clip(const int16_t *src, int SX, int SY, int SZ, int16_t *dst, int DX, int DY, int DZ,
                             const uint32_t *start, const uint32_t *len, hipStream_t s) {
  if ((DX & 7) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0u && DX > 0 && DY > 0 && DZ > 0 && (size_t)DY * (size_t)(DX >> 3) < (1ull << 31)) {
    // (one launch of single-voxel lanes per row was 295 k blocks at 512^3 -> 384^3: 0.157 ms, bound by the block rate)
    const unsigned items = (unsigned)DY * (unsigned)(DX >> 3);
    hipLaunchKernelGGL(k_apply_clip8, dim3((items + 255u) / 256u, (unsigned)DZ), dim3(256), 0, s, src, SX, SY, SZ, dst, DX, DY, DZ, start, len);
    return hipGetLastError();
  }
  const unsigned b = row_block(DX);
  hipLaunchKernelGGL(k_apply_clip, dim3(((unsigned)DX + b - 1u) / b, (unsigned)DY, (unsigned)DZ), dim3(b), 0, s, src, SX, SY,
                     SZ, dst, DX, DY, DZ, start, len);
  return hipGetLastError();
}

hipError_t launch_tf_sort_values(const int16_t *vol, int X, int Y, int Z, uint32_t *frame, int width, int height,
                                 float min_v, float max_v, float min_g, float max_g, hipStream_t s) {
  if ((X & 7) == 0 && (reinterpret_cast<uintptr_t>(vol) & 15u) == 0u) {
    // four blocks fit a CU (the hash table): 1024 are resident, and every block flushes its table once -- with 2048 blocks the launch
    // flushed twice as many tables into the same hot bins (0.54 ms against 0.49); 512 / 256 blocks: 0.66 / 1.15 ms
    hipLaunchKernelGGL(k_tf_sort_values_columns, dim3(std::min(column_grid(X, Y, Z), 1024u)), dim3(256), 0, s, vol, X, Y, Z, frame, width, height, min_v, max_v,
                       min_g, max_g);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_tf_sort_values, dim3(persistent_grid(X, Y, Z)), dim3(256), 0, s, vol, X, Y, Z, frame, width, height, min_v, max_v,
                     min_g, max_g);
  return hipGetLastError();
}

hipError_t launch_tf_flush_color_frame(uint32_t *color_frame, int fw, int fh, const int32_t *frame, const int32_t *lookup,
                                       int lookup_len, hipStream_t s) {
  hipLaunchKernelGGL(k_tf_flush_color_frame, dim3((unsigned)(fw + 15) / 16u, (unsigned)(fh + 15) / 16u), dim3(256), 0, s,
                     color_frame, fw, fh, frame, lookup, lookup_len);
  return hipGetLastError();
}

}  // namespace clvr
