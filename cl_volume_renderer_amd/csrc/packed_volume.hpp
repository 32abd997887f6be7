// packed_volume.hpp -- the MI355X-native resident form of (volume, SDF, transfer function).
//
// The reference samples two images per march step: the SDF at trunc(origin) before the step and the
// volume at floor(origin) after it (utility_ray.cl:148-154, :126-138).  Inside the volume both
// addresses of consecutive steps coincide, so the shim keeps ONE 4-byte record per voxel
//     bits  0..15  value   (int16, the volume texel)
//     bits 16..23  sdf     (int8, the distance-field texel)
//     bits 24..31  class   (1 + index of the first transfer-function rule the voxel satisfies, 0 = no
//                           event; rules that read `gradient` use the gradient at the voxel's integer
//                           position, see classify_step in render_device.hpp)
// and one gather per step returns the classification of the new position AND the step length of the
// next step.  The march needs even less: a second array keeps ONE BYTE
// per voxel, bit 7 = "class != 0" (a Hit), bits 0-6 = max(sdf, 0) (the step is max(sdf, 0.5), so
// negative distances all mean 0.5); the 4-byte record is only fetched at a Hit (colour) and for the
// 6-tap normal.  Records are stored in 8x8x8 bricks made of eight 4x4x4 sub-bricks (256 B each, x
// fastest inside), so the 64 rays of an 8x8 pixel tile, which walk a tube a few voxels wide, share
// a handful of 128-B lines instead of one line per (y,z) row as in the caller's x-fastest layout.
// The packed volume is derived data: built by k_repack at the first render after the volume, the
// SDF or the transfer function changed.
#pragma once

#include "device_math.hpp"

namespace clvr {

struct VolumePacked {
  const uint32_t *__restrict__ rec;
  const uint8_t *__restrict__ stepb;  // 1 byte per voxel, same brick order: bit7 = class != 0, bits0-6 = max(sdf, 0)
  int X, Y, Z;
  int NBX, NBY;  // bricks per row / per slice

  // offset of a voxel inside its 8x8x8 brick
  __host__ __device__ static inline unsigned inner_index(unsigned ux, unsigned uy, unsigned uz) {
#ifdef CLVR_BRICK_SLAB  // experiment: plain z-y-x order, a 64-byte line of step bytes is an 8x8x1 slab (5 ALU ops instead of 15)
    return ((uz & 7u) << 6) | ((uy & 7u) << 3) | (ux & 7u);
#else
    return ((uz & 4u) << 6) | ((uy & 4u) << 5) | ((ux & 4u) << 4) | ((uz & 3u) << 4) | ((uy & 3u) << 2) | (ux & 3u);
#endif
  }
  // inverse of inner_index
  __host__ __device__ static inline void inner_coords(unsigned inner, unsigned &x, unsigned &y, unsigned &z) {
#ifdef CLVR_BRICK_SLAB
    x = inner & 7u; y = (inner >> 3) & 7u; z = inner >> 6;
#else
    x = (inner & 3u) | ((inner >> 4) & 4u); y = ((inner >> 2) & 3u) | ((inner >> 5) & 4u); z = ((inner >> 4) & 3u) | ((inner >> 6) & 4u);
#endif
  }

  // The record index is separable: index(x, y, z) = part_x(x) + part_y(y) + part_z(z) -- the brick number is a sum
  // of per-axis terms and the in-brick bit fields of the three axes are disjoint.  The 6-tap normal needs nine
  // part evaluations instead of six full index computations.  SMALL (fewer than 2^23 bricks): parts are 32-bit and
  // use full-rate 24-bit multiply-adds (the compiler turns __umul24 back into quarter-rate 32/64-bit multiplies).
  template <bool SMALL> struct Index { using type = size_t; };
  __device__ __forceinline__ static unsigned mul24_uniform(unsigned v, int uniform) {
    unsigned r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(v), "s"(uniform));
    return r;
  }
  template <bool SMALL>
  __device__ __forceinline__ typename Index<SMALL>::type part_x(unsigned u) const {
    return ((typename Index<SMALL>::type)(u >> 3) << 9) | inner_index(u, 0u, 0u);
  }
  template <bool SMALL>
  __device__ __forceinline__ typename Index<SMALL>::type part_y(unsigned u) const {
    const unsigned b = SMALL ? mul24_uniform(u >> 3, NBX) : (u >> 3) * (unsigned)NBX;
    return ((typename Index<SMALL>::type)b << 9) | inner_index(0u, u, 0u);
  }
  template <bool SMALL>
  __device__ __forceinline__ typename Index<SMALL>::type part_z(unsigned u) const {
    const unsigned b = SMALL ? mul24_uniform(u >> 3, NBX * NBY) : (u >> 3) * (unsigned)(NBX * NBY);
    return ((typename Index<SMALL>::type)b << 9) | inner_index(0u, 0u, u);
  }

  // coordinates are non-negative and in range; the brick number fits 32 bits for every volume that
  // fits the GPU (2048^3 has 2^24 bricks), so only the final scale by 512 records is 64-bit
  __host__ __device__ static inline size_t record_index(int x, int y, int z, int nbx, int nby) {
    const unsigned ux = (unsigned)x, uy = (unsigned)y, uz = (unsigned)z;
    const unsigned brick = ((uz >> 3) * (unsigned)nby + (uy >> 3)) * (unsigned)nbx + (ux >> 3);
    return ((size_t)brick << 9) + inner_index(ux, uy, uz);
  }

  // int coordinates (read_imagei(img, int4)): out of range -> border record 0
  __device__ __forceinline__ uint32_t fetch_i(int x, int y, int z) const {
    if ((unsigned)x >= (unsigned)X || (unsigned)y >= (unsigned)Y || (unsigned)z >= (unsigned)Z) return 0u;
    return rec[record_index(x, y, z, NBX, NBY)];
  }
  // float coordinates (read_imagei(img, smp, float4)): texel = floor(coord); out of range / NaN -> 0
  __device__ __forceinline__ uint32_t fetch_f(float fx, float fy, float fz) const {
    const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
    if (!(gx >= 0.0f && gy >= 0.0f && gz >= 0.0f && gx < (float)X && gy < (float)Y && gz < (float)Z)) return 0u;
    return rec[record_index((int)gx, (int)gy, (int)gz, NBX, NBY)];
  }

  // the march's per-step byte (gradient-free transfer functions): one 64-byte line holds a whole 4x4x4
  // sub-brick, the whole 512^3 array is 128 MiB and stays resident in the 256 MiB Infinity Cache
  template <bool SMALL = false>
  __device__ __forceinline__ unsigned step_i(int x, int y, int z) const {
    if ((unsigned)x >= (unsigned)X || (unsigned)y >= (unsigned)Y || (unsigned)z >= (unsigned)Z) return 0u;
    return stepb[part_x<SMALL>((unsigned)x) + part_y<SMALL>((unsigned)y) + part_z<SMALL>((unsigned)z)];
  }
  // The march's fetch after `!exited_volume(pos)`: every coordinate is then >= 0 (or -0.0), <= its dimension, or
  // NaN.  trunc == floor for such values, so only `coordinate < dimension` remains to be tested (false for NaN
  // and for coordinate == dimension, which both read the border), and the in-brick offset is formed in 32 bits
  // so that the load can use scalar-base + 32-bit-offset addressing.  SMALL: the volume has fewer than 2^23
  // bricks, every step byte has a 32-bit offset and the brick number is formed with 24-bit multiplies.
  template <bool SMALL>
  __device__ __forceinline__ unsigned step_marched(float fx, float fy, float fz) const {
    if (!(fx < (float)X && fy < (float)Y && fz < (float)Z)) return 0u;
    const unsigned ux = (unsigned)(int)fx, uy = (unsigned)(int)fy, uz = (unsigned)(int)fz;
    return stepb[part_x<SMALL>(ux) + part_y<SMALL>(uy) + part_z<SMALL>(uz)];
  }
  __device__ __forceinline__ unsigned step_f(float fx, float fy, float fz) const {
    const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
    if (!(gx >= 0.0f && gy >= 0.0f && gz >= 0.0f && gx < (float)X && gy < (float)Y && gz < (float)Z)) return 0u;
    return stepb[record_index((int)gx, (int)gy, (int)gz, NBX, NBY)];
  }

  // branch-free form of fetch_f for batches (the 6-tap normal): the load is always issued, from record 0
  // when the texel is outside, and the result is masked afterwards -- so several taps can be in flight
  // before the first wait instead of one full memory latency per tap
  __device__ __forceinline__ uint32_t fetch_f_masked(float fx, float fy, float fz) const {
    const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
    const bool ok = gx >= 0.0f && gy >= 0.0f && gz >= 0.0f && gx < (float)X && gy < (float)Y && gz < (float)Z;
    const size_t idx = record_index((int)(ok ? gx : 0.0f), (int)(ok ? gy : 0.0f), (int)(ok ? gz : 0.0f), NBX, NBY);
    const uint32_t r = rec[idx];
    return ok ? r : 0u;
  }

  __device__ __forceinline__ static int value_of(uint32_t r) { return (int)(int16_t)(r & 0xFFFFu); }
  __device__ __forceinline__ static int sdf_of(uint32_t r) { return (int)(int8_t)((r >> 16) & 0xFFu); }
  __device__ __forceinline__ static unsigned class_of(uint32_t r) { return r >> 24; }

  __device__ __forceinline__ int value_at(float fx, float fy, float fz) const { return value_of(fetch_f(fx, fy, fz)); }
  __device__ __forceinline__ int sdf_at(int x, int y, int z) const { return sdf_of(fetch_i(x, y, z)); }
};
template <> struct VolumePacked::Index<true> { using type = uint32_t; };

}  // namespace clvr
