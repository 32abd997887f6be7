// packed_volume.hpp -- the MI355X-native resident form of (volume, SDF, transfer function).
//
// The reference samples two images per march step: the SDF at trunc(origin) before the step and the
// volume at floor(origin) after it (utility_ray.cl:148-154, :126-138); at a Hit it reads six more volume
// texels for the normal (utility_filter.cl:2-35).  On this GPU every one of those gathers that misses L2
// moves a whole 128-byte line (profiles/r02_miss_bytes_probe.txt), and the kernel's time IS its count of
// missing lines, so the resident form is built to touch as few lines as possible:
//
//   step bytes   ONE BYTE per voxel: bit 7 = "class != 0" (a Hit), bits 0-6 = max(sdf, 0) (the step is
//                max(sdf, 0.5), so negative distances all mean 0.5).  Inside the volume the SDF address of
//                step k+1 and the volume address of step k coincide, so one 1-byte gather per step returns
//                the classification of the new position AND the length of the next step.
//   hit records  8 bytes per voxel, read only at a Hit: the voxel's central differences gx, gy, gz (17 bits
//                each: differences of two int16 with border 0) and its class (1 + index of the first
//                transfer-function rule it satisfies; rules that read `gradient` use the gradient at the
//                voxel's integer position, see classify_step in render_device.hpp).  One 8-byte load -- one
//                line -- replaces the colour fetch plus six taps spread over two or three lines.  The taps of
//                the reference are at floor(p +- 1), which are the voxel's own neighbours except when an
//                addition rounds across an integer; those positions (detected exactly) and the literal
//                7-fetch test mode read the caller's x-fastest images instead.
//
// Both arrays are stored in 8x8x8 bricks made of eight 4x4x4 sub-bricks (x fastest inside), so the rays of an
// 8x8 pixel tile, which walk a tube a few voxels wide, share lines instead of touching one line per (y,z) row.
// They are derived data: built by k_repack at the first render after the volume, the SDF or the transfer
// function changed.
#pragma once

#include "device_math.hpp"

namespace clvr {

struct VolumePacked {
  const uint2 *__restrict__ grec;     // hit records, brick order (see pack_hit / unpack below)
  const uint8_t *__restrict__ stepb;  // 1 byte per voxel, same brick order: bit7 = class != 0, bits0-6 = max(sdf, 0)
  const int16_t *__restrict__ vol_lin;  // the caller's images, x fastest: literal taps only (rare paths)
  const int8_t *__restrict__ sdf_lin;
  int X, Y, Z;
  int NBX, NBY;  // bricks per row / per slice
  // SMALL == 2 only: the three per-axis index terms as tables in LDS (part_x below), y's and z's after x's
  const uint32_t *parts = nullptr;
  int parts_y0 = 0, parts_z0 = 0;

  // offset of a voxel inside its 8x8x8 brick
  __host__ __device__ static inline unsigned inner_index(unsigned ux, unsigned uy, unsigned uz) {
    return ((uz & 4u) << 6) | ((uy & 4u) << 5) | ((ux & 4u) << 4) | ((uz & 3u) << 4) | ((uy & 3u) << 2) | (ux & 3u);
  }
  // inverse of inner_index
  __host__ __device__ static inline void inner_coords(unsigned inner, unsigned &x, unsigned &y, unsigned &z) {
    x = (inner & 3u) | ((inner >> 4) & 4u); y = ((inner >> 2) & 3u) | ((inner >> 5) & 4u); z = ((inner >> 4) & 3u) | ((inner >> 6) & 4u);
  }

  // hit record: dword0 = gx[16:0] | gy[14:0] << 17, dword1 = gy[16:15] | gz[16:0] << 2 | class << 19
  __host__ __device__ static inline uint2 pack_hit(int gx, int gy, int gz, unsigned cls) {
    const uint32_t ux = (uint32_t)gx & 0x1FFFFu, uy = (uint32_t)gy & 0x1FFFFu, uz = (uint32_t)gz & 0x1FFFFu;
    uint2 r;
    r.x = ux | (uy << 17);
    r.y = (uy >> 15) | (uz << 2) | ((cls & 0xFFu) << 19);
    return r;
  }
  __device__ __forceinline__ static unsigned hit_class(uint2 r) { return (r.y >> 19) & 0xFFu; }
  __device__ __forceinline__ static void hit_gradient(uint2 r, int &gx, int &gy, int &gz) {
    gx = ((int)(r.x << 15)) >> 15;
    gy = ((int)(((r.y & 3u) << 30) | ((r.x >> 17) << 15))) >> 15;
    gz = ((int)(r.y << 13)) >> 15;
  }

  // The brick index is separable: index(x, y, z) = part_x(x) + part_y(y) + part_z(z) -- the brick number is a sum
  // of per-axis terms and the in-brick bit fields of the three axes are disjoint.  SMALL != 0 (fewer than 2^23
  // bricks): parts are 32-bit and use full-rate 24-bit multiply-adds (the compiler turns __umul24 back into
  // quarter-rate 32/64-bit multiplies).  SMALL == 2: the parts come from tables in LDS -- three ds_read_b32 and
  // their three address instructions replace some twenty VALU instructions per index, in a kernel that is bound by
  // VALU issue (k_bounce fills the tables; kPartsMaxEntries bounds X + Y + Z).
  static constexpr int kPartsMaxEntries = 6144;
  template <int SMALL> struct Index { using type = size_t; };
  __device__ __forceinline__ static unsigned mul24_uniform(unsigned v, int uniform) {
    unsigned r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(v), "s"(uniform));
    return r;
  }
  template <int SMALL>
  __device__ __forceinline__ typename Index<SMALL>::type part_x(unsigned u) const {
    if constexpr (SMALL == 2) return parts[u];
    else return ((typename Index<SMALL>::type)(u >> 3) << 9) | inner_index(u, 0u, 0u);
  }
  template <int SMALL>
  __device__ __forceinline__ typename Index<SMALL>::type part_y(unsigned u) const {
    if constexpr (SMALL == 2) return parts[(unsigned)parts_y0 + u];
    else {
      const unsigned b = SMALL ? mul24_uniform(u >> 3, NBX) : (u >> 3) * (unsigned)NBX;
      return ((typename Index<SMALL>::type)b << 9) | inner_index(0u, u, 0u);
    }
  }
  template <int SMALL>
  __device__ __forceinline__ typename Index<SMALL>::type part_z(unsigned u) const {
    if constexpr (SMALL == 2) return parts[(unsigned)parts_z0 + u];
    else {
      const unsigned b = SMALL ? mul24_uniform(u >> 3, NBX * NBY) : (u >> 3) * (unsigned)(NBX * NBY);
      return ((typename Index<SMALL>::type)b << 9) | inner_index(0u, 0u, u);
    }
  }

  // coordinates are non-negative and in range; the brick number fits 32 bits for every volume that
  // fits the GPU (2048^3 has 2^24 bricks), so only the final scale by 512 voxels is 64-bit
  __host__ __device__ static inline size_t record_index(int x, int y, int z, int nbx, int nby) {
    const unsigned ux = (unsigned)x, uy = (unsigned)y, uz = (unsigned)z;
    const unsigned brick = ((uz >> 3) * (unsigned)nby + (uy >> 3)) * (unsigned)nbx + (ux >> 3);
    return ((size_t)brick << 9) + inner_index(ux, uy, uz);
  }

  // ---- the caller's own images (literal taps: positions whose +-1 taps are not the voxel's neighbours, the
  // literal test mode, and every gather of a transfer function evaluated per step with its six taps)
  // read_imagei(volume, smp, float4): texel = floor(coord); out of range / NaN -> border 0
  __device__ __forceinline__ int value_at(float fx, float fy, float fz) const {
    const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
    if (!(gx >= 0.0f && gy >= 0.0f && gz >= 0.0f && gx < (float)X && gy < (float)Y && gz < (float)Z)) return 0;
    return vol_lin[((size_t)(int)gz * (size_t)Y + (size_t)(int)gy) * (size_t)X + (size_t)(int)gx];
  }
  __device__ __forceinline__ int sdf_at_f(float fx, float fy, float fz) const {
    const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
    if (!(gx >= 0.0f && gy >= 0.0f && gz >= 0.0f && gx < (float)X && gy < (float)Y && gz < (float)Z)) return 0;
    return sdf_lin[((size_t)(int)gz * (size_t)Y + (size_t)(int)gy) * (size_t)X + (size_t)(int)gx];
  }

  // the hit record of the voxel at floor(pos); the caller guarantees 0 <= coordinate < dimension
  template <int SMALL = 0>
  __device__ __forceinline__ uint2 hit_record(float fx, float fy, float fz) const {
    const unsigned ux = (unsigned)(int)fx, uy = (unsigned)(int)fy, uz = (unsigned)(int)fz;
    return grec[part_x<SMALL>(ux) + part_y<SMALL>(uy) + part_z<SMALL>(uz)];
  }

  // the march's per-step byte: one 128-byte line holds two 4x4x4 sub-bricks, the whole 512^3 array is 128 MiB
  template <int SMALL = 0>
  __device__ __forceinline__ unsigned step_i(int x, int y, int z) const {
    if ((unsigned)x >= (unsigned)X || (unsigned)y >= (unsigned)Y || (unsigned)z >= (unsigned)Z) return 0u;
    return stepb[part_x<SMALL>((unsigned)x) + part_y<SMALL>((unsigned)y) + part_z<SMALL>((unsigned)z)];
  }
  // The march's fetch at a position that HAS a voxel: classify_step has tested `coordinate < dimension` for all three
  // (false for NaN and for coordinate == dimension, which read the border texel instead) after `!exited_volume(pos)`,
  // so every coordinate is in [0, dimension) or -0.0 and trunc == floor.  The in-brick offset is formed in 32 bits so
  // that the load can use scalar-base + 32-bit-offset addressing.  SMALL: the volume has fewer than 2^23 bricks, every
  // step byte has a 32-bit offset and the brick number is formed with 24-bit multiplies.
  template <int SMALL>
  __device__ __forceinline__ unsigned step_marched(float fx, float fy, float fz) const {
    const unsigned ux = (unsigned)(int)fx, uy = (unsigned)(int)fy, uz = (unsigned)(int)fz;
    return stepb[part_x<SMALL>(ux) + part_y<SMALL>(uy) + part_z<SMALL>(uz)];
  }
  // the same fetch as an agent-scope load (`sc1`): served by L2, leaves no line in the CU's L1
  template <int SMALL>
  __device__ __forceinline__ unsigned step_marched_past_l1(float fx, float fy, float fz) const {
    const unsigned ux = (unsigned)(int)fx, uy = (unsigned)(int)fy, uz = (unsigned)(int)fz;
    return __hip_atomic_load(stepb + (part_x<SMALL>(ux) + part_y<SMALL>(uy) + part_z<SMALL>(uz)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};
template <> struct VolumePacked::Index<1> { using type = uint32_t; };
template <> struct VolumePacked::Index<2> { using type = uint32_t; };

}  // namespace clvr
