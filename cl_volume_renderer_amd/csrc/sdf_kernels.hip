// sdf_kernels.hip -- signed-distance-field builder for the empty-space skip, gfx950.
//
// What is computed follows opencl_kernels/signed_distance_field.cl (:6-54 create_base_image,
// :56-87 neightbour_distance_calc, :89-112 create_signed_distance_field) and the host loop of
// app/signed_distance_field.cpp:7-35.  v0: one thread per voxel on x-fastest rows (coalesced byte
// loads/stores along x), per-wave aggregation of the progress counter (one atomic per wave instead
// of one per voxel), and a device-side early-out chain so the fused build needs no host round trip
// per layer.
#include "clwh_internal.hpp"
#include "device_math.hpp"
#include "render_device.hpp"

namespace clvr {

struct VolumeIntLinear {
  const int16_t *__restrict__ vol;
  int X, Y, Z;
  // read_imagei(volume, int4): out of range -> border 0
  __device__ __forceinline__ int at(int x, int y, int z) const {
    if ((unsigned)x >= (unsigned)X || (unsigned)y >= (unsigned)Y || (unsigned)z >= (unsigned)Z) return 0;
    return vol[((size_t)z * (size_t)Y + (size_t)y) * (size_t)X + (size_t)x];
  }
};

template <bool USE_GRAD>
__device__ __forceinline__ bool event_at(const VolumeIntLinear &v, const TfDev &tf, const uint8_t *cls_in, int x, int y, int z) {
  if (cls_in) return cls_in[((size_t)z * (size_t)v.Y + (size_t)y) * (size_t)v.X + (size_t)x] != 0;  // opaque TF (tf_jit.cpp)
  const int value = v.at(x, y, z);
  int gradient = 0;
  if (USE_GRAD) {
    const float dx = (float)(v.at(x + 1, y, z) - v.at(x - 1, y, z));
    const float dy = (float)(v.at(x, y + 1, z) - v.at(x, y - 1, z));
    const float dz = (float)(v.at(x, y, z + 1) - v.at(x, y, z - 1));
    gradient = (int)(short)f2i(sqrtf((dx * dx + dy * dy) + dz * dz));
  }
  uint32_t color = 0u;
  return tf_eval(tf, value, gradient, color);
}

// create_base_image: -1 inside an event region, +1 outside; times max_iterations where the 8 clamped
// CORNER neighbours agree with the centre
template <bool USE_GRAD>
__global__ __launch_bounds__(256) void k_sdf_base(const SdfArgs a) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int z = blockIdx.z;
  if (x >= a.X) return;
  const VolumeIntLinear v{a.volume, a.X, a.Y, a.Z};
  const bool e = event_at<USE_GRAD>(v, a.tf, a.cls_in, x, y, z);
  bool homogenous = true;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int nx = min(max(x + ((c & 1) ? 1 : -1), 0), a.X - 1);
    const int ny = min(max(y + ((c & 2) ? 1 : -1), 0), a.Y - 1);
    const int nz = min(max(z + ((c & 4) ? 1 : -1), 0), a.Z - 1);
    homogenous &= (event_at<USE_GRAD>(v, a.tf, a.cls_in, nx, ny, nz) == e);
  }
  int r = e ? -1 : 1;
  if (homogenous) r *= a.max_iterations;
  const size_t i = ((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)a.X + (size_t)x;
  a.ping[i] = (int8_t)r;
  a.pong[i] = (int8_t)r;
}

// one propagation layer
__global__ __launch_bounds__(256) void k_sdf_layer(const SdfArgs a) {
  const int it = a.iteration;
  if (a.done) {
    // fused build: layer `it` runs only while the reference's host loop would still be running:
    // it stops after the first ODD layer whose counter stayed 0 (signed_distance_field.cpp:29-31)
    const bool stop = it > 1 && (a.done[it - 1] != 0 || (((it - 1) & 1) && a.counters[it - 1] == 0));
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) a.done[it] = stop ? 1 : 0;
    if (stop) return;
  }
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int z = blockIdx.z;
  bool wrote = false;
  if (x < a.X) {
    const int8_t *__restrict__ in = a.ping;
    const size_t row = ((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)a.X;
    int local_value = in[row + x];
    int abs_value = abs(local_value);
    if (abs_value >= it) {
      if (abs_value > it) {
        int neighbour_distance = 127, abs_added = 0, added = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int nx = min(max(x + ((c & 1) ? 1 : -1), 0), a.X - 1);
          const int ny = min(max(y + ((c & 2) ? 1 : -1), 0), a.Y - 1);
          const int nz = min(max(z + ((c & 4) ? 1 : -1), 0), a.Z - 1);
          const int nv = in[((size_t)nz * (size_t)a.Y + (size_t)ny) * (size_t)a.X + (size_t)nx];
          const int t = (int)(int8_t)abs(nv);
          abs_added += t;
          added += nv;
          neighbour_distance = min(neighbour_distance, t);
        }
        if (abs(added) != abs_added) neighbour_distance = 0;
        if (neighbour_distance != 0 && neighbour_distance == it) {
          abs_value = it + 1;
          local_value = local_value < 0 ? -abs_value : abs_value;
        }
      }
      if (abs_value < a.max_iterations) {
        a.pong[row + x] = (int8_t)local_value;
        wrote = true;
      }
    }
  }
  // atomic_inc(add_buffer) per written voxel -> one add per wave
  const unsigned long long m = __ballot(wrote);
  if (m != 0ull && (threadIdx.x & 63u) == (unsigned)__ffsll((long long)m) - 1u) {
    const int n = __popcll(m);
    if (a.counter_out) atomicAdd(a.counter_out, n);
    if (a.counters) atomicAdd(a.counters + it, n);
  }
}

// ------------------------------------------------------------------------------------------------
// Fused build (clwh_sdf_build): the layer iteration is a breadth-first distance transform over the
// "8 corner neighbours" graph -- a homogeneous voxel settles at layer i to +-(i+1) exactly when a corner
// neighbour holds +-i (signed_distance_field.cl:56-112; its neighbourhood is sign-uniform by
// construction of the base image).  Only voxels next to the layer-i front can change at layer i, so
// the fused build works IN PLACE on the caller's SDF image and visits only the 8x8x8 tiles whose
// 27-neighbourhood changed in the previous layer.  Reading a neighbour that another wave settles
// concurrently is harmless: it moves from +-max to +-(i+1), both > i, and signs never change, so the
// "min |neighbour| == i" test sees the same answer either way.  The result is the fixed point the
// reference's ping/pong loop converges to (both of its buffers hold every settled voxel).
// base image for the fused build: same values as k_sdf_base into ONE buffer, plus the layer-1 tile flags
// A block classifies a 32 x 8 x 4 box of voxels: the event flag of every voxel of the box and of its one-voxel
// halo (clamped to the volume, signed_distance_field.cl:17-31) is evaluated ONCE into LDS -- 2040 evaluations for
// 1024 voxels instead of nine per voxel -- and the homogeneity test reads the eight corner flags from there.
constexpr int kBaseX = 32, kBaseY = 8, kBaseZ = 4;
constexpr int kBaseRX = kBaseX + 2, kBaseRY = kBaseY + 2, kBaseRZ = kBaseZ + 2;
template <bool USE_GRAD>
__global__ __launch_bounds__(256) void k_sdf_base_front(const SdfArgs a, uint8_t *flags, int32_t TX, int32_t TY) {
  __shared__ uint8_t ev[kBaseRZ][kBaseRY][kBaseRX];
  __shared__ int any_one;
  const int x0 = blockIdx.x * kBaseX, y0 = blockIdx.y * kBaseY, z0 = blockIdx.z * kBaseZ;
  const VolumeIntLinear v{a.volume, a.X, a.Y, a.Z};
  if (threadIdx.x == 0) any_one = 0;
  for (int i = threadIdx.x; i < kBaseRX * kBaseRY * kBaseRZ; i += 256) {
    const int rx = i % kBaseRX, ry = (i / kBaseRX) % kBaseRY, rz = i / (kBaseRX * kBaseRY);
    const int gx = min(max(x0 - 1 + rx, 0), a.X - 1), gy = min(max(y0 - 1 + ry, 0), a.Y - 1), gz = min(max(z0 - 1 + rz, 0), a.Z - 1);
    ev[rz][ry][rx] = event_at<USE_GRAD>(v, a.tf, a.cls_in, gx, gy, gz) ? 1 : 0;
  }
  __syncthreads();
  bool block_has_one = false;
  for (int i = threadIdx.x; i < kBaseX * kBaseY * kBaseZ; i += 256) {
    const int lx = i % kBaseX, ly = (i / kBaseX) % kBaseY, lz = i / (kBaseX * kBaseY);
    const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
    if (x >= a.X || y >= a.Y || z >= a.Z) continue;
    const unsigned e = ev[lz + 1][ly + 1][lx + 1];
    bool homogenous = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      // the halo cell of a voxel on the volume's face holds the clamped neighbour's flag
      const int nx = min(max(x + ((c & 1) ? 1 : -1), 0), a.X - 1) - x0 + 1;
      const int ny = min(max(y + ((c & 2) ? 1 : -1), 0), a.Y - 1) - y0 + 1;
      const int nz = min(max(z + ((c & 4) ? 1 : -1), 0), a.Z - 1) - z0 + 1;
      homogenous &= (ev[nz][ny][nx] == e);
    }
    int r = e ? -1 : 1;
    if (homogenous) r *= a.max_iterations;
    a.ping[((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)a.X + (size_t)x] = (int8_t)r;
    if (r == 1 || r == -1) {
      // layer 1 must visit every tile that holds a corner neighbour of a |v| == 1 voxel (idempotent byte stores)
      block_has_one = true;
      const size_t own = ((size_t)(z >> 3) * TY + (size_t)(y >> 3)) * TX + (size_t)(x >> 3);
      flags[own] = 1;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int nx = min(max(x + ((c & 1) ? 1 : -1), 0), a.X - 1);
        const int ny = min(max(y + ((c & 2) ? 1 : -1), 0), a.Y - 1);
        const int nz = min(max(z + ((c & 4) ? 1 : -1), 0), a.Z - 1);
        const size_t t = ((size_t)(nz >> 3) * TY + (size_t)(ny >> 3)) * TX + (size_t)(nx >> 3);
        if (t != own) flags[t] = 1;
      }
    }
  }
  // only zero / non-zero of the counts is ever needed (loop termination), so plain stores replace the
  // same-address atomics that would serialise at one L2 channel
  if (block_has_one) any_one = 1;
  __syncthreads();
  if (threadIdx.x == 0 && any_one) a.counters[0] = 1;
}

#ifndef CLVR_SDF_TILES_PER_BLOCK
#define CLVR_SDF_TILES_PER_BLOCK 32  // measured 16 / 32 / 64 / 256 consecutive tiles per block: 6.96 / 6.61 / 8.26 / 14.3 ms for the 512^3 build
#endif
constexpr unsigned kFrontTilesPerBlock = CLVR_SDF_TILES_PER_BLOCK;  // one wave tests the block's tiles, four waves process the active ones
constexpr int kRowStride = 16, kSliceStride = 160;  // LDS image of a tile + halo: rows of 16 bytes [x0-4, x0+12)

#ifndef CLVR_SDF_WAVES
#define CLVR_SDF_WAVES 4
#endif
constexpr unsigned kFrontWaves = CLVR_SDF_WAVES;  // waves per block sharing the block's list of active tiles
__global__ __launch_bounds__(64 * CLVR_SDF_WAVES) void k_sdf_front(const SdfFrontArgs a) {
  __shared__ uint32_t s_list[kFrontTilesPerBlock];
  __shared__ uint32_t s_count;
  __shared__ __attribute__((aligned(16))) int8_t s_region[kFrontWaves][10 * kSliceStride];
  const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  const uint32_t n_tiles = (uint32_t)a.TX * (uint32_t)a.TY * (uint32_t)a.TZ;
  if (tid == 0) s_count = 0u;
  __syncthreads();

  // which of this block's tiles can change in this layer?  A block owns CONSECUTIVE tiles (one row of tiles at
  // 512^3): their flag reads are coalesced and neighbouring tiles share halo rows in L1 / L2.  Spreading a block's
  // tiles over the volume for balance (the first version) cost 13 ms against 8 ms for the whole 512^3 build.
  if (wave == 0u) {
    const uint32_t tile = lane < kFrontTilesPerBlock ? blockIdx.x * kFrontTilesPerBlock + lane : n_tiles;
    bool active = false;
    if (tile < n_tiles) {
      a.flags_clear[tile] = 0;
      // flagged by whoever settled a voxel next to (or inside) this tile -- unless every voxel of the tile
      // is settled already (the front has passed): such a tile can never change again
      active = a.flags_cur[tile] != 0 && a.tile_done[tile] == 0;
    }
    const unsigned long long am = __ballot(active);
    if (active) s_list[__builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u))] = tile;
    if (lane == 0u) s_count = (uint32_t)__popcll(am);
  }
  __syncthreads();
  const uint32_t n_active = s_count;
  const int it = a.iteration;
  const bool rows_aligned = (a.X & 3) == 0;

  for (uint32_t k0 = 0; k0 < n_active; k0 += kFrontWaves) {
    const uint32_t k = k0 + wave;
    const bool have = k < n_active;
    int x0 = 0, y0 = 0, z0 = 0;
    uint32_t t = 0u;
    if (have) {
      t = s_list[k];
      x0 = (int)(t % (uint32_t)a.TX) * 8;
      y0 = (int)((t / (uint32_t)a.TX) % (uint32_t)a.TY) * 8;
      z0 = (int)(t / ((uint32_t)a.TX * (uint32_t)a.TY)) * 8;
      // tile + 1-voxel halo, neighbour coordinates clamped to the volume (signed_distance_field.cl:72)
      const bool wide = rows_aligned && x0 >= 4 && x0 + 12 <= a.X;  // one aligned 16-byte load per row
      for (unsigned r = lane; r < 100u; r += 64u) {
        const int rz = (int)(r / 10u), ry = (int)(r % 10u);
        const int gz = min(max(z0 - 1 + rz, 0), a.Z - 1), gy = min(max(y0 - 1 + ry, 0), a.Y - 1);
        const int8_t *row = a.sdf + ((size_t)gz * (size_t)a.Y + (size_t)gy) * (size_t)a.X;
        int8_t *dst = &s_region[wave][rz * kSliceStride + ry * kRowStride];
        if (wide) {
          // 16 bytes [x0-4, x0+12): the source is only 4-byte aligned (x0 - 4 = 4 mod 8), so four dword loads
          const uint32_t *src = reinterpret_cast<const uint32_t *>(row + x0 - 4);
          uint4 v;
          v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
          *reinterpret_cast<uint4 *>(dst) = v;
        } else {
          // bytes 0..2 and 13..15 of the row are never neighbours of an own voxel, but the packed-byte test below
          // classifies whole dwords: keep them in the value range (an arbitrary 0x80 would carry into byte 3)
          *reinterpret_cast<uint4 *>(dst) = uint4{0u, 0u, 0u, 0u};
#pragma unroll
          for (int rx = 0; rx < 10; ++rx) dst[3 + rx] = row[min(max(x0 - 1 + rx, 0), a.X - 1)];
        }
      }
    }
    // every wave works on its own s_region slice: only the wave's own LDS writes must be visible to its reads
    // (LDS operations of one wave execute in order), so a wave-level fence replaces the block barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int settled = 0;
    bool open_voxels = false;  // does this lane still hold a voxel that may settle in a later layer?
    unsigned face_mask = 0u;  // which faces of the tile this lane's settled voxels lie on: -x +x -y +y -z +z
    if (have) {
      const int ly = (int)(lane & 7u), lz = (int)(lane >> 3);
      const int y = y0 + ly, z = z0 + lz;
      if (y < a.Y && z < a.Z) {
        // A lane owns the 8 voxels of one x-row of the tile.  The reference's test (signed_distance_field.cl:56-112)
        //   |v| > it  and  the 8 corner neighbours all have one sign (zeros allowed)  and  their smallest |value| == it
        // is evaluated for the 8 voxels at once on packed bytes: the four neighbouring rows (y+-1, z+-1) are read as four
        // 16-byte LDS words, each byte is classified with carry-free SWAR arithmetic (all magnitudes are <= 127, so adding
        // 0x7F / subtracting from 0x80 | x never crosses a byte), the per-row flags are AND / OR-ed over the four rows, and
        // a voxel's corner neighbours are the flag bytes one to the left and one to the right of its own byte.  The first
        // version read 72 single bytes per lane and spent ~550 VALU instructions per row of 8 voxels; the layers of the 512^3
        // build were bound by exactly that arithmetic (profiles/r02_sdf_front_variants_negative_results.txt).
        const uint32_t b1 = 0x01010101u, b80 = 0x80808080u, b7f = 0x7F7F7F7Fu;
        const uint32_t itb = (uint32_t)it * b1, itp1b = (uint32_t)(it + 1) * b1, itp2b = (uint32_t)(it + 2) * b1;
        const int8_t *own_row = &s_region[wave][(lz + 1) * kSliceStride + (ly + 1) * kRowStride];
        uint32_t all_ne[4] = {~0u, ~0u, ~0u, ~0u}, all_ge[4] = {~0u, ~0u, ~0u, ~0u}, any_neg[4] = {0u, 0u, 0u, 0u},
                 any_pos[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint4 row = *reinterpret_cast<const uint4 *>(own_row + ((q & 1) ? kRowStride : -kRowStride) +
                                                             ((q & 2) ? kSliceStride : -kSliceStride));
          const uint32_t w[4] = {row.x, row.y, row.z, row.w};
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const uint32_t sgn = (w[d] >> 7) & b1;
            const uint32_t mag = (w[d] ^ ((sgn << 8) - sgn)) + sgn;  // |byte| per byte
            all_ne[d] &= (mag ^ itb) + b7f;                          // bit 7: |byte| != it
            all_ge[d] &= (mag | b80) - itb;                          // bit 7: |byte| >= it
            any_neg[d] |= w[d];                                      // bit 7: byte < 0
            any_pos[d] |= (mag + b7f) & ~w[d];                       // bit 7: byte > 0
          }
        }
        // flags of the bytes left (x - 1) and right (x + 1) of the own bytes 4..11, i.e. of dwords 1 and 2
        auto left = [](const uint32_t *f, int d) { return (f[d] << 8) | (f[d - 1] >> 24); };
        auto right = [](const uint32_t *f, int d) { return (f[d] >> 8) | (f[d + 1] << 24); };
        const uint32_t *own_words = reinterpret_cast<const uint32_t *>(own_row + 4);  // 4-byte aligned: two dword reads
        const uint32_t own[2] = {own_words[0], own_words[1]};
        uint32_t settle[2], opened[2], fresh[2];
        const bool can_settle = it + 1 < a.max_iterations;  // the reference only writes values below max_iterations
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int d = h + 1;
          const uint32_t sgn = (own[h] >> 7) & b1, neg_mask = (sgn << 8) - sgn;
          const uint32_t mag = (own[h] ^ neg_mask) + sgn;
          const uint32_t unsettled = (mag | b80) - itp1b;                                   // |v| > it
          const uint32_t eq_any = ~(left(all_ne, d) & right(all_ne, d));                    // some corner neighbour holds it
          const uint32_t ge_all = left(all_ge, d) & right(all_ge, d);                       // none holds less (or zero)
          const uint32_t mixed = (left(any_neg, d) | right(any_neg, d)) & (left(any_pos, d) | right(any_pos, d));
          // voxels beyond the volume's x extent (last tile of a row when X is not a multiple of 8) do not exist
          uint32_t valid = b80;
          if (x0 + 8 > a.X) {
            valid = 0u;
#pragma unroll
            for (int bx = 0; bx < 4; ++bx)
              if (x0 + h * 4 + bx < a.X) valid |= 0x80u << (8 * bx);
          }
          settle[h] = can_settle ? (unsettled & eq_any & ge_all & ~mixed & valid) : 0u;
          opened[h] = ((mag | b80) - itp2b) & ~settle[h] & valid;                           // |v| > it + 1 and not settled now
          const uint32_t sel = settle[h] >> 7, sel_mask = (sel << 8) - sel;
          fresh[h] = (own[h] & ~sel_mask) | (((itp1b ^ neg_mask) + sgn) & sel_mask);        // +-(it + 1) with the voxel's sign
        }
        settled = __popc(settle[0]) + __popc(settle[1]);
        open_voxels = (opened[0] | opened[1]) != 0u;
        if (settled) {
          int8_t *out = a.sdf + ((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)a.X + (size_t)x0;
          if (rows_aligned && x0 + 8 <= a.X) {
            if (settle[0]) *reinterpret_cast<uint32_t *>(out) = fresh[0];
            if (settle[1]) *reinterpret_cast<uint32_t *>(out + 4) = fresh[1];
          } else {
#pragma unroll
            for (int lx = 0; lx < 8; ++lx)
              if ((settle[lx >> 2] >> (8 * (lx & 3) + 7)) & 1u) out[lx] = (int8_t)(fresh[lx >> 2] >> (8 * (lx & 3)));
          }
          face_mask = ((settle[0] >> 7) & 1u) | ((settle[1] >> 31) ? 2u : 0u) | (ly == 0 ? 4u : 0u) | (ly == 7 ? 8u : 0u) |
                      (lz == 0 ? 16u : 0u) | (lz == 7 ? 32u : 0u);
        }
      }
    }
    if (have && __ballot(open_voxels) == 0ull && lane == 0u) a.tile_done[t] = 1;
    // per wave: count, and flag every tile that holds a corner neighbour of a voxel settled here: the
    // tile itself and the (up to 26) neighbours its settled boundary voxels touch
    const unsigned long long sm = __ballot(settled > 0);
    if (have && sm != 0ull) {
      int total = settled;
      unsigned touch = face_mask;
      for (int off = 32; off > 0; off >>= 1) {
        total += __shfl_xor(total, off);
        touch |= (unsigned)__shfl_xor((int)touch, off);
      }
      if (lane == 0u && total > 0) a.counters[it] = 1;  // non-zero marker (see k_sdf_base_front)
      if (lane < 27u) {
        const int dx = (int)(lane % 3u) - 1, dy = (int)((lane / 3u) % 3u) - 1, dz = (int)(lane / 9u) - 1;
        const bool ok_x = dx == 0 || (touch & (dx < 0 ? 1u : 2u)), ok_y = dy == 0 || (touch & (dy < 0 ? 4u : 8u)),
                   ok_z = dz == 0 || (touch & (dz < 0 ? 16u : 32u));
        const int nx = x0 / 8 + dx, ny = y0 / 8 + dy, nz = z0 / 8 + dz;
        if (ok_x && ok_y && ok_z && nx >= 0 && ny >= 0 && nz >= 0 && nx < a.TX && ny < a.TY && nz < a.TZ)
          a.flags_next[((size_t)nz * a.TY + ny) * a.TX + nx] = 1;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the region is overwritten by the wave's next tile
    __builtin_amdgcn_wave_barrier();
  }
}

static unsigned sdf_block(const SdfArgs &a) { return a.X <= 64 ? 64u : (a.X <= 128 ? 128u : 256u); }
static dim3 sdf_grid(const SdfArgs &a) {
  const unsigned b = sdf_block(a);
  return dim3(((unsigned)a.X + b - 1u) / b, (unsigned)a.Y, (unsigned)a.Z);
}

hipError_t launch_sdf_base(const SdfArgs &a, hipStream_t s) {
  if (a.tf.uses_gradient)
    hipLaunchKernelGGL(k_sdf_base<true>, sdf_grid(a), dim3(sdf_block(a)), 0, s, a);
  else
    hipLaunchKernelGGL(k_sdf_base<false>, sdf_grid(a), dim3(sdf_block(a)), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_sdf_base_front(const SdfArgs &a, uint8_t *flags, int32_t TX, int32_t TY, hipStream_t s) {
  const dim3 grid(((unsigned)a.X + kBaseX - 1u) / kBaseX, ((unsigned)a.Y + kBaseY - 1u) / kBaseY, ((unsigned)a.Z + kBaseZ - 1u) / kBaseZ);
  if (a.tf.uses_gradient)
    hipLaunchKernelGGL(k_sdf_base_front<true>, grid, dim3(256), 0, s, a, flags, TX, TY);
  else
    hipLaunchKernelGGL(k_sdf_base_front<false>, grid, dim3(256), 0, s, a, flags, TX, TY);
  return hipGetLastError();
}

hipError_t launch_sdf_front(const SdfFrontArgs &a, hipStream_t s) {
  const uint32_t n_tiles = (uint32_t)a.TX * (uint32_t)a.TY * (uint32_t)a.TZ;
  hipLaunchKernelGGL(k_sdf_front, dim3((n_tiles + kFrontTilesPerBlock - 1u) / kFrontTilesPerBlock), dim3(64 * kFrontWaves), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_sdf_layer(const SdfArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_sdf_layer, sdf_grid(a), dim3(sdf_block(a)), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Bit-parallel fused build (clwh_sdf_build, default).  The layer iteration is a breadth-first search over the
// "8 clamped corner neighbours" graph (above): with D(v) = the number of corner moves from v to the nearest
// non-homogeneous voxel, the converged image holds sign * min(D + 1, max_iterations), and a voxel only settles
// while D + 1 < max_iterations.  The search front does not need the byte image at all: the set
// R_r = {v : D(v) <= r} is ONE BIT per voxel (x-fastest rows of 32-bit words), and one layer is
//   R_{r+1} = R_r | shift_x(+-1, clamped)( R_r(y-1,z-1) | R_r(y+1,z-1) | R_r(y-1,z+1) | R_r(y+1,z+1) )   (rows clamped)
// -- a dozen word operations for 128 voxels.  A block keeps a 128 x 32 x 32 voxel region of R (its 64 x 16 x 16 core and a
// halo of 8 rows / 32 bits) in 16 KB of LDS and runs EIGHT layers on it before anything returns to memory: information
// travels one voxel per layer, so after 8 layers the core is exact although the halo's rim is not.  125 dependent
// launches become 16, none of them with a host round trip; blocks whose core is complete, or whose 27-neighbourhood
// holds no reached voxel yet, leave after reading a few state bytes.  A voxel's value is written once, in the launch
// in which its bit appears (layer index recorded bit-sliced per core word), with the sign of its event bit.
// Bit-exact against the oracle / the reference's golden vector like the byte front it replaces (tests/test_gpu_sdf.py).

// event bit of every voxel: one wave = 64 voxels along x = two words
template <bool USE_GRAD>
__global__ __launch_bounds__(256) void k_sdfbit_events(const SdfArgs a, uint32_t *__restrict__ ev, int32_t WP) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y, z = blockIdx.z;
  const VolumeIntLinear v{a.volume, a.X, a.Y, a.Z};
  bool e = false;
  if (x < a.X) e = event_at<USE_GRAD>(v, a.tf, a.cls_in, x, y, z);
  const unsigned long long m = __ballot(e);
  if ((threadIdx.x & 63u) == 0u) {
    const int w = x >> 5;  // x is a multiple of 64 here
    if (w < WP) {
      uint32_t *row = ev + ((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)WP;
      row[w] = (uint32_t)m;
      row[w + 1] = (uint32_t)(m >> 32);  // WP is even
    }
  }
}

// the same for rule tables on rows of a multiple of 8 voxels: a lane classifies the 8 voxels of one 16-byte load (with `gradient`
// rules: plus the four neighbouring rows' loads) and writes their byte of the bit image (387 -> 57 us at 512^3 without
// gradient rules: the per-voxel kernel was issue-bound)
// (row, unit) of a thread for kernels that work on `units` items per (y, z) row: blockIdx.x counts groups of rows
// (rows_per_block = 256 / units when a row has fewer than 256 units), blockIdx.y chunks of 256 units within a row
__device__ __forceinline__ bool sdfbit_row_unit(uint32_t units, size_t n_rows, size_t &row, uint32_t &unit) {
  if (units >= 256u) {
    row = blockIdx.x;
    unit = blockIdx.y * 256u + threadIdx.x;
  } else {
    const uint32_t rows_per_block = 256u / units, r = threadIdx.x / units;
    row = (size_t)blockIdx.x * rows_per_block + r;
    unit = threadIdx.x - r * units;
    if (r >= rows_per_block) return false;
  }
  return row < n_rows && unit < units;
}
static dim3 sdfbit_row_grid(uint32_t units, size_t n_rows) {
  if (units >= 256u) return dim3((unsigned)n_rows, (units + 255u) / 256u);
  const uint32_t rows_per_block = 256u / units;
  return dim3((unsigned)((n_rows + rows_per_block - 1u) / rows_per_block), 1u);
}

template <bool USE_GRAD>
__global__ __launch_bounds__(256) void k_sdfbit_events8(const SdfArgs a, uint8_t *__restrict__ ev_bytes, int32_t WP) {
  size_t row;
  uint32_t unit;
  const size_t n_rows = (size_t)a.Y * (size_t)a.Z;
  if (!sdfbit_row_unit((uint32_t)WP * 4u, n_rows, row, unit)) return;
  const int x0 = (int)unit * 8;
  uint32_t bits = 0u;
  if (x0 < a.X) {  // X is a multiple of 8: all eight voxels exist
    const int16_t *own = a.volume + row * (size_t)a.X + (size_t)x0;
    auto load8 = [](const int16_t *p, int (&v)[8]) {
      const uint4 q = *reinterpret_cast<const uint4 *>(p);
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int h = 0; h < 8; ++h) v[h] = (int)(int16_t)(w[h >> 1] >> (16 * (h & 1)));
    };
    int value[8], gradient[8];
    load8(own, value);
#pragma unroll
    for (int h = 0; h < 8; ++h) gradient[h] = 0;
    if (USE_GRAD) {
      // utility_filter.cl:2-35 at the voxel: central differences of the six neighbours (border texel 0), the float length
      // converted to short (signed_distance_field.cl:13-20) -- event_at<true> for eight voxels of one row
      const int z = (int)(row / (size_t)a.Y), y = (int)(row - (size_t)z * (size_t)a.Y);
      int ym[8], yp[8], zm[8], zp[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) ym[h] = yp[h] = zm[h] = zp[h] = 0;
      if (y > 0) load8(own - a.X, ym);
      if (y + 1 < a.Y) load8(own + a.X, yp);
      if (z > 0) load8(own - (size_t)a.X * (size_t)a.Y, zm);
      if (z + 1 < a.Z) load8(own + (size_t)a.X * (size_t)a.Y, zp);
      const int left = x0 > 0 ? (int)own[-1] : 0, right = x0 + 8 < a.X ? (int)own[8] : 0;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const float dx = (float)((h < 7 ? value[h < 7 ? h + 1 : 7] : right) - (h > 0 ? value[h > 0 ? h - 1 : 0] : left));
        const float dy = (float)(yp[h] - ym[h]);
        const float dz = (float)(zp[h] - zm[h]);
        gradient[h] = (int)(short)f2i(sqrtf((dx * dx + dy * dy) + dz * dz));
      }
    }
    // tf_eval for the eight voxels at once, the rule (one scalar load of its bounds) in the outer loop: the first matching
    // rule decides, a terminal rule that does not match decides "no event" (render_device.hpp: tf_eval)
    uint32_t undecided = 0xFFu;
    for (int k = 0; k < a.tf.n && undecided; ++k) {
      const int lo = a.tf.rules[k].v_lo, hi = a.tf.rules[k].v_hi, g_lo = a.tf.rules[k].g_lo, g_hi = a.tf.rules[k].g_hi;
      const bool use_g = USE_GRAD && (a.tf.rules[k].flags & TF_USE_GRADIENT);
      uint32_t m = 0u;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        bool hit = value[h] >= lo && value[h] <= hi;
        if (use_g) hit = hit && gradient[h] >= g_lo && gradient[h] <= g_hi;
        m |= hit ? (1u << h) : 0u;
      }
      bits |= m & undecided;
      undecided &= ~m;
      if (a.tf.rules[k].flags & TF_TERMINAL) undecided = 0u;
    }
  }
  ev_bytes[row * (size_t)WP * 4u + unit] = (uint8_t)bits;
}

// bits of a row shifted to x - 1 and x + 1 with the reference's clamp (signed_distance_field.cl:72): the neighbour of
// x = 0 at x - 1 is x = 0 itself, the neighbour of x = X - 1 at x + 1 is itself -- `both` = bits at either neighbour
__device__ __forceinline__ uint32_t sdfbit_x_neighbours(uint32_t prev, uint32_t cur, uint32_t next, uint32_t clampfix) {
  return ((cur << 1) | (prev >> 31)) | ((cur >> 1) | (next << 31)) | (cur & clampfix);
}

// The reached-set buffers are TILED by region: tile (bx, by, bz) = the region's core, 48 * core_z rows of two words, row (cy, cz) at
// ((cz * 48 + cy) * 2): the rows a wave of k_sdfbit_layers loads / stores (64 lanes = 64 consecutive y) are contiguous 8-byte pairs.
// (In the x-fastest layout of the event bits the same rows lie 64 bytes apart at 512^3: every lane its own cache line, and the
// address unit, not the layers, set the pace of a region.)  Tiles are padded to full size; rows beyond the volume stay zero.
struct SdfBitTiles {
  int32_t BX, BY, core_z;
  __device__ __forceinline__ size_t tile_words() const { return (size_t)2 * 48u * (size_t)core_z; }
  __device__ __forceinline__ size_t word(int w, int y, int z) const {
    const int bx = w >> 1, by = y / 48, cy = y - by * 48, bz = z / core_z, cz = z - bz * core_z;
    return (((size_t)bz * BY + by) * BX + bx) * tile_words() + (size_t)((cz * 48 + cy) * 2 + (w & 1));
  }
};

// non-homogeneous voxels (create_base_image: some clamped corner neighbour's event flag differs) = the seeds R_0
__device__ __forceinline__ void sdfbit_seed_word(const uint32_t *__restrict__ ev, uint32_t *__restrict__ r0, int32_t X, int32_t Y, int32_t Z, int32_t WP,
                                                 int32_t *presence, const SdfBitTiles &tiles, int w, int y, int z) {
  const size_t rowi = (size_t)z * (size_t)Y + (size_t)y;
  const int x_lo = w * 32;
  uint32_t valid = 0u;
  if (x_lo < X) valid = (X - x_lo >= 32) ? 0xFFFFFFFFu : ((1u << (X - x_lo)) - 1u);
  const uint32_t lastbit = (((X - 1) >> 5) == w) ? (1u << ((X - 1) & 31)) : 0u;
  const uint32_t firstbit = (w == 0) ? 1u : 0u;
  const uint32_t own = ev[rowi * (size_t)WP + w];
  uint32_t differs = 0u;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ny = min(max(y + ((c & 1) ? 1 : -1), 0), Y - 1);
    const int nz = min(max(z + ((c & 2) ? 1 : -1), 0), Z - 1);
    const uint32_t *row = ev + ((size_t)nz * Y + ny) * (size_t)WP;
    const uint32_t cur = row[w], prev = w > 0 ? row[w - 1] : 0u, next = w + 1 < WP ? row[w + 1] : 0u;
    const uint32_t left = ((cur << 1) | (prev >> 31)) | (cur & firstbit);   // value at clamp(x - 1)
    const uint32_t right = ((cur >> 1) | (next << 31)) | (cur & lastbit);   // value at clamp(x + 1)
    differs |= (left ^ own) | (right ^ own);
  }
  differs &= valid;
  r0[tiles.word(w, y, z)] = differs;
  if (differs) presence[0] = 1;  // non-zero marker, plain store (see k_sdf_base_front)
}
__global__ __launch_bounds__(256) void k_sdfbit_seed(const uint32_t *__restrict__ ev, uint32_t *__restrict__ r0, int32_t X, int32_t Y,
                                                      int32_t Z, int32_t WP, int32_t *presence, SdfBitTiles tiles) {
  size_t rowi;
  uint32_t unit;
  if (!sdfbit_row_unit((uint32_t)WP, (size_t)Y * (size_t)Z, rowi, unit)) return;
  const int z = (int)(rowi / (size_t)Y), y = (int)(rowi - (size_t)z * (size_t)Y);
  sdfbit_seed_word(ev, r0, X, Y, Z, WP, presence, tiles, (int)unit, y, z);
}
// rows of a multiple of WPB words (16, 32 or 64 lanes along x): a block takes sixteen consecutive rows, so that every 128-byte line of the
// tiled image is written whole by one block (see k_sdfbit_expand16_rows16)
template <int WPB>
__global__ __launch_bounds__(16 * WPB) void k_sdfbit_seed_rows16(const uint32_t *__restrict__ ev, uint32_t *__restrict__ r0, int32_t X, int32_t Y,
                                                                 int32_t Z, int32_t WP, int32_t *presence, SdfBitTiles tiles) {
  const int w = (int)blockIdx.x * WPB + (int)(threadIdx.x % (unsigned)WPB), y = (int)blockIdx.y * 16 + (int)(threadIdx.x / (unsigned)WPB);
  if (y >= Y) return;
  sdfbit_seed_word(ev, r0, X, Y, Z, WP, presence, tiles, w, y, (int)blockIdx.z);
}

// The values, once: bit planes of the layer index + the final reached set + the event bits -> one signed byte per voxel.
// A seed (reached, index 0) holds +-1, a voxel reached by layer `index` holds +-(index + 1), an unreached one +-max_iterations
// (create_base_image's values for the first and the last, signed_distance_field.cl:40-53; the sign is the event class).
__device__ __forceinline__ uint32_t sdfbit_index_bits(const uint32_t *__restrict__ planes, size_t plane_words, size_t word, int p) {
  return planes[(size_t)p * plane_words + word];
}
__global__ __launch_bounds__(256) void k_sdfbit_expand(const uint32_t *__restrict__ ev, const uint32_t *__restrict__ reached, const uint32_t *__restrict__ planes,
                                                        size_t plane_words, int8_t *__restrict__ sdf, int32_t X, int32_t Y, int32_t Z, int32_t WP,
                                                        int32_t max_iterations, SdfBitTiles tiles) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y, z = blockIdx.z;
  if (x >= X) return;
  const size_t roww = ((size_t)z * Y + y) * (size_t)WP + (size_t)(x >> 5), tw = tiles.word(x >> 5, y, z);
  const uint32_t sh = (uint32_t)(x & 31);
  const uint32_t e = (ev[roww] >> sh) & 1u, r = (reached[tw] >> sh) & 1u;
  uint32_t index = 0u;
  if (r) {
#pragma unroll
    for (int p = 0; p < 7; ++p) index |= ((sdfbit_index_bits(planes, plane_words, tw, p) >> sh) & 1u) << p;
  }
  const int val = r ? (int)index + 1 : max_iterations;
  sdf[((size_t)z * Y + y) * (size_t)X + (size_t)x] = (int8_t)(e ? -val : val);
}

// the same, sixteen voxels (one 16-byte store) per lane: rows of a multiple of 16 voxels
__device__ __forceinline__ void sdfbit_expand16_voxels(const uint32_t *__restrict__ ev, const uint32_t *__restrict__ reached, const uint32_t *__restrict__ planes,
                                                       size_t plane_words, int8_t *__restrict__ sdf, int32_t X, int32_t WP, int32_t max_iterations,
                                                       const SdfBitTiles &tiles, int x0, int y, int z, size_t row) {
  const uint32_t b1 = 0x01010101u;
  const size_t tw = tiles.word(x0 >> 5, y, z);
  const uint32_t sh = (uint32_t)(x0 & 31);
  const uint32_t e16 = (ev[row * (size_t)WP + (size_t)(x0 >> 5)] >> sh) & 0xFFFFu, r16 = (reached[tw] >> sh) & 0xFFFFu;
  uint32_t p16[7];
#pragma unroll
  for (int p = 0; p < 7; ++p) p16[p] = 0u;
  if (r16 != 0u) {  // nothing reached here (the far field beyond 126 layers, empty volumes): no plane is read
#pragma unroll
    for (int p = 0; p < 7; ++p) p16[p] = (sdfbit_index_bits(planes, plane_words, tw, p) >> sh) & 0xFFFFu;
  }
  uint32_t out[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    auto spread = [&](uint32_t bits16) { return __umul24((bits16 >> (4 * q)) & 0xFu, 0x204081u) & b1; };  // four bits -> the low bit of four bytes
    auto bytes_ff = [](uint32_t ones) { return (ones << 8) - ones; };                                        // 0 / 1 per byte -> 0x00 / 0xFF
    uint32_t idx = 0u;
#pragma unroll
    for (int p = 0; p < 7; ++p) idx += spread(p16[p]) << p;                                                  // <= 127 per byte
    const uint32_t rb = bytes_ff(spread(r16)), eb = spread(e16);
    const uint32_t val = ((idx + b1) & rb) | (((uint32_t)max_iterations * b1) & ~rb);                        // reached: index + 1 (<= 128 - 1); else max
    out[q] = (val ^ bytes_ff(eb)) + eb;                                                                      // two's complement per byte where the voxel is an event
  }
  *reinterpret_cast<uint4 *>(sdf + row * (size_t)X + (size_t)x0) = uint4{out[0], out[1], out[2], out[3]};
}
__global__ __launch_bounds__(256) void k_sdfbit_expand16(const uint32_t *__restrict__ ev, const uint32_t *__restrict__ reached, const uint32_t *__restrict__ planes,
                                                          size_t plane_words, int8_t *__restrict__ sdf, int32_t X, int32_t Y, int32_t Z, int32_t WP,
                                                          int32_t max_iterations, SdfBitTiles tiles) {
  size_t row;
  uint32_t unit;
  if (!sdfbit_row_unit((uint32_t)(X / 16), (size_t)Y * (size_t)Z, row, unit)) return;
  const int z = (int)(row / (size_t)Y), y = (int)(row - (size_t)z * (size_t)Y);
  sdfbit_expand16_voxels(ev, reached, planes, plane_words, sdf, X, WP, max_iterations, tiles, (int)unit * 16, y, z, row);
}
// Rows of a multiple of 16 * UPB voxels (UPB = 32 or 64 lanes along x): a block takes UPB * 16 voxels of SIXTEEN consecutive rows -- the rows
// whose words share a 128-byte line in the tiled bit images (a tile keeps its 48 rows' word pairs contiguous).  With one or two rows per
// block eight consecutive blocks -- on eight XCDs -- each fetched every line of the planes: 32 GB read for 9 GB of bit images at 2048^3.
template <int UPB>
__global__ __launch_bounds__(16 * UPB) void k_sdfbit_expand16_rows16(const uint32_t *__restrict__ ev, const uint32_t *__restrict__ reached,
                                                                     const uint32_t *__restrict__ planes, size_t plane_words, int8_t *__restrict__ sdf,
                                                                     int32_t X, int32_t Y, int32_t Z, int32_t WP, int32_t max_iterations, SdfBitTiles tiles) {
  const int u = (int)(threadIdx.x % (unsigned)UPB), r = (int)(threadIdx.x / (unsigned)UPB);
  const int x0 = ((int)blockIdx.x * UPB + u) * 16, y = (int)blockIdx.y * 16 + r, z = (int)blockIdx.z;
  if (y >= Y) return;  // (x0 < X: X is a multiple of 16 * UPB)
  sdfbit_expand16_voxels(ev, reached, planes, plane_words, sdf, X, WP, max_iterations, tiles, x0, y, z, (size_t)z * (size_t)Y + (size_t)y);
}

// A barrier of k_sdfbit_layers: LDS only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores and atomics (a release fence
// at workgroup scope) -- here the write-back of a region (bit rows, plane ORs, state bytes), which nobody reads before the next launch; with
// it, the first barrier of a block's NEXT region stalled until those partial-line stores had been acknowledged.  LDS operations are
// still complete (lgkmcnt(0)) before the wave arrives, and the compiler may not move memory operations across it.
__device__ __forceinline__ void sdfbit_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int kBitCoreY = 48, kBitHalo = 8, kBitRows = 4;  // a wave = 64 rows along y (48 core + 2 x 8 halo) x 4 rows along z
// a block = NW waves (8 or 16) = NW strips of 4 z-rows: region z = 4 NW, core z = 4 NW - 16 (SdfBitArgs::core_z)

// block state: 0 = no reached voxel in the core, 1 = some, 2 = all (just now: the other bit buffer is not complete yet), 3 = all, both buffers.
// A region with reached voxels wakes the EMPTY regions among its 26 neighbours whose core lies within 8 voxels (Chebyshev; a corner move
// changes every coordinate by at most one) of the box around its reached voxels: lane q < 27 stamps neighbour q for the launch `stamp - 1`.
// No loads: the list kernel then needs one byte per region instead of up to 27 dependent state / box reads.
__device__ __forceinline__ void sdfbit_wake_neighbours(const SdfBitArgs &a, int bx, int by, int bz, int x0, int x1, int y0, int y1, int z0, int z1,
                                                       unsigned q, uint8_t stamp) {
  if (q >= 27u) return;
  const int nx = bx + (int)(q % 3u) - 1, ny = by + (int)((q / 3u) % 3u) - 1, nz = bz + (int)(q / 9u) - 1;
  if (nx < 0 || ny < 0 || nz < 0 || nx >= a.BX || ny >= a.BY || nz >= a.BZ) return;
  const int cx0 = nx * 64, cx1 = min(cx0 + 63, a.X - 1), cy0 = ny * kBitCoreY, cy1 = min(cy0 + kBitCoreY - 1, a.Y - 1), cz0 = nz * a.core_z,
            cz1 = min(cz0 + a.core_z - 1, a.Z - 1);
  const int rx0 = bx * 64 + x0, rx1 = bx * 64 + x1, ry0 = by * kBitCoreY + y0, ry1 = by * kBitCoreY + y1, rz0 = bz * a.core_z + z0, rz1 = bz * a.core_z + z1;
  const int gx = max(max(rx0 - cx1, cx0 - rx1), 0), gy = max(max(ry0 - cy1, cy0 - ry1), 0), gz = max(max(rz0 - cz1, cz0 - rz1), 0);
  if (gx <= kBitHalo && gy <= kBitHalo && gz <= kBitHalo) a.wake[((size_t)nz * a.BY + ny) * a.BX + nx] = stamp;
}

__global__ __launch_bounds__(64) void k_sdfbit_state(const SdfBitArgs a) {
  const int b = blockIdx.x;
  const int bx = b % a.BX, by = (b / a.BX) % a.BY, bz = b / (a.BX * a.BY);
  const unsigned lane = threadIdx.x;
  bool any = false, all = true;
  uint32_t orx[2] = {0u, 0u};
  int y0 = 255, y1 = -1, z0 = 255, z1 = -1;
  for (int r = (int)lane; r < kBitCoreY * a.core_z; r += 64) {
    const int cy = r % kBitCoreY, cz = r / kBitCoreY;
    const int gy = by * kBitCoreY + cy, gz = bz * a.core_z + cz;
    if (gy >= a.Y || gz >= a.Z) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gw = 2 * bx + j, x_lo = gw * 32;
      if (gw >= a.WP || x_lo >= a.X) continue;
      const uint32_t valid = (a.X - x_lo >= 32) ? 0xFFFFFFFFu : ((1u << (a.X - x_lo)) - 1u);
      const uint32_t wv = a.r_in[(size_t)b * (size_t)(2 * kBitCoreY * a.core_z) + (size_t)(r * 2 + j)];
      any |= wv != 0u;
      all &= wv == valid;
      orx[j] |= wv;
      if (wv) { y0 = min(y0, cy); y1 = max(y1, cy); z0 = min(z0, cz); z1 = max(z1, cz); }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    orx[0] |= (uint32_t)__shfl_xor((int)orx[0], off); orx[1] |= (uint32_t)__shfl_xor((int)orx[1], off);
    y0 = min(y0, __shfl_xor(y0, off)); y1 = max(y1, __shfl_xor(y1, off));
    z0 = min(z0, __shfl_xor(z0, off)); z1 = max(z1, __shfl_xor(z1, off));
  }
  const bool w_any = __ballot(any) != 0ull, w_all = __ballot(!all) == 0ull;
  if (lane == 0u) a.state[b] = w_all ? 2 : (w_any ? 1 : 0);
  if (w_any) {
    const unsigned long long orx64 = (unsigned long long)orx[0] | ((unsigned long long)orx[1] << 32);
    sdfbit_wake_neighbours(a, bx, by, bz, __ffsll((long long)orx64) - 1, 63 - __clzll((long long)orx64), y0, y1, z0, z1, lane, 1);  // launch 0
  }
}

// lane i <- lane i - 1 / lane i + 1 of the wave (0 beyond the ends): the neighbouring rows along y
__device__ __forceinline__ uint32_t sdfbit_lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t sdfbit_lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true); }

// the blocks that can change in the next launch: a front inside, complete since the last launch (the other buffer is still to be
// brought up to date), or empty and woken for this launch by a neighbour (sdfbit_wake_neighbours).
// (The first version woke an empty block as soon as a neighbour held ANY reached voxel, up to 16 launches before the front
// arrived; the second read the 27 neighbours' states and boxes here: 11 us per launch of dependent loads.)
__device__ __forceinline__ void sdfbit_build_list(const SdfBitArgs &a, uint32_t *count, int first, int stride) {
  const int n_blocks = a.BX * a.BY * a.BZ;
  const int rounds = (n_blocks + stride - 1) / stride;  // every wave runs the same number of rounds (ballots below)
  const uint8_t stamp = (uint8_t)(a.launch + 1);
  for (int r = 0; r < rounds; ++r) {
    const int b = first + r * stride;
    bool active = false, complete = false;
    if (b < n_blocks) {
      const int st = a.state[b];
      const uint8_t wk = a.wake[b];
      active = st == 1 || st == 2 || (st == 0 && wk == stamp);
      complete = st == 2;
    }
    const unsigned long long m = __ballot(active);
    if (m == 0ull) continue;
    uint32_t base = 0u;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (active)
      a.list[base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (uint32_t)b | (complete ? 0x80000000u : 0u);
  }
}

// (one launch per list: letting the last block of the layers kernel make the next list was tried -- a single block needs 30 us
// for the dependent state / bounding-box reads that 2816 threads spread over the chip finish in 5)
__global__ __launch_bounds__(256) void k_sdfbit_list(const SdfBitArgs a) {
  sdfbit_build_list(a, a.list_count, (int)(blockIdx.x * 256u + threadIdx.x), (int)(gridDim.x * 256u));
}

// Eight layers on a 128 x 64 x (4 waves) voxel region of the reached set.  A lane owns four consecutive z-rows (four words
// each: 32 halo bits, the block's 64 core bits, 32 halo bits) at one y; the rows y - 1 / y + 1 are the neighbouring LANES
// (two DPP moves per word, no memory), the rows z - 1 / z + 1 the lane's own registers -- only a strip's first and last
// row cross to the neighbouring wave through LDS (one barrier per layer, ping-pong buffers).  The first version kept the
// whole region in LDS and read four neighbour rows per row and layer.  INTERIOR: the region touches no face of the volume
// (no clamped neighbour, no missing row or bit): about half the instructions.
struct SdfBitLane {
  uint32_t cur[kBitRows][4];
  // newly reached bits of the core words (1, 2) and the layer they appeared in, bit-sliced
  uint32_t rec_any[kBitRows][2], rec_b0[kBitRows][2], rec_b1[kBitRows][2], rec_b2[kBitRows][2];
  uint32_t step_mask;
};

// REC_LDS: the bit-sliced layer records (32 registers of a lane) live in LDS instead -- `rec` points at this lane's first word, the words of
// (plane, row, word) lie kRecStride apart -- and the exchange rows are single-buffered (a second barrier per layer): 41 KB of LDS and 80
// VGPRs, three blocks per CU instead of two.
template <int NW>
struct SdfBitRec {
  static constexpr int kCoreStrips = NW - 2 * kBitHalo / kBitRows, kStride = kCoreStrips * kBitCoreY;  // words between consecutive (plane, row, word)
};
template <int NW, bool INTERIOR, bool REC_LDS>
__device__ __forceinline__ void sdfbit_steps(SdfBitLane &L, uint4 (*s_x)[NW][2][64], uint32_t *rec, int steps, int strip, unsigned lane, bool core_lane, bool core_strip,
                                             const uint32_t (&valid)[4], const uint32_t (&clampfix)[4], bool y_in, bool y_border, int zfirst, int Z) {
  constexpr int kRegZ = kBitRows * NW;
  constexpr int kRecStride = SdfBitRec<NW>::kStride;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k >= steps) break;
    // after k layers a row is exact if it lies at least k rows inside the region: layer k + 1 is computed for the rows
    // [k + 1, region - 2 - k] from the y-neighbour unions V of the rows [k, region - 1 - k] (the other rows' V is
    // computed too -- branch-free -- and only ever read by rows that are not needed either)
    uint32_t v[kBitRows][4];
#pragma unroll
    for (int i = 0; i < kBitRows; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[i][j] = sdfbit_lane_prev(L.cur[i][j]) | sdfbit_lane_next(L.cur[i][j]);
        if (!INTERIOR) v[i][j] |= y_border ? L.cur[i][j] : 0u;
      }
    uint4 *xbuf = &s_x[REC_LDS ? 0 : (k & 1)][0][0][0];
    xbuf[(strip * 2 + 0) * 64 + lane] = uint4{v[0][0], v[0][1], v[0][2], v[0][3]};
    xbuf[(strip * 2 + 1) * 64 + lane] = uint4{v[kBitRows - 1][0], v[kBitRows - 1][1], v[kBitRows - 1][2], v[kBitRows - 1][3]};
    sdfbit_lds_barrier();
    uint4 below = uint4{0u, 0u, 0u, 0u}, above = uint4{0u, 0u, 0u, 0u};
    if (strip > 0) below = xbuf[((strip - 1) * 2 + 1) * 64 + lane];
    if (strip < NW - 1) above = xbuf[((strip + 1) * 2 + 0) * 64 + lane];
    if (REC_LDS) sdfbit_lds_barrier();  // one buffer: everybody has read its neighbours' rows before the next layer overwrites them
#pragma unroll
    for (int i = 0; i < kBitRows; ++i) {
      const int rz = kBitRows * strip + i, gz = zfirst + i;
      bool need = rz >= k + 1 && rz <= kRegZ - 2 - k;  // wave-uniform
      if (!INTERIOR) need = need && gz >= 0 && gz < Z;
      if (!need) continue;
      const bool z_border = !INTERIOR && (gz == 0 || gz == Z - 1);
      uint32_t u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t lo = i > 0 ? v[i - 1][j] : (j == 0 ? below.x : j == 1 ? below.y : j == 2 ? below.z : below.w);
        const uint32_t hi = i < kBitRows - 1 ? v[i + 1][j] : (j == 0 ? above.x : j == 1 ? above.y : j == 2 ? above.z : above.w);
        u[j] = lo | hi;
        if (!INTERIOR) u[j] |= z_border ? v[i][j] : 0u;
      }
      uint32_t nxt[4];
      if (INTERIOR) {
        nxt[0] = L.cur[i][0] | sdfbit_x_neighbours(0u, u[0], u[1], 0u);
        nxt[1] = L.cur[i][1] | sdfbit_x_neighbours(u[0], u[1], u[2], 0u);
        nxt[2] = L.cur[i][2] | sdfbit_x_neighbours(u[1], u[2], u[3], 0u);
        nxt[3] = L.cur[i][3] | sdfbit_x_neighbours(u[2], u[3], 0u, 0u);
      } else {
        nxt[0] = L.cur[i][0] | (sdfbit_x_neighbours(0u, u[0], u[1], clampfix[0]) & valid[0]);
        nxt[1] = L.cur[i][1] | (sdfbit_x_neighbours(u[0], u[1], u[2], clampfix[1]) & valid[1]);
        nxt[2] = L.cur[i][2] | (sdfbit_x_neighbours(u[1], u[2], u[3], clampfix[2]) & valid[2]);
        nxt[3] = L.cur[i][3] | (sdfbit_x_neighbours(u[2], u[3], 0u, clampfix[3]) & valid[3]);
        if (!y_in) nxt[0] = nxt[1] = nxt[2] = nxt[3] = 0u;  // rows beyond the volume do not exist
      }
#ifndef CLVR_EXP_SDF_NO_REC
      if (core_strip) {
        const uint32_t nb0 = core_lane ? (nxt[1] & ~L.cur[i][1]) : 0u, nb1 = core_lane ? (nxt[2] & ~L.cur[i][2]) : 0u;
        if (REC_LDS) {
          // ds_or without return; plane p of (row i, word j) at rec[((p * kBitRows + i) * 2 + j) * kRecStride]; plane 0 = "reached in this launch"
          if (nb0) { atomicOr(rec + ((0 * kBitRows + i) * 2 + 0) * kRecStride, nb0);
                     if (k & 1) atomicOr(rec + ((1 * kBitRows + i) * 2 + 0) * kRecStride, nb0);
                     if (k & 2) atomicOr(rec + ((2 * kBitRows + i) * 2 + 0) * kRecStride, nb0);
                     if (k & 4) atomicOr(rec + ((3 * kBitRows + i) * 2 + 0) * kRecStride, nb0); }
          if (nb1) { atomicOr(rec + ((0 * kBitRows + i) * 2 + 1) * kRecStride, nb1);
                     if (k & 1) atomicOr(rec + ((1 * kBitRows + i) * 2 + 1) * kRecStride, nb1);
                     if (k & 2) atomicOr(rec + ((2 * kBitRows + i) * 2 + 1) * kRecStride, nb1);
                     if (k & 4) atomicOr(rec + ((3 * kBitRows + i) * 2 + 1) * kRecStride, nb1); }
        } else {
          L.rec_any[i][0] |= nb0; L.rec_any[i][1] |= nb1;
          if (k & 1) { L.rec_b0[i][0] |= nb0; L.rec_b0[i][1] |= nb1; }
          if (k & 2) { L.rec_b1[i][0] |= nb0; L.rec_b1[i][1] |= nb1; }
          if (k & 4) { L.rec_b2[i][0] |= nb0; L.rec_b2[i][1] |= nb1; }
        }
        if (nb0 | nb1) L.step_mask |= 1u << k;
      }
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j) L.cur[i][j] = nxt[j];
    }
  }
}

// The grid is persistent: its blocks take the active regions from the list k_sdfbit_list made (a launch over ALL regions
// spent 30 us on the inactive ones alone); a block's first region is list[blockIdx.x], the following ones come from a queue.
template <int NW, bool REC_LDS>
__global__ __launch_bounds__(64 * NW, REC_LDS ? 6 : 4) void k_sdfbit_layers(const SdfBitArgs a) {
  constexpr int kRegZ = kBitRows * NW, kCoreZ = kRegZ - 2 * kBitHalo;
  __shared__ uint4 s_x[REC_LDS ? 1 : 2][NW][2][64];
  constexpr int kRecStride = SdfBitRec<NW>::kStride;
  __shared__ uint32_t s_rec[REC_LDS ? 4 * kBitRows * 2 * kRecStride : 1];
  if (REC_LDS)
    for (int i = (int)threadIdx.x; i < 4 * kBitRows * 2 * kRecStride; i += 64 * NW) s_rec[i] = 0u;  // (a region leaves them cleared)
  __shared__ uint32_t s_all, s_any, s_steps, s_entry, s_orx[2];
  __shared__ int s_box[4];  // min y, max y, min z, max z of the core's reached voxels
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const int strip = __builtin_amdgcn_readfirstlane((int)(tid >> 6));  // wave-uniform: scalar branches on the row ranges below
  const uint32_t n_active = *a.list_count;
  const bool core_lane = lane >= (unsigned)kBitHalo && lane < (unsigned)(kBitHalo + kBitCoreY);
  const bool core_strip = strip >= kBitHalo / kBitRows && strip < NW - kBitHalo / kBitRows;
  for (uint32_t round = 0u;; ++round) {
    sdfbit_lds_barrier();  // the previous region's flags and exchange rows are no longer read
    if (tid == 0u) {
      // dynamic: regions differ in cost (complete ones only copy); a static round-robin over the list measured 1.84 ms against 1.60.
      // (Fetching the NEXT region's ticket and list entry while the block works on the current one -- two dependent round trips off
      // every visit -- was measured at 1.41 ms against 1.33: two more live registers in a kernel that already spills.)
      s_entry = round == 0u ? blockIdx.x : gridDim.x + atomicAdd(a.list_head, 1u);
      s_all = 1u; s_any = 0u; s_steps = 0u; s_orx[0] = 0u; s_orx[1] = 0u;
      s_box[0] = 255; s_box[1] = -1; s_box[2] = 255; s_box[3] = -1;
    }
    sdfbit_lds_barrier();
#ifdef CLVR_SDFBIT_TIMING
    const bool probe = tid == 64u * 2u + 8u;
    const unsigned long long tq0 = wall_clock64();
#endif
    const uint32_t entry = s_entry;
    if (entry >= n_active) return;
    const uint32_t item = a.list[entry];  // block | complete-since-the-previous-launch << 31
    const int b = (int)(item & 0x7FFFFFFFu);
    const int bx = b % a.BX, by = (b / a.BX) % a.BY, bz = b / (a.BX * a.BY);
    uint32_t valid[4], clampfix[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gw = 2 * bx - 1 + j, x_lo = gw * 32;
      valid[j] = 0u;
      clampfix[j] = 0u;
      if (gw >= 0 && gw < a.WP && x_lo < a.X) {
        valid[j] = (a.X - x_lo >= 32) ? 0xFFFFFFFFu : ((1u << (a.X - x_lo)) - 1u);
        clampfix[j] = (gw == 0 ? 1u : 0u) | ((((a.X - 1) >> 5) == gw) ? (1u << ((a.X - 1) & 31)) : 0u);
      }
    }
    if (item >> 31) {
      // complete since the previous launch: bring the other buffer up to date, then never come back
      for (int r = (int)tid; r < kBitCoreY * kCoreZ; r += 64 * NW) {
        const int gy = by * kBitCoreY + (r % kBitCoreY), gz = bz * kCoreZ + (r / kBitCoreY);
        if (gy >= a.Y || gz >= a.Z) continue;
#pragma unroll
        for (int j = 1; j <= 2; ++j) {
          const int gw = 2 * bx - 1 + j;
          if (gw < a.WP) a.r_out[(size_t)b * (size_t)(2 * kBitCoreY * kCoreZ) + (size_t)(r * 2 + j - 1)] = valid[j];
        }
      }
      if (tid == 0u) a.state[b] = 3;
      sdfbit_wake_neighbours(a, bx, by, bz, 0, 63, 0, kBitCoreY - 1, 0, kCoreZ - 1, tid, (uint8_t)(a.launch + 2));
      continue;
    }

    const int gy = by * kBitCoreY - kBitHalo + (int)lane;
    const int zfirst = bz * kCoreZ - kBitHalo + kBitRows * strip;  // gz of this lane's row 0
    const bool y_in = gy >= 0 && gy < a.Y;
    const bool y_border = gy == 0 || gy == a.Y - 1;  // the clamped neighbour along y is the row itself (signed_distance_field.cl:72)
    // no face of the volume inside the region or next to it: every word, row and neighbour exists, nothing is clamped
    const bool interior = bx >= 1 && (2 * bx + 3) * 32 < a.X && by * kBitCoreY - kBitHalo >= 1 && by * kBitCoreY - kBitHalo + 63 <= a.Y - 2 &&
                          bz * kCoreZ - kBitHalo >= 1 && bz * kCoreZ - kBitHalo + kRegZ - 1 <= a.Z - 2;
#ifdef CLVR_SDFBIT_TIMING
    const unsigned long long tq1 = wall_clock64();
#endif
    SdfBitLane L;
    // this lane's y: its tile row (by - 1 / by / by + 1) and row inside the tile
    const int lane_by = by + ((int)lane < kBitHalo ? -1 : ((int)lane >= kBitHalo + kBitCoreY ? 1 : 0));
    const int lane_cy = (int)lane < kBitHalo ? kBitCoreY - kBitHalo + (int)lane : ((int)lane >= kBitHalo + kBitCoreY ? (int)lane - kBitHalo - kBitCoreY : (int)lane - kBitHalo);
    const bool y_tile = lane_by >= 0 && lane_by < a.BY;
    constexpr size_t kTileWords = (size_t)2 * kBitCoreY * kCoreZ;
#pragma unroll
    for (int i = 0; i < kBitRows; ++i) {
      const int rz = kBitRows * strip + i;
      const int row_bz = bz + (rz < kBitHalo ? -1 : (rz >= kBitHalo + kCoreZ ? 1 : 0));
      const int row_cz = rz < kBitHalo ? kCoreZ - kBitHalo + rz : (rz >= kBitHalo + kCoreZ ? rz - kBitHalo - kCoreZ : rz - kBitHalo);
      const bool row_in = y_tile && row_bz >= 0 && row_bz < a.BZ;  // the tile exists; its rows beyond the volume hold zeros
      // all loads are issued unconditionally (one round trip): a missing tile reads tile 0 and is masked afterwards
      const size_t t_mid = row_in ? (((size_t)row_bz * a.BY + lane_by) * a.BX + bx) : (size_t)0;
      const size_t in_tile = (size_t)((row_cz * kBitCoreY + lane_cy) * 2);
      const uint32_t *mid = a.r_in + t_mid * kTileWords + in_tile;
      const bool left_in = row_in && bx > 0, right_in = row_in && bx + 1 < a.BX;
      const uint32_t w0 = (left_in ? mid - kTileWords : a.r_in)[1], w3 = (right_in ? mid + kTileWords : a.r_in)[0];
      const uint2 w12 = *reinterpret_cast<const uint2 *>(mid);
      L.cur[i][0] = left_in ? w0 : 0u;
      L.cur[i][1] = row_in ? w12.x : 0u;
      L.cur[i][2] = row_in ? w12.y : 0u;
      L.cur[i][3] = right_in ? w3 : 0u;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (!REC_LDS) L.rec_any[i][j] = L.rec_b0[i][j] = L.rec_b1[i][j] = L.rec_b2[i][j] = 0u;
    }
    L.step_mask = 0u;
#ifdef CLVR_SDFBIT_TIMING
    if (probe) __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long tq2 = wall_clock64();
#endif
    // this lane's first record word (core lanes of core strips only; the others never touch the records)
    uint32_t *rec = s_rec + (core_strip && core_lane ? (strip - kBitHalo / kBitRows) * kBitCoreY + ((int)lane - kBitHalo) : 0);
    if (interior)
      sdfbit_steps<NW, true, REC_LDS>(L, s_x, rec, a.steps, strip, lane, core_lane, core_strip, valid, clampfix, y_in, y_border, zfirst, a.Z);
    else
      sdfbit_steps<NW, false, REC_LDS>(L, s_x, rec, a.steps, strip, lane, core_lane, core_strip, valid, clampfix, y_in, y_border, zfirst, a.Z);

#ifdef CLVR_SDFBIT_TIMING
    const unsigned long long tq3 = wall_clock64();
#endif
    // core rows back to the other bit buffer; the block's state and the box around its reached voxels (whom to wake) for the next launch
    bool any = false, all = true;
    if (core_strip && core_lane && y_in) {
#pragma unroll
      for (int i = 0; i < kBitRows; ++i) {
        const int gz = zfirst + i;
        if (gz < 0 || gz >= a.Z) continue;
        const int cz = kBitRows * strip + i - kBitHalo;
        *reinterpret_cast<uint2 *>(a.r_out + (size_t)b * kTileWords + (size_t)((cz * kBitCoreY + ((int)lane - kBitHalo)) * 2)) = uint2{L.cur[i][1], L.cur[i][2]};
        any |= (L.cur[i][1] | L.cur[i][2]) != 0u;
        all &= L.cur[i][1] == valid[1] && L.cur[i][2] == valid[2];  // a word beyond the volume: 0 == 0
        if (L.cur[i][1] | L.cur[i][2]) {  // words beyond the volume are zero (valid mask)
          atomicMin(&s_box[2], cz);
          atomicMax(&s_box[3], cz);
        }
      }
    }
    if (any) {
      s_any = 1u;
      uint32_t o1 = 0u, o2 = 0u;
#pragma unroll
      for (int i = 0; i < kBitRows; ++i) { o1 |= L.cur[i][1]; o2 |= L.cur[i][2]; }  // rows outside the volume hold zeros
      if (o1) atomicOr(&s_orx[0], o1);
      if (o2) atomicOr(&s_orx[1], o2);
      atomicMin(&s_box[0], (int)lane - kBitHalo);
      atomicMax(&s_box[1], (int)lane - kBitHalo);
    }
    if (!all) s_all = 0u;
    if (L.step_mask) atomicOr(&s_steps, L.step_mask);

    // values: NOT written here.  Round 2 let every lane rewrite the 64 bytes of its rows that gained voxels -- a load and a store of
    // 16 bytes per lane at a 512-byte stride, partial lines whose completion the next barrier waited for: two thirds of a region's
    // time.  Now the layer in which a voxel was reached goes into seven bit planes (tiled like the reached sets: a wave's rows are
    // contiguous) with fire-and-forget atomic ORs -- a voxel is reached exactly once, so the launches never write the same bit -- and
    // k_sdfbit_expand turns planes + final reached set + event bits into bytes ONCE, after the last launch, with full-line stores.
    // layer index = r0 + k + 1 (1..127; r0 = 8 x launch, k the layer inside the launch as recorded bit-sliced in rec_b0..2)
#ifndef CLVR_EXP_SDF_NO_PLANES
    if (core_strip && core_lane && y_in) {
      const uint32_t hi_lo = (uint32_t)a.r0 >> 3, hi_carry = hi_lo + 1u;  // bits 3.. of the index while k + 1 < 8 / when k + 1 == 8
      // (a plane holds at most 2^28 words: 32-bit word offsets from the plane's own, wave-uniform base keep the addresses out of the VGPRs)
      const uint32_t lane_word = (uint32_t)b * (uint32_t)kTileWords + (uint32_t)(((kBitRows * strip - kBitHalo) * kBitCoreY + ((int)lane - kBitHalo)) * 2);
#pragma unroll
      for (int i = 0; i < kBitRows; ++i) {
        // the row's two core words are one aligned 8-byte pair in every plane: one 64-bit OR per plane and row
        uint32_t r_any[2], r_b0[2], r_b1[2], r_b2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (REC_LDS) {
            r_any[j] = rec[((0 * kBitRows + i) * 2 + j) * kRecStride];
            r_b0[j] = r_b1[j] = r_b2[j] = 0u;
          } else {
            r_any[j] = L.rec_any[i][j]; r_b0[j] = L.rec_b0[i][j]; r_b1[j] = L.rec_b1[i][j]; r_b2[j] = L.rec_b2[i][j];
          }
        }
        if ((r_any[0] | r_any[1]) == 0u) continue;  // (then the row also lies inside the volume)
        if (REC_LDS) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            r_b0[j] = rec[((1 * kBitRows + i) * 2 + j) * kRecStride];
            r_b1[j] = rec[((2 * kBitRows + i) * 2 + j) * kRecStride];
            r_b2[j] = rec[((3 * kBitRows + i) * 2 + j) * kRecStride];
#pragma unroll
            for (int p = 0; p < 4; ++p) rec[((p * kBitRows + i) * 2 + j) * kRecStride] = 0u;  // cleared for the block's next region
          }
        }
        uint32_t v[7][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint32_t any = r_any[j], b0 = r_b0[j], b1 = r_b1[j], b2 = r_b2[j];
          const uint32_t m7 = b0 & b1 & b2, m = any & ~m7;  // k == 7: the index's low three bits are 0 and bit 3.. carries
          v[0][j] = ~b0 & m; v[1][j] = (b1 ^ b0) & m; v[2][j] = (b2 ^ (b1 & b0)) & m;  // k + 1, bit-sliced
#pragma unroll
          for (int p = 0; p < 4; ++p) v[3 + p][j] = (((hi_lo >> p) & 1u) ? m : 0u) | (((hi_carry >> p) & 1u) ? m7 : 0u);
        }
        const uint32_t w = lane_word + (uint32_t)(i * kBitCoreY * 2);
#pragma unroll
        for (int p = 0; p < 7; ++p)
          if (v[p][0] | v[p][1])
            atomicOr(reinterpret_cast<unsigned long long *>(a.planes + (size_t)p * a.plane_words + w), (unsigned long long)v[p][0] | ((unsigned long long)v[p][1] << 32));
      }
    }
#endif
#ifdef CLVR_SDFBIT_TIMING
#ifndef CLVR_EXP_SDF_PROBE_NOWAIT
    if (probe) __builtin_amdgcn_s_waitcnt(0);
#endif
    const unsigned long long tq4 = wall_clock64();
#endif
    sdfbit_lds_barrier();
#ifdef CLVR_SDFBIT_TIMING
    if (probe) {
      atomicAdd(&a.timing[0], 1ull); atomicAdd(&a.timing[1], tq1 - tq0); atomicAdd(&a.timing[2], tq2 - tq1); atomicAdd(&a.timing[3], tq3 - tq2);
      atomicAdd(&a.timing[4], tq4 - tq3); atomicAdd(&a.timing[5], wall_clock64() - tq4); atomicAdd(&a.timing[6], interior ? 1ull : 0ull);
    }
#endif
    if (s_any) {
      const unsigned long long orx64 = (unsigned long long)s_orx[0] | ((unsigned long long)s_orx[1] << 32);
      sdfbit_wake_neighbours(a, bx, by, bz, __ffsll((long long)orx64) - 1, 63 - __clzll((long long)orx64), s_box[0], s_box[1], s_box[2], s_box[3], tid,
                             (uint8_t)(a.launch + 2));  // for the next launch
    }
    if (tid == 0u) {
      a.state[b] = s_all ? 2 : (s_any ? 1 : 0);
      for (uint32_t m = s_steps; m; m &= m - 1u) a.presence[a.r0 + __ffs((int)m)] = 1;  // layer r0 + k + 1 settled something
    }
  }
}

void sdfbit_block_grid(int X, int Y, int Z, int waves, int32_t *BX, int32_t *BY, int32_t *BZ, int32_t *core_z) {
  *core_z = kBitRows * waves - 2 * kBitHalo;
  *BX = (X + 63) / 64;
  *BY = (Y + kBitCoreY - 1) / kBitCoreY;
  *BZ = (Z + *core_z - 1) / *core_z;
}

hipError_t launch_sdfbit_events(const SdfArgs &a, uint32_t *ev, int32_t WP, hipStream_t s) {
  if (!a.cls_in && (a.X % 8) == 0) {
    const dim3 grid8 = sdfbit_row_grid((uint32_t)WP * 4u, (size_t)a.Y * (size_t)a.Z);
    if (a.tf.uses_gradient)
      hipLaunchKernelGGL(k_sdfbit_events8<true>, grid8, dim3(256), 0, s, a, (uint8_t *)ev, WP);
    else
      hipLaunchKernelGGL(k_sdfbit_events8<false>, grid8, dim3(256), 0, s, a, (uint8_t *)ev, WP);
    return hipGetLastError();
  }
  const dim3 grid(((unsigned)a.X + 255u) / 256u, (unsigned)a.Y, (unsigned)a.Z);
  if (a.tf.uses_gradient)
    hipLaunchKernelGGL(k_sdfbit_events<true>, grid, dim3(256), 0, s, a, ev, WP);
  else
    hipLaunchKernelGGL(k_sdfbit_events<false>, grid, dim3(256), 0, s, a, ev, WP);
  return hipGetLastError();
}

hipError_t launch_sdfbit_seed(const SdfBitArgs &a, hipStream_t s) {
  const size_t n_rows = (size_t)a.Y * (size_t)a.Z;
  const SdfBitTiles tiles{a.BX, a.BY, a.core_z};
  const dim3 rows16((unsigned)a.WP, ((unsigned)a.Y + 15u) / 16u, (unsigned)a.Z);  // x: divided by WPB below
  if ((a.WP % 64) == 0)
    hipLaunchKernelGGL(k_sdfbit_seed_rows16<64>, dim3(rows16.x / 64u, rows16.y, rows16.z), dim3(1024), 0, s, a.ev, a.r_out, a.X, a.Y, a.Z, a.WP, a.presence, tiles);
  else if ((a.WP % 32) == 0)
    hipLaunchKernelGGL(k_sdfbit_seed_rows16<32>, dim3(rows16.x / 32u, rows16.y, rows16.z), dim3(512), 0, s, a.ev, a.r_out, a.X, a.Y, a.Z, a.WP, a.presence, tiles);
  else if ((a.WP % 16) == 0)
    hipLaunchKernelGGL(k_sdfbit_seed_rows16<16>, dim3(rows16.x / 16u, rows16.y, rows16.z), dim3(256), 0, s, a.ev, a.r_out, a.X, a.Y, a.Z, a.WP, a.presence, tiles);
  else
    hipLaunchKernelGGL(k_sdfbit_seed, sdfbit_row_grid((uint32_t)a.WP, n_rows), dim3(256), 0, s, a.ev, a.r_out, a.X, a.Y, a.Z, a.WP, a.presence, tiles);
  return hipGetLastError();
}

hipError_t launch_sdfbit_expand(const SdfBitArgs &a, const uint32_t *reached, int32_t max_iterations, hipStream_t s) {
  const size_t n_rows = (size_t)a.Y * (size_t)a.Z;
  const SdfBitTiles tiles{a.BX, a.BY, a.core_z};
  if ((a.X % 1024) == 0 && max_iterations >= 1) {
    hipLaunchKernelGGL(k_sdfbit_expand16_rows16<64>, dim3((unsigned)(a.X / 1024), ((unsigned)a.Y + 15u) / 16u, (unsigned)a.Z), dim3(1024), 0, s, a.ev, reached,
                       (const uint32_t *)a.planes, a.plane_words, a.sdf, a.X, a.Y, a.Z, a.WP, max_iterations, tiles);
  } else if ((a.X % 512) == 0 && max_iterations >= 1) {
    hipLaunchKernelGGL(k_sdfbit_expand16_rows16<32>, dim3((unsigned)(a.X / 512), ((unsigned)a.Y + 15u) / 16u, (unsigned)a.Z), dim3(512), 0, s, a.ev, reached,
                       (const uint32_t *)a.planes, a.plane_words, a.sdf, a.X, a.Y, a.Z, a.WP, max_iterations, tiles);
  } else if ((a.X % 16) == 0 && max_iterations >= 1) {
    hipLaunchKernelGGL(k_sdfbit_expand16, sdfbit_row_grid((uint32_t)(a.X / 16), n_rows), dim3(256), 0, s, a.ev, reached, (const uint32_t *)a.planes,
                       a.plane_words, a.sdf, a.X, a.Y, a.Z, a.WP, max_iterations, tiles);
  } else {
    hipLaunchKernelGGL(k_sdfbit_expand, dim3(((unsigned)a.X + 255u) / 256u, (unsigned)a.Y, (unsigned)a.Z), dim3(256), 0, s, a.ev, reached,
                       (const uint32_t *)a.planes, a.plane_words, a.sdf, a.X, a.Y, a.Z, a.WP, max_iterations, tiles);
  }
  return hipGetLastError();
}

hipError_t launch_sdfbit_state(const SdfBitArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_sdfbit_state, dim3((unsigned)(a.BX * a.BY * a.BZ)), dim3(64), 0, s, a);
  return hipGetLastError();
}

// one launch = the list of the regions that can change + up to eight layers on them (persistent grid of `grid_blocks`)
hipError_t launch_sdfbit_layers(const SdfBitArgs &a, int waves, unsigned grid_blocks, bool rec_in_lds, hipStream_t s) {
  const unsigned n_blocks = (unsigned)(a.BX * a.BY * a.BZ);
  hipLaunchKernelGGL(k_sdfbit_list, dim3(std::min((n_blocks + 255u) / 256u, 1024u)), dim3(256), 0, s, a);
  const unsigned grid = std::min(n_blocks, grid_blocks);
  if (waves == 16)
    hipLaunchKernelGGL((k_sdfbit_layers<16, false>), dim3(grid), dim3(64 * 16), 0, s, a);
  else if (rec_in_lds)
    hipLaunchKernelGGL((k_sdfbit_layers<8, true>), dim3(grid), dim3(64 * 8), 0, s, a);
  else
    hipLaunchKernelGGL((k_sdfbit_layers<8, false>), dim3(grid), dim3(64 * 8), 0, s, a);
  return hipGetLastError();
}

}  // namespace clvr
