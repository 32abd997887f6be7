// render_kernels.hip -- the `render` pass for gfx950, organised for wave64 hardware.
//
// The reference runs one work-item per pixel through the whole path (ray_marching.cl:152-199):
// camera ray -> box entry -> primary march -> on a Hit, two distribution rays of up to three
// marches each -> accumulate -> read the accumulator back.  On a 64-wide wave that shape wastes the
// machine twice: only a fraction of the pixels hit anything, and the lanes that do walk paths of
// very different lengths.  Here the pass is split where the divergence is:
//
//   k_repack   (when volume / SDF / TF changed)  bricked step bytes + hit records (packed_volume.hpp)
//   k_primary  (when the camera changed)         one lane per pixel: ray, box entry, primary march;
//                                                hits are compacted into 64-byte records with a
//                                                wave ballot + prefix (one atomic per wave);
//                                                misses keep their environment colour
//   k_bounce   (every launch, 1..64 seeds)       persistent waves pull (hit, seed) items from eight
//                                                work queues; each lane runs the sample's two
//                                                distribution rays as a small state machine; idle
//                                                lanes are refilled by ballot/prefix compaction once
//                                                enough of them are idle; march steps and event
//                                                handling run in separate wave-wide phases
//   k_env_fixup / k_commit                       exact environment lookups the fast path could not
//                                                certify; per-hit sums -> float4 accumulator
//   k_resolve  (when a frame is wanted)          read the accumulator AFTER the pass (deterministic;
//                                                one legal outcome of the reference's race, SURVEY
//                                                fact 4), tone curve, RGBA8
//   k_ao       (shading = CLWH_SHADE_AO)         compute_ao (ray_marching.cl:104-149) per primary hit
//
// The primary march does not depend on the pass's seed, so its result is kept while camera, volume,
// SDF and transfer function stay the same; per sample, every float operation is the one the reference
// kernel performs, in the same order (device_math.hpp), only scheduled differently.
#include <algorithm>

#include "render_device.hpp"

namespace clvr {

// blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of tile slots so
// that neighbouring tiles (which walk neighbouring voxels) share one L2
__device__ __forceinline__ uint32_t xcd_contiguous_slot(uint32_t b, uint32_t nblocks) {
  const uint32_t per = nblocks >> 3;
  if (per == 0u || b >= (per << 3)) return b;
  return (b & 7u) * per + (b >> 3);
}

__device__ __forceinline__ VolumePacked make_volume(const RenderArgs &a) {
  return VolumePacked{a.grec, a.stepb, a.volume_lin, a.sdf_lin, a.X, a.Y, a.Z, a.NBX, a.NBY};
}

// n / d and n % d for n, d < 2^24 through the float reciprocal (a 32-bit integer division is ~35 VALU instructions)
__device__ __forceinline__ uint32_t udivmod24(uint32_t n, uint32_t d, uint32_t &rem) {
  // n < 2^24 and d < 2^24 convert exactly; the estimate's relative error is below 2^-22, so for the quotients met
  // here (below 2^20, or d a power of two) it is off by at most one either way
  uint32_t q = (uint32_t)((float)n * __builtin_amdgcn_rcpf((float)d));
  uint32_t qd;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(qd) : "v"(q), "v"(d));
  int32_t r = (int32_t)(n - qd);
  if (r < 0) { q -= 1u; r += (int32_t)d; }
  if (r >= (int32_t)d) { q += 1u; r -= (int32_t)d; }
  rem = (uint32_t)r;
  return q;
}

// number of set bits of `mask` below this lane
__device__ __forceinline__ unsigned prefix_count(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ unsigned lane_id() { return prefix_count(~0ull); }

// ------------------------------------------------------------------------------------------------
// volume + SDF + transfer function -> bricked step bytes + hit records (packed_volume.hpp).
// A block turns a 64 x 8 x 8 box of the caller's x-fastest images (eight bricks side by side: whole 128-byte lines of the volume; with
// 32 voxels two blocks -- on two XCDs -- each fetched every line: 1.7 GB read for 0.4 GB of images at 512^3) into brick order: the box
// and its one-voxel halo are read ONCE with coalesced loads (a wave reads 64 consecutive voxels of a row) and staged
// in LDS; the central differences and the class come from there; every wave then writes whole 4x4x4 sub-bricks (512
// contiguous bytes of hit records, 64 of step bytes).  The first version let each wave gather its sub-brick's rows and
// the six taps straight from global memory: 8-byte pieces of 128-byte lines, 2.1 GB fetched for a 0.27 GB volume.
#ifndef CLVR_REPACK_X
#define CLVR_REPACK_X 64
#endif
constexpr int kRepackX = CLVR_REPACK_X;  // voxels per block along x (32 or 64)
constexpr int kRepackPitch = kRepackX + 16;  // LDS row: 7 unused shorts, x0 - 1, the box's voxels from a 16-byte aligned offset, one voxel beyond, padding
constexpr int kRepackX0 = 8;        // index of voxel x0 in a row
__global__ __launch_bounds__(256) void k_repack(const RepackArgs a) {
  __shared__ __attribute__((aligned(16))) int16_t s_val[10][10][kRepackPitch];  // [z][y][kRepackX0 + lx], lx = -1 .. 32: values with halo
  __shared__ __attribute__((aligned(16))) int8_t s_sdf[8][8][kRepackX];
  const int x0 = (int)blockIdx.x * kRepackX, y0 = (int)blockIdx.y * 8, z0 = (int)blockIdx.z * 8;
  const unsigned tid = threadIdx.x;
  if ((a.X & 15) == 0 && x0 + kRepackX <= a.X && ((reinterpret_cast<uintptr_t>(a.volume) | reinterpret_cast<uintptr_t>(a.sdf)) & 15u) == 0u) {
    // rows of a multiple of 16 voxels, box inside the volume along x: a lane moves 16 bytes (the staging loop below spent more
    // instructions on its per-voxel index arithmetic than the classification that follows)
    constexpr unsigned kChunks = kRepackX / 8;  // 16-byte pieces of a row of values
    for (unsigned i = tid; i < 10u * 10u * kChunks; i += 256u) {
      const unsigned row = i / kChunks, c = i % kChunks;
      const int ry = (int)(row % 10u), rz = (int)(row / 10u);
      const int y = y0 - 1 + ry, z = z0 - 1 + rz;
      uint4 v = uint4{0u, 0u, 0u, 0u};  // border texel (utility_filter.cl:2-35 reads with CLK_ADDRESS_CLAMP: 0 outside)
      int16_t left = 0, right = 0;
      if ((unsigned)y < (unsigned)a.Y && (unsigned)z < (unsigned)a.Z) {
        const int16_t *src = a.volume + ((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)a.X + (size_t)x0;
        v = *reinterpret_cast<const uint4 *>(src + 8 * c);
        if (c == 0u && x0 > 0) left = src[-1];
        if (c == kChunks - 1u && x0 + kRepackX < a.X) right = src[kRepackX];
      }
      *reinterpret_cast<uint4 *>(&s_val[rz][ry][kRepackX0 + 8 * (int)c]) = v;
      if (c == 0u) s_val[rz][ry][kRepackX0 - 1] = left;
      if (c == kChunks - 1u) s_val[rz][ry][kRepackX0 + kRepackX] = right;
    }
    constexpr unsigned kSdfChunks = kRepackX / 16;  // 16-byte pieces of a row of SDF bytes
    for (unsigned i = tid; i < 8u * 8u * kSdfChunks; i += 256u) {
      const unsigned row = i / kSdfChunks, c = i % kSdfChunks;
      const int ry = (int)(row & 7u), rz = (int)(row >> 3);
      const int y = y0 + ry, z = z0 + rz;
      uint4 v = uint4{0u, 0u, 0u, 0u};
      if (y < a.Y && z < a.Z) v = *reinterpret_cast<const uint4 *>(a.sdf + ((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)a.X + (size_t)x0 + 16u * c);
      *reinterpret_cast<uint4 *>(&s_sdf[rz][ry][16 * (int)c]) = v;
    }
  } else {
    constexpr unsigned kRX = kRepackX + 2;
    for (unsigned i = tid; i < 10u * 10u * kRX; i += 256u) {
      const int rx = (int)(i % kRX), ry = (int)((i / kRX) % 10u), rz = (int)(i / (10u * kRX));
      const int x = x0 - 1 + rx, y = y0 - 1 + ry, z = z0 - 1 + rz;
      int16_t v = 0;
      if ((unsigned)x < (unsigned)a.X && (unsigned)y < (unsigned)a.Y && (unsigned)z < (unsigned)a.Z)
        v = a.volume[((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)a.X + (size_t)x];
      s_val[rz][ry][kRepackX0 - 1 + rx] = v;
    }
    for (unsigned i = tid; i < 8u * 8u * (unsigned)kRepackX; i += 256u) {
      const int rx = (int)(i % (unsigned)kRepackX), ry = (int)((i / (unsigned)kRepackX) % 8u), rz = (int)(i / (8u * (unsigned)kRepackX));
      const int x = x0 + rx, y = y0 + ry, z = z0 + rz;
      int8_t v = 0;
      if (x < a.X && y < a.Y && z < a.Z) v = a.sdf[((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)a.X + (size_t)x];
      s_sdf[rz][ry][rx] = v;
    }
  }
  __syncthreads();
  const unsigned wave = tid >> 6, lane = tid & 63u;
  const size_t brick_row = ((size_t)blockIdx.z * (size_t)a.NBY + (size_t)blockIdx.y) * (size_t)a.NBX;
  // A wave writes the sub-bricks `wave` and `wave + 4` of the block's bricks: the lane's place inside the sub-brick is worked out twice, not
  // once per brick, and the rule table is walked without a per-lane `break` (a divergent loop exit costs more than the two rules it skips).
  for (unsigned half = 0u; half < 2u; ++half) {
   const unsigned sub = wave + 4u * half;
   unsigned ix, iy, iz;
   VolumePacked::inner_coords(sub * 64u + lane, ix, iy, iz);
   const int ly = (int)iy, lz = (int)iz, y = y0 + ly, z = z0 + lz;
   for (unsigned bq = 0u; bq < (unsigned)(kRepackX / 8); ++bq) {
    const int bx = (int)blockIdx.x * (kRepackX / 8) + (int)bq;
    if (bx >= a.NBX) break;
    const int lx = (int)(bq * 8u + ix);
    const int x = x0 + lx;
    uint2 r = uint2{0u, 0u};
    uint8_t q = 0u;
    bool record_read = false;  // can a march ever read this voxel's hit record?
    uint32_t free_min = 255u;  // for the exit certificates: 0 = this voxel may be an event / has no positive SDF value
    if (x < a.X && y < a.Y && z < a.Z) {
      const int cx = kRepackX0 + lx;
      const int value = s_val[lz + 1][ly + 1][cx];
      const int sd = s_sdf[lz][ly][lx];
      // central differences at the voxel's integer position, border 0 (utility_filter.cl:2-35)
      const int dx = s_val[lz + 1][ly + 1][cx + 1] - s_val[lz + 1][ly + 1][cx - 1];
      const int dy = s_val[lz + 1][ly + 2][cx] - s_val[lz + 1][ly][cx];
      const int dz = s_val[lz + 2][ly + 1][cx] - s_val[lz][ly + 1][cx];
      int gradient = 0;
      if (a.tf.uses_gradient) {
        const float gx = (float)dx, gy = (float)dy, gz = (float)dz;
        gradient = (int)(short)f2i(sqrtf((gx * gx + gy * gy) + gz * gz));  // |gradient| to short, as at the call (utility_ray.cl:134)
      }
      // class = 1 + index of the first matching rule; a terminal rule (`return (cond);`) ends the evaluation
      unsigned cls = a.cls_in ? a.cls_in[((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)a.X + (size_t)x] : 0u;
      // `maybe`: could this voxel be an event for SOME gradient?  (A rule that reads `gradient` is evaluated literally, with
      // other taps, at the rare positions whose +-1 taps are not the voxel's neighbours: its value window alone decides here.)
      bool maybe = cls != 0u, decided = a.cls_in != nullptr;
      for (int k = 0; k < a.tf.n; ++k) {  // (wave-uniform trip count; `decided` lanes only ride along)
        const TfRuleDev &rule = a.tf.rules[k];
        const bool in_window = value >= rule.v_lo && value <= rule.v_hi;
        bool m = in_window;
        if (rule.flags & TF_USE_GRADIENT) m = m && gradient >= rule.g_lo && gradient <= rule.g_hi;
        maybe = maybe || (!decided && in_window);
        if (!decided && m && cls == 0u) cls = (unsigned)k + 1u;
        decided = decided || m || (rule.flags & TF_TERMINAL) != 0;
      }
      r = VolumePacked::pack_hit(dx, dy, dz, cls);
      q = (uint8_t)((cls ? 0x80u : 0u) | (uint32_t)(sd > 0 ? sd : 0));
      free_min = (maybe || sd <= 0) ? 0u : (uint32_t)sd;
      record_read = maybe;  // (cls != 0 implies maybe)
    }
    const size_t out = ((brick_row + (size_t)bx) << 9) + sub * 64u + lane;
    // A hit record is read at Hit positions only -- the rule colour and the normal's gradient of a voxel whose class is not 0
    // (render_device.hpp: hit_color, hit_gradient_and_color, gradient_nn; positions with irregular taps and the border never use it).
    // Sub-bricks without such a voxel -- nine in ten on CT-like data -- keep whatever their 512 bytes held: two thirds of what this
    // kernel wrote (8 of 12 bytes per voxel) was never read.
    if (__ballot(record_read) != 0ull) a.grec[out] = r;
    a.stepb[out] = q;
    free_min = wave_min_u32(free_min);
    if (lane == 0u) atomicMin(&a.brick_min[brick_row + (size_t)bx], free_min);  // eight sub-bricks per brick
   }
  }
}

// Exit-certificate table (certify_exit below).  One entry per macro cell (16^3 voxels up to 512^3, growing with the volume:
// macro_cell_shift in clwh_internal.hpp) and direction octant o
// (o = [d.x < 0] | [d.y < 0] << 1 | [d.z < 0] << 2): a march that starts anywhere in the cell with a direction of that
// octant stays in the box between the cell and the volume corner the octant heads for.  The entry is an upper bound of
// the number of steps such a march takes until it leaves the volume, or 255 if the box is not free.
// Step 1: m = the smallest "free value" over the cell's bricks -- 0 if a voxel there may be an event or has an SDF value below
// kCertMinStep, else the smallest SDF value; all eight octant entries start as m.  (Round 2 also took the bricks AROUND the cell, "because
// the real march is off the ideal line by its roundings".  It is, but the proof never needed the line: a march's coordinates are monotone
// in binary32 as well -- adding a product of the direction's sign never moves a coordinate the other way -- so every position it visits
// lies in the box between its cell and the octant's corner exactly, whatever the roundings.  Without the dilation the instrumented
// oracle saves 6.58 instead of 5.79 step fetches per item, 7.68 with certificates tried from a step length of 8, and still counts zero
// wrong certificates: profiles/r03_exit_certificate_finer_estimate.txt.)
constexpr uint32_t kCertMinStep = 2u;
#ifndef CLVR_CERT_PHASE_MIN_LANES
#define CLVR_CERT_PHASE_MIN_LANES 16
#endif
constexpr int kCertPhaseMinLanes = CLVR_CERT_PHASE_MIN_LANES;  // certificates are looked up once this many lanes of a wave wait for one
__global__ __launch_bounds__(256) void k_macro_table(const uint32_t *__restrict__ brick_min, int NBX, int NBY, int NBZ,
                                                     uint2 *__restrict__ macro, int MNX, int MNY, int MNZ, int shift) {
  const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (c >= MNX * MNY * MNZ) return;
  const int cx = c % MNX, cy = (c / MNX) % MNY, cz = c / (MNX * MNY);
  uint32_t m = 255u;
  const int bpc = 1 << (shift - 3);  // bricks per cell and axis
  for (int bz = cz * bpc; bz <= min(cz * bpc + bpc - 1, NBZ - 1); ++bz)
    for (int by = cy * bpc; by <= min(cy * bpc + bpc - 1, NBY - 1); ++by)
      for (int bx = cx * bpc; bx <= min(cx * bpc + bpc - 1, NBX - 1); ++bx)
        m = min(m, brick_min[((size_t)bz * (size_t)NBY + (size_t)by) * (size_t)NBX + (size_t)bx]);
  if (m < kCertMinStep) m = 0u;
  m *= 0x01010101u;
  macro[c] = make_uint2(m, m);
}

// per-byte minimum of two packed octant entries
__device__ __forceinline__ uint2 min_bytes(uint2 a, uint2 b) {
  uint2 r;
  r.x = r.y = 0u;
  for (int k = 0; k < 32; k += 8) {
    r.x |= min((a.x >> k) & 0xFFu, (b.x >> k) & 0xFFu) << k;
    r.y |= min((a.y >> k) & 0xFFu, (b.y >> k) & 0xFFu) << k;
  }
  return r;
}
// Step 2, once per axis: octant entry o of a cell becomes the minimum over the cells from here to the end of the line in
// o's direction along this axis -- after the three passes, the minimum over the whole box.  One thread per line.
__global__ __launch_bounds__(64) void k_macro_octants(uint2 *__restrict__ macro, int MNX, int MNY, int MNZ, int axis) {
  const int n[3] = {MNX, MNY, MNZ};
  const int len = n[axis], u_n = n[(axis + 1) % 3], v_n = n[(axis + 2) % 3];
  const int line = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (line >= u_n * v_n) return;
  int c[3];
  c[(axis + 1) % 3] = line % u_n;
  c[(axis + 2) % 3] = line / u_n;
  // bytes of the octants that run towards coordinate 0 on this axis (octant o is byte o of the 8-byte entry)
  const uint2 neg = axis == 0 ? make_uint2(0xFF00FF00u, 0xFF00FF00u) : (axis == 1 ? make_uint2(0xFFFF0000u, 0xFFFF0000u) : make_uint2(0u, 0xFFFFFFFFu));
  auto at = [&](int i) -> uint2 & {
    c[axis] = i;
    return macro[((size_t)c[2] * (size_t)MNY + (size_t)c[1]) * (size_t)MNX + (size_t)c[0]];
  };
  uint2 run = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
  for (int i = len - 1; i >= 0; --i) {  // positive direction: accumulate from the far end backwards
    const uint2 v = at(i);
    run = min_bytes(run, v);
    at(i) = make_uint2((v.x & neg.x) | (run.x & ~neg.x), (v.y & neg.y) | (run.y & ~neg.y));
  }
  run = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
  for (int i = 0; i < len; ++i) {  // negative direction: from coordinate 0 forwards
    const uint2 v = at(i);
    run = min_bytes(run, v);
    at(i) = make_uint2((v.x & ~neg.x) | (run.x & neg.x), (v.y & ~neg.y) | (run.y & neg.y));
  }
}
// Step 3: the table keeps the box MINIMUM per octant (0 = the box is not free: no certificate).  Round 2 turned it into a step count
// right here -- box diagonal / minimum + 5 -- because the diagonal is the longest path inside the box; the ray's own distance to the
// face it leaves through is shorter and costs a dozen instructions in certify_exit: 5.79 instead of 5.16 step fetches saved per item
// on the instrumented oracle (tools/exit_certificate.py --variants --finer, profiles/r03_exit_certificate_finer_estimate.txt).
__global__ __launch_bounds__(256) void k_macro_bounds(uint2 *__restrict__ macro, int MNX, int MNY, int MNZ) {
  const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (c >= MNX * MNY * MNZ) return;
  const uint2 v = macro[c];
  uint2 r = make_uint2(0u, 0u);
  for (int o = 0; o < 8; ++o) {
    uint32_t m = ((o < 4 ? v.x : v.y) >> ((o & 3) * 8)) & 0xFFu;
    if (m == 255u) m = 0u;  // (a box without any brick: never the case for a cell inside the volume)
    if (o < 4) r.x |= m << (o * 8); else r.y |= m << ((o - 4) * 8);
  }
  macro[c] = r;
}

// ------------------------------------------------------------------------------------------------
// k_primary: ray_marching.cl:152-186 up to (and including) the first march_to_next_event of
// compute_light (:33), plus the hit's normal (:42).  One wave = one 8x8 pixel tile.
template <bool USE_GRAD>
__global__ __launch_bounds__(64) void k_primary(const RenderArgs a) {
  const uint32_t slot = xcd_contiguous_slot(blockIdx.x, a.num_tile_slots);
  int tx, ty;
  if (!tile_from_slot(a, slot, tx, ty)) return;
  const uint32_t lane = threadIdx.x;
  const uint32_t x = (uint32_t)tx * 8u + (lane & 7u);
  const uint32_t y = (uint32_t)ty * 8u + (lane >> 3);
  const uint32_t pslot = slot * 64u + lane;

  const VolumePacked vol = make_volume(a);
  const f3 cam_o = f3{a.cam_pos[0], a.cam_pos[1], a.cam_pos[2]};
  const f3 cam_d = f3{a.cam_dir[0], a.cam_dir[1], a.cam_dir[2]};
  const Ray vray = generate_ray(cam_o, cam_d, (int)x, (int)y, a.frame_w, a.frame_h);
  const float dx = (float)a.X, dy = (float)a.Y, dz = (float)a.Z;

  bool cut_ok;
  f3 cut_point;
  if (!(within(vray.origin.x, dx) && within(vray.origin.y, dy) && within(vray.origin.z, dz))) {
    cut_ok = cut_box(dx, dy, dz, vray, cut_point);
  } else {
    cut_ok = true;
    cut_point = vray.origin;
  }

  bool hit = false;
  Ray current_ray{cut_point, vray.direction};
  uint32_t current_color = 0u;
  // (tried in round 3: the exit-certificate table on the camera ray at its entry point -- "no event in the box towards the octant's
  // corner" would make the pixel a miss without a march.  From the default pose that box nearly always holds the object: 0.100 ms with
  // and without, tools/ab_list.sh; not kept.  Also without effect: the certified fast environment lookup for the miss pixels (0.100 /
  // 0.100 ms).  Without the hit counter's atomic -- one returning atomic per wave with a hit, all on one address -- the kernel takes
  // 0.089 ms: the rest is the marches, four generations of waves deep)
  if (cut_ok) {
    int ev;
    current_ray = march_to_next_event<USE_GRAD>(vol, a.tf, current_ray, ev, current_color);
    hit = (ev == EV_HIT);
  }

  // wave-level compaction of the hits: one atomic per wave, prefix of the ballot per lane
  const unsigned long long hit_mask = __ballot(hit);
  uint32_t base = 0u;
  if (hit_mask != 0ull) {
    const int leader = __ffsll((long long)hit_mask) - 1;
    if ((int)lane == leader) base = atomicAdd(&a.counters[0], (uint32_t)__popcll(hit_mask));
    base = __shfl(base, leader);
  }

  int64_t raw_entry = -1;
  if (hit) {
    const uint32_t h = base + prefix_count(hit_mask);
    const f3 normal = -normalize3(gradient_nn(vol, current_ray.origin));
    raw_entry = cache_entry_of(a.X, a.Z, current_ray.origin);
    int64_t entry = raw_entry;
    if (a.mode == CLWH_ACCUM_VOXEL_CACHE && !(entry >= 0 && entry < a.cache_entries)) entry = -2;
    uint4 q0, q1, q2, q3;
    q0.x = __float_as_uint(current_ray.origin.x); q0.y = __float_as_uint(current_ray.origin.y);
    q0.z = __float_as_uint(current_ray.origin.z); q0.w = __float_as_uint(current_ray.direction.x);
    q1.x = __float_as_uint(current_ray.direction.y); q1.y = __float_as_uint(current_ray.direction.z);
    q1.z = __float_as_uint(normal.x); q1.w = __float_as_uint(normal.y);
    q2.x = __float_as_uint(normal.z); q2.y = current_color;
    q2.z = (uint32_t)((uint64_t)entry & 0xFFFFFFFFull); q2.w = (uint32_t)((uint64_t)entry >> 32);
    q3.x = x | (y << 16); q3.y = pslot; q3.z = 0u; q3.w = 0u;
    uint4 *dst = reinterpret_cast<uint4 *>(&a.hits[h]);
    dst[0] = q0; dst[1] = q1; dst[2] = q2; dst[3] = q3;
    a.pix_slot[pslot] = PIX_HIT | h;
  } else {
    // miss: environment colour of the camera ray (ray_marching.cl:172-178, 188-195)
    const uint32_t e = sample_environment_map(a.env, a.env_w, a.env_h, vray.direction);
    a.pix_slot[pslot] = e & 0x00FFFFFFu;
  }
  if (a.hit_index_out) a.hit_index_out[(size_t)y * (size_t)a.launch_w + x] = raw_entry;
}

// ------------------------------------------------------------------------------------------------
// k_bounce: ray_marching.cl:39-77 for every (hit, seed) item.
//
// Lane state machine.  MARCH lanes take march steps; a lane that reaches an event (Hit / Exit /
// 70 steps, or a freshly fetched item) parks in EVENT until the wave runs its event phase; IDLE
// lanes have no item.
// One register holds both: ST_IDLE, ST_MARCH, ST_CERT (parked for an exit-certificate attempt), or ST_EVENT + the pending event.
enum : int { ST_IDLE = 0, ST_MARCH = 1, ST_CERT = 2, ST_EVENT = 8 };
enum : int { EV_START = 3,              // a freshly fetched item: start distribution ray 1
             EV_HIT_COLOR_PENDING = 4,  // a Hit whose rule colour is still to be fetched (classify_step DEFER_COLOR)
             EV_CHECK = 5 };            // the new position has no voxel: left the volume, or one of the rare in-between cases?
constexpr int kCertNever = 255;         // no step is this long

// Image-space accumulation of one launch: a sample adds r | g<<16 | b<<32 | 1<<48 to its HIT's 64-bit
// delta with ONE atomic (a launch has at most 64 seeds and a contribution is at most 255, so no field can
// carry); k_commit then folds the deltas into the caller's float4 buffer with plain read-modify-writes.
// Four float atomics per sample were a quarter of the kernel's L2-missing requests.
template <int MODE>
__device__ __forceinline__ void finish_item(const RenderArgs &a, int64_t entry, uint32_t hit, uint32_t gx, uint32_t gy,
                                            uint32_t bv_r, uint32_t bv_g, uint32_t bv_b) {
  // ray_marching.cl:75-76: halve (dist_count = 2), then add
  const uint32_t cr = (bv_r / 2u) & 0xFFFFu, cg = (bv_g / 2u) & 0xFFFFu, cb = (bv_b / 2u) & 0xFFFFu;
  if (MODE == CLWH_ACCUM_VOXEL_CACHE && a.grants == nullptr) {
    cache_add(a.cache, entry, cr, cg, cb, 0u);
  } else {
    const unsigned long long packed = (unsigned long long)cr | ((unsigned long long)cg << 16) |
                                      ((unsigned long long)cb << 32) | (1ull << 48);
    atomicAdd(a.delta + hit, packed);
  }
  if (a.contrib_out) {
    uint32_t *q = a.contrib_out + ((size_t)gy * (size_t)a.launch_w + gx) * 4;
    q[0] = cr; q[1] = cg; q[2] = cb; q[3] = 1u;
  }
}

// fold one launch's per-hit deltas into the float4 accumulation buffer (one lane per hit = per pixel)
__global__ __launch_bounds__(256) void k_commit(const RenderArgs a) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= (a.n_hits_on_device ? a.counters[0] : a.n_hits)) return;
  const unsigned long long d = a.delta[h];
  if (d == 0ull) return;
  a.delta[h] = 0ull;  // ready for the next launch: the host never clears the deltas (it does not know how many there are)
  const uint32_t pslot = a.hits[h].pslot;
  float4 acc = a.accum[pslot];
  // integer-valued floats below 2^24: exact
  acc.x += (float)(uint32_t)(d & 0xFFFFull);
  acc.y += (float)(uint32_t)((d >> 16) & 0xFFFFull);
  acc.z += (float)(uint32_t)((d >> 32) & 0xFFFFull);
  acc.w += (float)(uint32_t)(d >> 48);
  a.accum[pslot] = acc;
}

// ------------------------------------------------------------------------------------------------
// Planned voxel-cache launches.  The reference takes a token per sample with an atomic on the voxel's entry and adds the sample with
// two more (utility.cl:20-54).  With the seeds of a launch fused, the 64 samples of a pixel -- and those of every other pixel that hit
// the same voxel -- do that to ONE 8-byte entry at the same time: 17.2 ms for the launch that takes 3.9 ms in image space.  Which
// samples get a voxel's remaining tokens is unspecified in the reference (whoever reaches the atomic first); how many is not:
// min(requests, 256 - count).  So the tokens are dealt out before the launch: the camera's hits are grouped by voxel once (a stable
// sort of their cache entries), k_vox_grant walks each group and gives hit after hit as many of the launch's seeds as the voxel has
// tokens left, adds the tokens to the entry's count, and the launch runs without a single atomic on the cache: a granted sample
// accumulates into its hit's 64-bit delta like an image-space sample, and k_commit_voxel adds each hit's sum to its voxel with two
// atomics per hit instead of three per sample.  Counts are exact, entries below the cap equal the reference's bit for bit.
__global__ __launch_bounds__(256) void k_vox_keys(const RenderArgs a, int64_t *__restrict__ keys, uint32_t *__restrict__ iota, uint32_t n) {
  const uint32_t h = blockIdx.x * 256u + threadIdx.x;
  if (h >= n) return;
  int64_t key = kVoxKeyNone;
  if (h < a.counters[0]) {
    const HitRec &r = a.hits[h];
    const int64_t e = (int64_t)(((uint64_t)(uint32_t)r.entry_hi << 32) | (uint64_t)(uint32_t)r.entry_lo);
    key = e >= 0 ? e : kVoxKeyInvalid;
  }
  keys[h] = key;
  iota[h] = h;
}

// one lane per sorted position; the lane at the head of a voxel's group deals the launch's tokens to the group's hits, in hit order
__global__ __launch_bounds__(256) void k_vox_grant(const RenderArgs a, const int64_t *__restrict__ keys, const uint32_t *__restrict__ order,
                                                   uint32_t n, uint32_t *__restrict__ grants) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const int64_t e = keys[i];
  if (e >= kVoxKeyInvalid) {
    if (e == kVoxKeyInvalid) grants[order[i]] = 0u;  // the hit lies outside the cache: never a token
    return;
  }
  if (i > 0u && keys[i - 1] == e) return;
  uint32_t *word1 = a.cache + 2 * e + 1;
  const uint32_t w1 = *word1;
  const uint32_t count = w1 >> 16;
  uint32_t remaining = count < 256u ? 256u - count : 0u, dealt = 0u;
  for (uint32_t j = i; j < n && keys[j] == e; ++j) {
    const uint32_t g = min((uint32_t)a.n_seeds, remaining);
    grants[order[j]] = g;
    remaining -= g;
    dealt += g;
  }
  *word1 = w1 + (dealt << 16);  // the tokens (utility.cl:28-31); the sums follow in k_commit_voxel
}

__global__ __launch_bounds__(256) void k_commit_voxel(const RenderArgs a) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= (a.n_hits_on_device ? a.counters[0] : a.n_hits)) return;
  const unsigned long long d = a.delta[h];
  if (d == 0ull) return;
  a.delta[h] = 0ull;
  const HitRec &r = a.hits[h];
  const int64_t e = (int64_t)(((uint64_t)(uint32_t)r.entry_hi << 32) | (uint64_t)(uint32_t)r.entry_lo);
  // <= 256 contributions of <= 255 per voxel in total: no lane carries (the count was added by k_vox_grant)
  atomicAdd(a.cache + 2 * e, (uint32_t)(d & 0xFFFFull) | ((uint32_t)((d >> 16) & 0xFFFFull) << 16));
  atomicAdd(a.cache + 2 * e + 1, (uint32_t)((d >> 32) & 0xFFFFull));
}

// Environment lookups use the certified fast path (env_fast.hpp).  A lookup that cannot be certified
// (about one in a thousand) does not stall the lane: the sample's pending term {atten*energy, factor,
// direction} goes into a fix-up record, the lane walks the rest of the sample as usual, and the tiny
// k_env_fixup launch that follows evaluates the exact binary64 lookup and finishes the arithmetic in
// the reference's order.  Every sample is accumulated exactly once, by one of the two kernels.
constexpr int kFixupDwords = 32;  // one record = 128 B: header[4] bv_before[3] n_pending[1] 2 x {P[3] factor dir[3]}

// ------------------------------------------------------------------------------------------------
// Exit certificates.  A march that ends in Exit_volume contributes through its DIRECTION only (ray_marching.cl:54-62
// samples the environment with current_ray.direction): where it leaves the volume is never used.  So when it can be
// PROVEN that a march will leave the volume without a Hit within the steps it has left, its remaining steps -- far-field
// fetches, one 128-byte line each, for a position nobody needs -- are skipped and the Exit event is raised at once; the
// result is bit-identical.  The proof is one table lookup: the ray's coordinates are monotone, so the rest of its path
// lies in the box between its macro cell (16^3 voxels) and the volume corner its direction octant heads for, and the
// table (k_macro_table .. k_macro_bounds) holds, per cell and octant, the smallest SDF value of that box if it is
// free: no voxel that could be an event (a Hit needs one) and SDF values of at least kCertMinStep.  The ray's distance to the face it
// leaves through, divided by that minimum, bounds the steps the march still takes; the bound must fit the march's budget (a march that ran out of steps would continue as the NEXT march,
// with another weight, ray_marching.cl:52-73).  tools/exit_certificate.py measured the idea on the oracle first: every
// exiting ray gets its certificate at some point, 5 of the 30 step fetches per item disappear (all far field), and not
// one certificate in millions was wrong.
__device__ __forceinline__ bool certify_exit(const RenderArgs &a, f3 p, f3 d, int budget) {
  // the position has a voxel or sits on the far face: 0 <= p <= dim (or -0.0)
  const unsigned cx = min((unsigned)(int)p.x >> a.macro_shift, (unsigned)a.MNX - 1u), cy = min((unsigned)(int)p.y >> a.macro_shift, (unsigned)a.MNY - 1u),
                 cz = min((unsigned)(int)p.z >> a.macro_shift, (unsigned)a.MNZ - 1u);
  const unsigned octant = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
  const int box_min = a.macro[(((cz * (unsigned)a.MNY + cy) * (unsigned)a.MNX + cx) << 3) | octant];  // (at most 2^22 entries)
  // One kind of position is outside the reasoning below: a coordinate that landed exactly ON the far face (== dimension: not exited,
  // utility_ray.cl:112-117) reads the border SDF 0 and advances 0.5 |d| per step; with a direction component too small to move that
  // coordinate (0.5 x 2^-10 is above half an ulp of every dimension below 2^13) the reference can crawl along the face and even run
  // out of its 70 steps.  Such directions get no certificate and march literally; for all others the next step leaves (the 5).
  const float dmin = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
  const float dsum = d.x + d.y + d.z;  // NaN direction: the position turns NaN and never leaves
  // Every step inside the box is max(sdf, 0.5) >= box_min long and the direction has unit length, so after (budget - 5) steps the march has
  // travelled T = (budget - 5) * box_min along the ray (an integer below 2^14: exact) and has passed the face of an axis as soon as
  // T * |d_axis| >= its distance to that face -- one axis is enough; the 5 steps kept back cover the roundings of the march, of these three
  // products, and the strictness of exited_volume.  (Round 3 first took the smallest (face - p) / d over the axes: three reciprocals,
  // quarter-rate instructions, for the same decision.)
  const float T = (float)((budget - 5) * box_min);
  const float fx = d.x < 0.0f ? p.x : (float)a.X - p.x, fy = d.y < 0.0f ? p.y : (float)a.Y - p.y, fz = d.z < 0.0f ? p.z : (float)a.Z - p.z;
  const bool leaves = T * fabsf(d.x) >= fx || T * fabsf(d.y) >= fy || T * fabsf(d.z) >= fz;
  return box_min != 0 && budget > 5 && leaves && dmin >= 0.0009765625f && dsum == dsum;
}

#ifndef CLVR_BOUNCE_WAVES_PER_SIMD
// Waves per SIMD.  With the event-only state in LDS (`cold` below) every variant of the kernel needs 64 registers or fewer, so
// the register file allows eight; LDS (22 KB per 256-thread block at 512^3: the cold state, the index tables, div255) allows seven
// blocks per CU.  Measured on the 64-pass launch, 6 / 7 / 8 (8 with 512-thread blocks) -> 4.27 / 4.14 / 4.22 ms
// (profiles/r02_sweep_k_bounce_lds_state.txt); before that change the kernel held 80 registers + 9 spilled and ran six.
#define CLVR_BOUNCE_WAVES_PER_SIMD 7
#endif
#ifndef CLVR_BOUNCE_THREADS
#define CLVR_BOUNCE_THREADS 256
#endif
constexpr int kBounceThreads = CLVR_BOUNCE_THREADS;  // the waves of a block share the LDS index tables
template <bool USE_GRAD, int MODE, bool SMALL_VOLUME>
__global__ __launch_bounds__(kBounceThreads, CLVR_BOUNCE_WAVES_PER_SIMD) void k_bounce(const RenderArgs a) {
  constexpr int SMALL = SMALL_VOLUME ? 2 : 0;  // 2: the per-axis index terms are read from LDS (packed_volume.hpp)
  VolumePacked vol = make_volume(a);
  extern __shared__ uint32_t lds_parts[];
  if (SMALL == 2) {
    vol.parts = lds_parts;
    vol.parts_y0 = a.X;
    vol.parts_z0 = a.X + a.Y;
    for (int k = (int)threadIdx.x; k < a.X + a.Y + a.Z; k += kBounceThreads)
      lds_parts[k] = k < a.X ? vol.part_x<1>((unsigned)k) : (k < a.X + a.Y ? vol.part_y<1>((unsigned)(k - a.X)) : vol.part_z<1>((unsigned)(k - a.X - a.Y)));
  }
  // c / 255.0f for the 256 possible colour bytes (a correctly rounded division is ~10 VALU instructions, the
  // kernel is VALU-issue bound, and every bounce needs four of them): one LDS read instead
  __shared__ float div255[256];
  if (threadIdx.x < 256u) div255[threadIdx.x] = (float)threadIdx.x / 255.0f;
  __syncthreads();
  // counters: [0] hits, [2] fix-up records, [32 * (q + 1)] head of unit queue q (the overflow flag lives in sticky_flags)
  // the hit count of this camera: a kernel argument, or still only on the device (single-pass launches, clwh_render)
  const uint32_t n_hits = a.n_hits_on_device ? a.counters[0] : a.n_hits;
  const uint32_t n_chunks = (n_hits + 63u) >> 6;
  const unsigned lane = lane_id();
  // HW_REG_XCC_ID (id 20), bits [3:0]: the XCD this wave runs on
  unsigned home_queue = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
  if (a.unit_affinity == 1) home_queue = (blockIdx.x * (unsigned)(kBounceThreads / 64) + (threadIdx.x >> 6)) & 7u;
  if (a.unit_affinity == 2) home_queue = 0u;
  unsigned queue_dry = 0u;  // bit q: queue q is known to be empty (lane 0's copy is the one that matters)
  // The sample.  What only the event phase needs lives in LDS, one dword per lane and field (a conflict-free
  // ds_read/ds_write each): the registers decide how many waves a SIMD holds, and with one dependent fetch per step it
  // is the number of waves in flight that sets the pace.
  enum : int { C_START_X, C_START_Y, C_START_Z,  // hit origin + hit direction: where both distribution rays start from
               C_NORMAL_X, C_NORMAL_Y, C_NORMAL_Z,
               C_ENTRY_LO, C_ENTRY_HI, C_PIXEL, C_HIT, C_SEED,
               C_FIX,  // (fix + 2) << 2 | npend; fix: fix-up record of the sample (-1: none, -2: dropped, the buffer overflowed),
                       // npend: pending environment terms written to it
               C_BV_R, C_BV_G, C_BV_B, C_FIELDS };
  __shared__ uint32_t cold[C_FIELDS][kBounceThreads];
#define COLD(f) cold[f][threadIdx.x]
  // path state; energies and colour carry over from distribution ray 1 into ray 2 (SURVEY "hard parts")
  Ray ray{{0, 0, 0}, {0, 0, 0}};
  float atten = 0.0f, r_energy = 0.0f, g_energy = 0.0f, b_energy = 0.0f;
  uint32_t color = 0u;
  int o = 0, i = 0;
  // scheduling state
  int st = ST_IDLE;        // ST_IDLE / ST_MARCH / ST_CERT / ST_EVENT + event
  int sd = 0;              // SDF value for the next step of a MARCH lane
  int steps_left = 0;
  const int cert_min_lanes = a.cert_min_lanes;
  // a lane asks for an exit certificate when its next step is at least this long (wave-uniform)
  const int cert_at = a.cert_min_step != 0 ? a.cert_min_step : kCertNever;
  bool exhausted = false;  // wave-uniform: the queue has no more items
#ifdef CLVR_BOUNCE_STATS
  uint32_t st_step_iters = 0, st_step_lanes = 0, st_event_phases = 0, st_event_lanes = 0, st_refills = 0, st_refill_lanes = 0;
  uint32_t st_ev_kind[4] = {0, 0, 0, 0};
  uint32_t st_cert_phases = 0, st_cert_lanes = 0, st_cert_granted = 0;
#endif

  for (;;) {
    // ---- refill: idle lanes pull consecutive items of the unit queues ---------------------------------
    // An item is (hit, seed); 64 consecutive items of a queue are one unit = (chunk of 64 consecutive hits,
    // seed).  Units are dealt to eight queues by chunk number; a wave serves the queue of the XCD it runs on
    // first (so the seeds of one chunk -- thousands of rays leaving the same few voxels -- meet in ONE L2)
    // and steals from the other queues when its own is dry.  As soon as `refill_min_lanes` lanes are idle
    // they take the next items in queue order (one atomic per refill), so a wave does not drain down to its
    // slowest sample before it gets new work.  Placement only affects speed.
    const unsigned long long idle_mask = __ballot(st == ST_IDLE);
    const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
    if (!exhausted && n_idle >= (uint32_t)a.refill_min_lanes) {
      const uint32_t NQ = (uint32_t)a.unit_queues, S = (uint32_t)a.n_seeds, G = (uint32_t)a.unit_group;
      const uint32_t KB = (uint32_t)a.unit_block_log2, n_blocks = (n_chunks + (1u << KB) - 1u) >> KB;
      uint32_t base = 0u, count = 0u, q_sel = 0u;
      if (lane == 0u) {
        for (uint32_t tries = 0; tries < NQ && count == 0u; ++tries) {
          const uint32_t q = (home_queue + tries) % NQ;
          // blocks of 2^unit_block_log2 consecutive chunks are dealt round-robin to the queues (a block past the
          // last chunk is padding: its items name hits that do not exist and are skipped)
          const uint32_t blocks_q = (n_blocks + NQ - 1u - q) / NQ;
          const uint32_t chunks_q = blocks_q << KB;
          if (chunks_q == 0u || ((queue_dry >> q) & 1u)) continue;
          const uint32_t total = chunks_q * S * 64u;
          // every head sits on its own 128-byte line: same-address atomics serialise at one L2 channel
          const uint32_t p = atomicAdd(&a.counters[32u * (q + 1u)], n_idle);
          if (p + n_idle >= total) queue_dry |= 1u << q;  // remembered: never asked again
          if (p < total) {
            base = p;
            count = min(n_idle, total - p);
            q_sel = q;
          }
        }
      }
      base = __shfl(base, 0);
      count = __shfl(count, 0);
      q_sel = __shfl(q_sel, 0);
#ifdef CLVR_BOUNCE_STATS
      st_refills += 1; st_refill_lanes += count;
#endif
      if (count == 0u) {
        exhausted = true;  // every queue is dry
      } else if (st == ST_IDLE) {
        const uint32_t rank = (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
        if (rank < count) {
          const uint32_t item = base + rank, p = item >> 6;
          // queue order: groups of `unit_group` chunks, inside a group seed-major -- the seeds of a chunk are
          // `unit_group` units apart (their accumulation atomics do not collide) yet close enough to find
          // each other's voxels still in L2
          // (units per queue stay below 2^24: launch_bounce checks)
          const uint32_t chunks_q = ((n_blocks + NQ - 1u - q_sel) / NQ) << KB;  // wave-uniform: scalar
          uint32_t r, c_in;
          const uint32_t g = udivmod24(p, G * S, r);
          const uint32_t in_group = min(G, chunks_q - g * G);  // the last group may be short
          const uint32_t s = udivmod24(r, in_group, c_in), ch = g * G + c_in;
          const uint32_t chunk = ((q_sel + NQ * (ch >> KB)) << KB) + (ch & ((1u << KB) - 1u));
          const uint32_t h = chunk * 64u + (item & 63u);
          if (h < n_hits) {
            const uint4 *src = reinterpret_cast<const uint4 *>(&a.hits[h]);
            const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
            const f3 hit_origin = f3{__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z)};
            const f3 hit_direction = f3{__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y)};
            const f3 start = hit_origin + hit_direction;  // ray_bounce_fake_reflectance's origin (utility_ray.cl:100-103)
            color = q2.y;
            const int64_t entry = (int64_t)(((uint64_t)q2.w << 32) | (uint64_t)q2.z);
            bool granted = true;
            if (MODE == CLWH_ACCUM_VOXEL_CACHE)
              granted = a.grants ? s < a.grants[h] : (entry >= 0 && cache_take_token(a.cache, entry, 256u));
            if (granted) {
              COLD(C_START_X) = __float_as_uint(start.x); COLD(C_START_Y) = __float_as_uint(start.y); COLD(C_START_Z) = __float_as_uint(start.z);
              COLD(C_NORMAL_X) = q1.z; COLD(C_NORMAL_Y) = q1.w; COLD(C_NORMAL_Z) = q2.x;
              COLD(C_ENTRY_LO) = q2.z; COLD(C_ENTRY_HI) = q2.w;
              COLD(C_PIXEL) = q3.x;
              COLD(C_HIT) = h;
              COLD(C_SEED) = (uint32_t)a.seeds[s];
              COLD(C_FIX) = (uint32_t)(-1 + 2) << 2;
              COLD(C_BV_R) = 0u; COLD(C_BV_G) = 0u; COLD(C_BV_B) = 0u;
              r_energy = div255[color & 255u];
              g_energy = div255[(color >> 8) & 255u];
              b_energy = div255[(color >> 16) & 255u];
              o = 1;
              st = ST_EVENT + EV_START;
            } else if (a.contrib_out) {
              uint32_t *q = a.contrib_out + ((size_t)(q3.x >> 16) * (size_t)a.launch_w + (q3.x & 0xFFFFu)) * 4;
              q[0] = 0u; q[1] = 0u; q[2] = 0u; q[3] = 0u;
            }
          }
        }
      }
    }
    if (__ballot(st != ST_IDLE) == 0ull) {
#ifdef CLVR_BOUNCE_STATS
      if (exhausted && lane == 0u) {
        atomicAdd(&a.counters[8], st_step_iters); atomicAdd(&a.counters[9], st_step_lanes);
        atomicAdd(&a.counters[10], st_event_phases); atomicAdd(&a.counters[11], st_event_lanes);
        atomicAdd(&a.counters[12], st_refills); atomicAdd(&a.counters[13], st_refill_lanes);
        for (int k = 0; k < 4; ++k) atomicAdd(&a.counters[14 + k], st_ev_kind[k]);
        atomicAdd(&a.counters[18], st_cert_phases); atomicAdd(&a.counters[19], st_cert_lanes); atomicAdd(&a.counters[20], st_cert_granted);
      }
#endif
      if (exhausted) break;  // nothing in flight and nothing left to fetch
      continue;              // every fetched sample was refused its token (or was padding): fetch again
    }

    // ---- step phase: MARCH lanes step until fewer than kStepPhaseMinLanes are still marching -----
    if (__ballot(st == ST_MARCH) != 0ull) {
      do {
#ifdef CLVR_BOUNCE_STATS
        st_step_iters += 1; st_step_lanes += (uint32_t)__popcll(__ballot(st == ST_MARCH));
#endif
#ifdef CLVR_EXP_STEP_PAD  // experiment: N extra dependent FMAs per step iteration
        {
          float pad = ray.origin.x;
#pragma unroll
          for (int q = 0; q < CLVR_EXP_STEP_PAD; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(pad));
          if (pad == 123.456f) atten = pad;
        }
#endif
        if (st == ST_MARCH) {
          // sd is an integer in 0..127: fmaxf is cl_max here
          ray.origin = ray.origin + ray.direction * fmaxf((float)sd, 0.5f);
          --steps_left;
          if (!USE_GRAD) {
            // Has the new position a voxel?  +0 <= coordinate < dimension is ONE unsigned compare of the float's bits per axis
            // (negative values, -0.0 and NaN all have larger patterns than any dimension).  Whatever fails it -- nearly always
            // a ray that left the volume; else the far face, NaN or -0.0 -- is sorted out exactly in the event phase (EV_CHECK),
            // at the event phase's price instead of eight more compares in every iteration of this loop.
            const bool has_voxel = __float_as_uint(ray.origin.x) < __float_as_uint((float)a.X) &&
                                   __float_as_uint(ray.origin.y) < __float_as_uint((float)a.Y) &&
                                   __float_as_uint(ray.origin.z) < __float_as_uint((float)a.Z);
            if (has_voxel) {
              // the step byte says everything a table without `gradient` rules needs (classify_step's last case)
#ifdef CLVR_EXP_FAR_SC1  // experiment: the fetch after a long step (a new line, rarely used again) does not allocate in L1
              const unsigned q = sd >= CLVR_EXP_FAR_SC1 ? vol.template step_marched_past_l1<SMALL>(ray.origin.x, ray.origin.y, ray.origin.z)
                                                        : vol.template step_marched<SMALL>(ray.origin.x, ray.origin.y, ray.origin.z);
#else
              const unsigned q = vol.template step_marched<SMALL>(ray.origin.x, ray.origin.y, ray.origin.z);
#endif
#ifdef CLVR_EXP_STEP_LOADPAD  // experiment: N more loads per step, of neighbours inside the line just requested (address unit / L1 sensitivity)
              {
                unsigned pad_acc = 0u;
#pragma unroll
                for (int k = 1; k <= CLVR_EXP_STEP_LOADPAD; ++k)
                  pad_acc += vol.template step_marched<SMALL>((float)(f2i(ray.origin.x) ^ k), ray.origin.y, ray.origin.z);
                if (pad_acc == 255u * CLVR_EXP_STEP_LOADPAD) atten = 1.0f;  // (possible in principle: a timing experiment, not a product build)
              }
#endif
#ifdef CLVR_EXP_STEP_L2PAD  // experiment: N more loads per step from random lines of the first MiB of the step bytes (L2 hits, L1 misses)
              {
                unsigned pad_acc = 0u, hsh = __float_as_uint(ray.origin.x) * 2654435761u + __float_as_uint(ray.origin.y) * 40503u + __float_as_uint(ray.origin.z);
#pragma unroll
                for (int k = 0; k < CLVR_EXP_STEP_L2PAD; ++k) {
                  hsh ^= hsh << 13; hsh ^= hsh >> 17; hsh ^= hsh << 5;
                  pad_acc += vol.stepb[hsh & 0xFFFFFu];
                }
                if (pad_acc == 255u * CLVR_EXP_STEP_L2PAD) atten = 1.0f;
              }
#endif
              sd = (int)(q & 0x7Fu);
              if (q & 0x80u) st = ST_EVENT + EV_HIT_COLOR_PENDING;
              else if (steps_left == 0) st = ST_EVENT + EV_NONE;
              else if (sd >= cert_at) st = ST_CERT;  // far from every surface: can the rest of this march be proven to exit?
            } else {
              st = ST_EVENT + EV_CHECK;
            }
          } else if (exited_volume(vol, ray.origin)) {
            st = ST_EVENT + EV_EXIT;
          } else {
            int next_sd;
            bool pending = false;
            const bool is_hit = classify_step<USE_GRAD, SMALL, true>(vol, a.tf, ray.origin, color, next_sd, &pending);
            if (is_hit) {
              st = ST_EVENT + (pending ? EV_HIT_COLOR_PENDING : EV_HIT);
            } else if (steps_left == 0) {
              st = ST_EVENT + EV_NONE;
            } else {
              sd = next_sd;
              if (sd >= cert_at) st = ST_CERT;
            }
          }
        }
        // ---- exit certificates, inside the step loop: lanes park until enough of them wait (or nobody marches), one table
        // lookup each (certify_exit), and go back to marching or on to their Exit event; the loop, and with it the lane
        // count of the event phase, is the same as without certificates
        const int n_cert = __popcll(__ballot(st == ST_CERT));
        if (n_cert != 0 && (n_cert >= cert_min_lanes || __popcll(__ballot(st == ST_MARCH)) < a.step_min_lanes)) {
#ifdef CLVR_BOUNCE_STATS
          st_cert_phases += 1; st_cert_lanes += (uint32_t)n_cert;
          const int marching_before = __popcll(__ballot(st == ST_MARCH));
#endif
          if (st == ST_CERT) {
            if (certify_exit(a, ray.origin, ray.direction, steps_left)) {
              st = ST_EVENT + EV_EXIT;  // the march WOULD end in Exit_volume; only its direction matters from here on
            } else {
              // it tries again at its next step that is long enough (trying less often -- only once the step has doubled, or grown by
              // half -- left more lines to fetch than the look-ups cost: 4.03 / 3.92 / 3.91 ms)
              st = ST_MARCH;
            }
          }
#ifdef CLVR_BOUNCE_STATS
          st_cert_granted += (uint32_t)(n_cert - (__popcll(__ballot(st == ST_MARCH)) - marching_before));
#endif
        }
      } while (__popcll(__ballot(st == ST_MARCH)) >= a.step_min_lanes);
    }

    // ---- event phase: every parked lane handles its event; the bounce is one shared block ---------
#ifdef CLVR_EXP_EVENT_PAD  // experiment: how sensitive is the launch to VALU work in the event phase?  N extra dependent FMAs per phase
    {
      float pad = ray.origin.x;
#pragma unroll
      for (int q = 0; q < CLVR_EXP_EVENT_PAD; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(pad));
      if (pad == 123.456f) atten = pad;
    }
#endif
#ifdef CLVR_BOUNCE_STATS
    st_event_phases += 1; st_event_lanes += (uint32_t)__popcll(__ballot(st >= ST_EVENT));
    st_ev_kind[0] += (uint32_t)__popcll(__ballot(st == ST_EVENT + EV_START));
    st_ev_kind[1] += (uint32_t)__popcll(__ballot(st == ST_EVENT + EV_EXIT || st == ST_EVENT + EV_CHECK));
    st_ev_kind[2] += (uint32_t)__popcll(__ballot(st == ST_EVENT + EV_HIT || st == ST_EVENT + EV_HIT_COLOR_PENDING));
    st_ev_kind[3] += (uint32_t)__popcll(__ballot(st == ST_EVENT + EV_NONE));
#endif
    if (st >= ST_EVENT) {
      int ev = st - ST_EVENT;
      if (ev == EV_CHECK) {
        // The step loop's quick test found no voxel at the new position: the reference's own tests, in its order
        // (utility_ray.cl:157-168: exited? event? out of steps?).  Nearly always the ray has left the volume.  The rare march
        // that goes on -- a position exactly on the far face, NaN, -0.0 -- is finished right here, step by literal step: sent
        // back to the step loop it would come here again after every step (a NaN position has no voxel ever), and in a short
        // launch the seventy event phases of one such ray were the tail of the whole launch (0.30 -> 0.44 ms per pass).
        for (;;) {
          if (exited_volume(vol, ray.origin)) { ev = EV_EXIT; break; }
          int next_sd;
          bool pending = false;
          if (classify_step<USE_GRAD, SMALL, true>(vol, a.tf, ray.origin, color, next_sd, &pending)) {
            ev = pending ? EV_HIT_COLOR_PENDING : EV_HIT;
            break;
          }
          if (steps_left == 0) { ev = EV_NONE; break; }
          ray.origin = ray.origin + ray.direction * fmaxf((float)next_sd, 0.5f);
          --steps_left;
        }
      }
      bool start_path = (ev == EV_START);  // begin distribution ray `o` from the primary hit
      bool bounce = false, from_hit = false;
      f3 bn{0, 0, 0}, bstart{0, 0, 0};  // the bounce's normal and its origin + direction
      int bseed = 0;                    // its seed, less the sample's

      if (ev == EV_EXIT) {
        // ray_marching.cl:54-62: left the volume -> environment light ends this distribution ray
        // i is 8, 9 or 10 here (it starts at 8 and a path ends once it exceeds 10); the quotients are folded at compile time
        const float factor = i == 8 ? 8.0f / 8.0f : (i == 9 ? 8.0f / 9.0f : (i == 10 ? 8.0f / 10.0f : 8.0f / (float)i));
        const float p_r = atten * r_energy, p_g = atten * g_energy, p_b = atten * b_energy;
        uint32_t light = 0u;
        const uint32_t fix_word = COLD(C_FIX);
        int fix = (int)(fix_word >> 2) - 2, npend = (int)(fix_word & 3u);
        bool certain = (fix == -1) && sample_environment_map_fast(a.env, a.env_w, a.env_h, ray.direction, light);
        if (certain) {
          // uint += float: promote, add, truncate back
          COLD(C_BV_R) = f2u((float)COLD(C_BV_R) + p_r * (float)(light & 255u) * factor / 1.0f);
          COLD(C_BV_G) = f2u((float)COLD(C_BV_G) + p_g * (float)((light >> 8) & 255u) * factor / 1.0f);
          COLD(C_BV_B) = f2u((float)COLD(C_BV_B) + p_b * (float)((light >> 16) & 255u) * factor / 1.0f);
        } else {
          if (fix == -1) {
            // first undecided lookup of this sample: open a fix-up record
            const uint32_t slot = atomicAdd(&a.counters[2], 1u);
            if (slot < a.fixup_capacity) {
              fix = (int)slot;
              uint32_t *rec = a.fixups + (size_t)slot * kFixupDwords;
              rec[0] = COLD(C_HIT);
              rec[1] = COLD(C_ENTRY_LO);
              rec[2] = COLD(C_ENTRY_HI);
              rec[3] = COLD(C_PIXEL);
              rec[4] = COLD(C_BV_R); rec[5] = COLD(C_BV_G); rec[6] = COLD(C_BV_B);
            } else {
              a.sticky_flags[0] = 1u;  // reported by the host as an error; the sample is dropped
              fix = -2;
            }
          }
          if (fix >= 0) {
            uint32_t *e = a.fixups + (size_t)fix * kFixupDwords + 8 + 7 * npend;
            e[0] = __float_as_uint(p_r); e[1] = __float_as_uint(p_g); e[2] = __float_as_uint(p_b);
            e[3] = __float_as_uint(factor);
            e[4] = __float_as_uint(ray.direction.x); e[5] = __float_as_uint(ray.direction.y);
            e[6] = __float_as_uint(ray.direction.z);
            npend += 1;
          }
          COLD(C_FIX) = ((uint32_t)(fix + 2) << 2) | (uint32_t)npend;
        }
        o += 1;
        start_path = true;
      } else if (ev == EV_HIT || ev == EV_HIT_COLOR_PENDING) {
        // ray_marching.cl:63-72: secondary hit -> bounce around the local normal, attenuate; the rule colour (still
        // pending when the Hit came through the step byte) and the gradient arrive in one 8-byte load
        bn = -normalize3(hit_gradient_and_color<SMALL>(vol, a.tf, ray.origin, ev == EV_HIT_COLOR_PENDING, color));
        bstart = ray.origin + ray.direction;
        bseed = o + i;
        bounce = true;
        from_hit = true;
        i += 1;
        if (i > 10) {
          // third march of this distribution ray: the reference still multiplies the energies (they
          // carry into the next distribution ray) but its bounced ray is never marched
          r_energy *= div255[color & 255u];
          g_energy *= div255[(color >> 8) & 255u];
          b_energy *= div255[(color >> 16) & 255u];
          bounce = false;
          o += 1;
          start_path = true;
        }
      } else if (ev == EV_NONE) {
        // 70 steps without an event: the next march continues from where this one stopped
        i += 1;
        if (i > 10) {
          o += 1;
          start_path = true;
        }
      }

      if (start_path) {
        if (o > 2) {
          const int fix = (int)(COLD(C_FIX) >> 2) - 2;
          if (fix == -1)
            finish_item<MODE>(a, (int64_t)(((uint64_t)COLD(C_ENTRY_HI) << 32) | (uint64_t)COLD(C_ENTRY_LO)), COLD(C_HIT), COLD(C_PIXEL) & 0xFFFFu,
                              COLD(C_PIXEL) >> 16, COLD(C_BV_R), COLD(C_BV_G), COLD(C_BV_B));
          else if (fix >= 0) a.fixups[(size_t)fix * kFixupDwords + 7] = COLD(C_FIX) & 3u;  // k_env_fixup finishes it
          st = ST_IDLE;
        } else {
          // ray_marching.cl:48: bounce from the primary hit around the primary normal
          bn = f3{__uint_as_float(COLD(C_NORMAL_X)), __uint_as_float(COLD(C_NORMAL_Y)), __uint_as_float(COLD(C_NORMAL_Z))};
          bstart = f3{__uint_as_float(COLD(C_START_X)), __uint_as_float(COLD(C_START_Y)), __uint_as_float(COLD(C_START_Z))};
          bseed = o;
          bounce = true;
          from_hit = false;
        }
      }

      if (bounce) {
        // ray_bounce_fake_reflectance, then origin += normal*2 (ray_marching.cl:48-50 / :65-67)
        const float roughness = div255[color >> 24];
        const uint32_t pixel = COLD(C_PIXEL);
        Ray nr;
        nr.direction = hemisphere_reflective(pixel & 0xFFFFu, pixel >> 16, bn, (int)COLD(C_SEED) + bseed, roughness);
        nr.origin = bstart + bn * 2.0f;
        const float d = fabsf(dot3(nr.direction, bn));
        if (from_hit) {
          atten *= d;
          r_energy *= div255[color & 255u];
          g_energy *= div255[(color >> 8) & 255u];
          b_energy *= div255[(color >> 16) & 255u];
        } else {
          atten = d;
          i = 8;
        }
        ray = nr;
      }

      if (st >= ST_EVENT) {
        // start (or continue) a march: its first SDF read is at trunc(origin) (utility_ray.cl:148-150)
        sd = (int)(vol.template step_i<SMALL>(f2i(ray.origin.x), f2i(ray.origin.y), f2i(ray.origin.z)) & 0x7Fu);
        steps_left = 70;
        st = ST_MARCH;
      }
    }
  }
#undef COLD
}

// ------------------------------------------------------------------------------------------------
// k_bounce2: the same pass with TWO rays per lane (CLWH_TUNE_BOUNCE_RAYS=2; long launches only) -- the round-3 experiment on lane
// utilisation (results in profiles/r03_k_bounce_two_rays_per_lane.txt).
//
// k_bounce runs its step iterations at 28.8 and its event phases at 36.7 of 64 lanes: a lane whose ray waits for the wave's event
// phase takes no march steps, a lane whose ray marches sits out the event phase.  Here a lane owns two rays, `cur` and `alt`: the hot
// march state of both in registers (origin, direction; state, step length and steps left -- alt's packed into one register), everything
// only an event needs in LDS (nine dwords per ray: pixel, hit | seed index, fix-up word with o and i, the packed radiance sums,
// attenuation, three energies, colour; the primary hit's normal and start are re-read from its 64-byte record).  The step loop works on
// `cur`; where it ends a lane whose cur is parked swaps in a marching alt (seven v_swap) and the loop runs once more; before the event
// phase a lane whose cur still marches swaps in a parked or idle alt -- so both phases see a ray of the right kind in most lanes.  No
// ray ever leaves its lane: no synchronisation between waves, the per-ray arithmetic is k_bounce's, instruction for instruction.
constexpr int kBounce2WavesPerSimd = 6;  // LDS: 18 KB of event state + 6 KB index tables + 1.3 KB per 256-thread block
template <bool USE_GRAD, int MODE, bool SMALL_VOLUME>
__global__ __launch_bounds__(256, kBounce2WavesPerSimd) void k_bounce2(const RenderArgs a) {
  constexpr int SMALL = SMALL_VOLUME ? 2 : 0;
  VolumePacked vol = make_volume(a);
  extern __shared__ uint32_t lds_parts[];
  if (SMALL == 2) {
    vol.parts = lds_parts;
    vol.parts_y0 = a.X;
    vol.parts_z0 = a.X + a.Y;
    for (int k = (int)threadIdx.x; k < a.X + a.Y + a.Z; k += 256)
      lds_parts[k] = k < a.X ? vol.part_x<1>((unsigned)k) : (k < a.X + a.Y ? vol.part_y<1>((unsigned)(k - a.X)) : vol.part_z<1>((unsigned)(k - a.X - a.Y)));
  }
  __shared__ float div255[256];
  __shared__ int32_t s_seeds[CLWH_MAX_SEEDS];
  div255[threadIdx.x] = (float)threadIdx.x / 255.0f;
  if (threadIdx.x < (unsigned)CLWH_MAX_SEEDS) s_seeds[threadIdx.x] = a.seeds[threadIdx.x < (unsigned)a.n_seeds ? threadIdx.x : 0u];
  enum : int { C_PIXEL, C_HITSEED,  // x | y << 16 ; hit | seed index << 26
               C_FIX,               // npend | (o - 1) << 2 | (i - 8) << 3 | (fix + 2) << 5
               C_BV,                // the sample's radiance sums so far, r | g << 10 | b << 20: two exits of at most 255 each per channel
               C_ATTEN, C_ER, C_EG, C_EB, C_COLOR, C_FIELDS };
  __shared__ uint32_t cold[C_FIELDS][2][256];
  __syncthreads();
  uint32_t cs = 0u;  // which of the lane's two event-state slots belongs to `cur`
#define COLD(f) cold[f][cs][threadIdx.x]
  const uint32_t n_hits = a.n_hits_on_device ? a.counters[0] : a.n_hits;
  const uint32_t n_chunks = (n_hits + 63u) >> 6;
  const unsigned lane = lane_id();
  unsigned home_queue = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
  if (a.unit_affinity == 1) home_queue = (blockIdx.x * 4u + (threadIdx.x >> 6)) & 7u;
  if (a.unit_affinity == 2) home_queue = 0u;
  unsigned queue_dry = 0u;
  // cur: the ray the step loop and the event phase work on; alt: the lane's other ray
  Ray ray{{0, 0, 0}, {0, 0, 0}}, aray{{0, 0, 0}, {0, 0, 0}};
  int st = ST_IDLE, sd = 0, steps_left = 0;
  uint32_t ameta = (uint32_t)ST_IDLE;  // alt's st | sd << 8 | steps_left << 16
  const int cert_min_lanes = a.cert_min_lanes;
  const int cert_at = a.cert_min_step != 0 ? a.cert_min_step : kCertNever;
  bool exhausted = false;
#ifdef CLVR_BOUNCE_STATS
  uint32_t st_step_iters = 0, st_step_lanes = 0, st_event_phases = 0, st_event_lanes = 0, st_refills = 0, st_refill_lanes = 0, st_swaps = 0;
#endif
  auto swap_rays = [&](bool doit) {
    if (doit) {
      float t;
      t = ray.origin.x; ray.origin.x = aray.origin.x; aray.origin.x = t;
      t = ray.origin.y; ray.origin.y = aray.origin.y; aray.origin.y = t;
      t = ray.origin.z; ray.origin.z = aray.origin.z; aray.origin.z = t;
      t = ray.direction.x; ray.direction.x = aray.direction.x; aray.direction.x = t;
      t = ray.direction.y; ray.direction.y = aray.direction.y; aray.direction.y = t;
      t = ray.direction.z; ray.direction.z = aray.direction.z; aray.direction.z = t;
      const uint32_t m = (uint32_t)st | ((uint32_t)sd << 8) | ((uint32_t)steps_left << 16);
      st = (int)(ameta & 255u); sd = (int)((ameta >> 8) & 255u); steps_left = (int)(ameta >> 16);
      ameta = m;
      cs ^= 1u;
    }
  };

  for (;;) {
    // ---- present what the event phase and the refill can work on in `cur` ------------------------------
    bool will_refill;
    {
      const int ast = (int)(ameta & 255u);
      const bool cur_busy = st == ST_MARCH || st == ST_CERT, alt_ev = ast >= ST_EVENT, alt_idle = ast == ST_IDLE;
      will_refill = !exhausted && (uint32_t)__popcll(__ballot(st == ST_IDLE || alt_idle)) >= (uint32_t)a.refill_min_lanes;
      swap_rays((cur_busy && alt_ev) || (st == ST_IDLE && alt_ev) || (will_refill && cur_busy && alt_idle));
#ifdef CLVR_BOUNCE_STATS
      st_swaps += 1;
#endif
    }
    // ---- refill: lanes whose cur is idle pull consecutive items of the unit queues (k_bounce's scheme) ---
    const unsigned long long idle_mask = __ballot(st == ST_IDLE);
    const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
    if (will_refill && n_idle != 0u) {
      const uint32_t NQ = (uint32_t)a.unit_queues, S = (uint32_t)a.n_seeds, G = (uint32_t)a.unit_group;
      const uint32_t KB = (uint32_t)a.unit_block_log2, n_blocks = (n_chunks + (1u << KB) - 1u) >> KB;
      uint32_t base = 0u, count = 0u, q_sel = 0u;
      if (lane == 0u) {
        for (uint32_t tries = 0; tries < NQ && count == 0u; ++tries) {
          const uint32_t q = (home_queue + tries) % NQ;
          const uint32_t blocks_q = (n_blocks + NQ - 1u - q) / NQ;
          const uint32_t chunks_q = blocks_q << KB;
          if (chunks_q == 0u || ((queue_dry >> q) & 1u)) continue;
          const uint32_t total = chunks_q * S * 64u;
          const uint32_t p = atomicAdd(&a.counters[32u * (q + 1u)], n_idle);
          if (p + n_idle >= total) queue_dry |= 1u << q;
          if (p < total) {
            base = p;
            count = min(n_idle, total - p);
            q_sel = q;
          }
        }
      }
      base = __shfl(base, 0);
      count = __shfl(count, 0);
      q_sel = __shfl(q_sel, 0);
#ifdef CLVR_BOUNCE_STATS
      st_refills += 1; st_refill_lanes += count;
#endif
      if (count == 0u) {
        exhausted = true;
      } else if (st == ST_IDLE) {
        const uint32_t rank = (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
        if (rank < count) {
          const uint32_t item = base + rank, p = item >> 6;
          const uint32_t chunks_q = ((n_blocks + NQ - 1u - q_sel) / NQ) << KB;
          uint32_t r, c_in;
          const uint32_t g = udivmod24(p, G * S, r);
          const uint32_t in_group = min(G, chunks_q - g * G);
          const uint32_t s = udivmod24(r, in_group, c_in), ch = g * G + c_in;
          const uint32_t chunk = ((q_sel + NQ * (ch >> KB)) << KB) + (ch & ((1u << KB) - 1u));
          const uint32_t h = chunk * 64u + (item & 63u);
          if (h < n_hits) {
            const uint4 *src = reinterpret_cast<const uint4 *>(&a.hits[h]);
            const uint4 q2 = src[2], q3 = src[3];
            const uint32_t color = q2.y;
            const int64_t entry = (int64_t)(((uint64_t)q2.w << 32) | (uint64_t)q2.z);
            bool granted = true;
            if (MODE == CLWH_ACCUM_VOXEL_CACHE)
              granted = a.grants ? s < a.grants[h] : (entry >= 0 && cache_take_token(a.cache, entry, 256u));
            if (granted) {
              COLD(C_PIXEL) = q3.x;
              COLD(C_HITSEED) = h | (s << 26);
              COLD(C_FIX) = (uint32_t)(-1 + 2) << 5;  // no fix-up record, o = 1, i = 8, nothing pending
              COLD(C_BV) = 0u;
              COLD(C_ATTEN) = 0u;
              COLD(C_ER) = __float_as_uint(div255[color & 255u]);
              COLD(C_EG) = __float_as_uint(div255[(color >> 8) & 255u]);
              COLD(C_EB) = __float_as_uint(div255[(color >> 16) & 255u]);
              COLD(C_COLOR) = color;
              st = ST_EVENT + EV_START;
            } else if (a.contrib_out) {
              uint32_t *q = a.contrib_out + ((size_t)(q3.x >> 16) * (size_t)a.launch_w + (q3.x & 0xFFFFu)) * 4;
              q[0] = 0u; q[1] = 0u; q[2] = 0u; q[3] = 0u;
            }
          }
        }
      }
    }
    if (__ballot(st != ST_IDLE || (ameta & 255u) != (uint32_t)ST_IDLE) == 0ull) {
#ifdef CLVR_BOUNCE_STATS
      if (exhausted && lane == 0u) {
        atomicAdd(&a.counters[8], st_step_iters); atomicAdd(&a.counters[9], st_step_lanes);
        atomicAdd(&a.counters[10], st_event_phases); atomicAdd(&a.counters[11], st_event_lanes);
        atomicAdd(&a.counters[12], st_refills); atomicAdd(&a.counters[13], st_refill_lanes);
        atomicAdd(&a.counters[21], st_swaps);
      }
#endif
      if (exhausted) break;
      continue;
    }

    // ---- event phase on cur: k_bounce's, with the event-only state read from / written to the ray's LDS slot ----
#ifdef CLVR_BOUNCE_STATS
    if (__ballot(st >= ST_EVENT) != 0ull) { st_event_phases += 1; st_event_lanes += (uint32_t)__popcll(__ballot(st >= ST_EVENT)); }
#endif
    if (st >= ST_EVENT) {
      int ev = st - ST_EVENT;
      uint32_t fixw = COLD(C_FIX);
      int o = (int)((fixw >> 2) & 1u) + 1, i = (int)((fixw >> 3) & 3u) + 8;
      uint32_t color = COLD(C_COLOR);
      if (ev == EV_CHECK) {
        for (;;) {
          if (exited_volume(vol, ray.origin)) { ev = EV_EXIT; break; }
          int next_sd;
          bool pending = false;
          if (classify_step<USE_GRAD, SMALL, true>(vol, a.tf, ray.origin, color, next_sd, &pending)) {
            ev = pending ? EV_HIT_COLOR_PENDING : EV_HIT;
            break;
          }
          if (steps_left == 0) { ev = EV_NONE; break; }
          ray.origin = ray.origin + ray.direction * fmaxf((float)next_sd, 0.5f);
          --steps_left;
        }
      }
      bool start_path = (ev == EV_START);
      bool bounce = false, from_hit = false;
      f3 bn{0, 0, 0}, bstart{0, 0, 0};
      int bseed = 0;
      const uint32_t hitseed = COLD(C_HITSEED);
      const uint32_t h = hitseed & 0x03FFFFFFu;

      if (ev == EV_EXIT) {
        const float factor = i == 8 ? 8.0f / 8.0f : (i == 9 ? 8.0f / 9.0f : (i == 10 ? 8.0f / 10.0f : 8.0f / (float)i));
        const float atten = __uint_as_float(COLD(C_ATTEN));
        const float p_r = atten * __uint_as_float(COLD(C_ER)), p_g = atten * __uint_as_float(COLD(C_EG)), p_b = atten * __uint_as_float(COLD(C_EB));
        uint32_t light = 0u;
        int fix = (int)(fixw >> 5) - 2, npend = (int)(fixw & 3u);
        const bool certain = (fix == -1) && sample_environment_map_fast(a.env, a.env_w, a.env_h, ray.direction, light);
        const uint32_t bv = COLD(C_BV);
        if (certain) {
          // uint += float: promote, add, truncate back
          const uint32_t br = f2u((float)(bv & 1023u) + p_r * (float)(light & 255u) * factor / 1.0f);
          const uint32_t bg = f2u((float)((bv >> 10) & 1023u) + p_g * (float)((light >> 8) & 255u) * factor / 1.0f);
          const uint32_t bb = f2u((float)(bv >> 20) + p_b * (float)((light >> 16) & 255u) * factor / 1.0f);
          COLD(C_BV) = br | (bg << 10) | (bb << 20);
        } else {
          if (fix == -1) {
            const uint32_t slot = atomicAdd(&a.counters[2], 1u);
            if (slot < a.fixup_capacity) {
              fix = (int)slot;
              uint32_t *rec = a.fixups + (size_t)slot * kFixupDwords;
              const HitRec &hr = a.hits[h];
              rec[0] = h;
              rec[1] = (uint32_t)hr.entry_lo;
              rec[2] = (uint32_t)hr.entry_hi;
              rec[3] = COLD(C_PIXEL);
              rec[4] = bv & 1023u; rec[5] = (bv >> 10) & 1023u; rec[6] = bv >> 20;
            } else {
              a.sticky_flags[0] = 1u;
              fix = -2;
            }
          }
          if (fix >= 0) {
            uint32_t *e = a.fixups + (size_t)fix * kFixupDwords + 8 + 7 * npend;
            e[0] = __float_as_uint(p_r); e[1] = __float_as_uint(p_g); e[2] = __float_as_uint(p_b);
            e[3] = __float_as_uint(factor);
            e[4] = __float_as_uint(ray.direction.x); e[5] = __float_as_uint(ray.direction.y);
            e[6] = __float_as_uint(ray.direction.z);
            npend += 1;
          }
          fixw = (fixw & 0x1Cu) | (uint32_t)npend | ((uint32_t)(fix + 2) << 5);
        }
        o += 1;
        start_path = true;
      } else if (ev == EV_HIT || ev == EV_HIT_COLOR_PENDING) {
        bn = -normalize3(hit_gradient_and_color<SMALL>(vol, a.tf, ray.origin, ev == EV_HIT_COLOR_PENDING, color));
        bstart = ray.origin + ray.direction;
        bseed = o + i;
        bounce = true;
        from_hit = true;
        i += 1;
        if (i > 10) {
          COLD(C_ER) = __float_as_uint(__uint_as_float(COLD(C_ER)) * div255[color & 255u]);
          COLD(C_EG) = __float_as_uint(__uint_as_float(COLD(C_EG)) * div255[(color >> 8) & 255u]);
          COLD(C_EB) = __float_as_uint(__uint_as_float(COLD(C_EB)) * div255[(color >> 16) & 255u]);
          bounce = false;
          o += 1;
          start_path = true;
        }
      } else if (ev == EV_NONE) {
        i += 1;
        if (i > 10) {
          o += 1;
          start_path = true;
        }
      }

      if (start_path) {
        if (o > 2) {
          const int fix = (int)(fixw >> 5) - 2;
          if (fix == -1) {
            const uint32_t bv = COLD(C_BV), pixel = COLD(C_PIXEL);
            int64_t entry = 0;
            if (MODE == CLWH_ACCUM_VOXEL_CACHE && a.grants == nullptr) {
              const HitRec &hr = a.hits[h];
              entry = (int64_t)(((uint64_t)(uint32_t)hr.entry_hi << 32) | (uint64_t)(uint32_t)hr.entry_lo);
            }
            finish_item<MODE>(a, entry, h, pixel & 0xFFFFu, pixel >> 16, bv & 1023u, (bv >> 10) & 1023u, bv >> 20);
          } else if (fix >= 0) {
            a.fixups[(size_t)fix * kFixupDwords + 7] = fixw & 3u;  // k_env_fixup finishes it
          }
          st = ST_IDLE;
        } else {
          // ray_marching.cl:48: bounce from the primary hit around the primary normal (re-read from the hit's record)
          const uint4 *src = reinterpret_cast<const uint4 *>(&a.hits[h]);
          const uint4 q0 = src[0], q1 = src[1];
          const uint32_t nz = reinterpret_cast<const uint32_t *>(&a.hits[h])[8];
          const f3 hit_origin = f3{__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z)};
          const f3 hit_direction = f3{__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y)};
          bn = f3{__uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(nz)};
          bstart = hit_origin + hit_direction;
          bseed = o;
          bounce = true;
          from_hit = false;
        }
      }

      if (bounce) {
        const float roughness = div255[color >> 24];
        const uint32_t pixel = COLD(C_PIXEL);
        Ray nr;
        nr.direction = hemisphere_reflective(pixel & 0xFFFFu, pixel >> 16, bn, s_seeds[hitseed >> 26] + bseed, roughness);
        nr.origin = bstart + bn * 2.0f;
        const float d = fabsf(dot3(nr.direction, bn));
        if (from_hit) {
          COLD(C_ATTEN) = __float_as_uint(__uint_as_float(COLD(C_ATTEN)) * d);
          COLD(C_ER) = __float_as_uint(__uint_as_float(COLD(C_ER)) * div255[color & 255u]);
          COLD(C_EG) = __float_as_uint(__uint_as_float(COLD(C_EG)) * div255[(color >> 8) & 255u]);
          COLD(C_EB) = __float_as_uint(__uint_as_float(COLD(C_EB)) * div255[(color >> 16) & 255u]);
        } else {
          COLD(C_ATTEN) = __float_as_uint(d);
          i = 8;
        }
        ray = nr;
      }

      if (st >= ST_EVENT) {
        sd = (int)(vol.template step_i<SMALL>(f2i(ray.origin.x), f2i(ray.origin.y), f2i(ray.origin.z)) & 0x7Fu);
        steps_left = 70;
        st = ST_MARCH;
        COLD(C_FIX) = (fixw & ~0x1Cu) | ((uint32_t)(o - 1) << 2) | ((uint32_t)(i - 8) << 3);
        COLD(C_COLOR) = color;
      }
    }

    // ---- two rounds of the step loop on cur; between them lanes whose cur is parked swap in a marching alt ----
    for (int round = 0; round < 2; ++round) {
      if (round == 1) {
        const int ast = (int)(ameta & 255u);
        swap_rays(!(st == ST_MARCH || st == ST_CERT) && (ast == ST_MARCH || ast == ST_CERT));
#ifdef CLVR_BOUNCE_STATS
        st_swaps += 1;
#endif
      }
      if (__ballot(st == ST_MARCH || st == ST_CERT) == 0ull) continue;
      do {
#ifdef CLVR_BOUNCE_STATS
        st_step_iters += 1; st_step_lanes += (uint32_t)__popcll(__ballot(st == ST_MARCH));
#endif
        if (st == ST_MARCH) {
          ray.origin = ray.origin + ray.direction * fmaxf((float)sd, 0.5f);
          --steps_left;
          if (!USE_GRAD) {
            const bool has_voxel = __float_as_uint(ray.origin.x) < __float_as_uint((float)a.X) &&
                                   __float_as_uint(ray.origin.y) < __float_as_uint((float)a.Y) &&
                                   __float_as_uint(ray.origin.z) < __float_as_uint((float)a.Z);
            if (has_voxel) {
              const unsigned q = vol.template step_marched<SMALL>(ray.origin.x, ray.origin.y, ray.origin.z);
              sd = (int)(q & 0x7Fu);
              if (q & 0x80u) st = ST_EVENT + EV_HIT_COLOR_PENDING;
              else if (steps_left == 0) st = ST_EVENT + EV_NONE;
              else if (sd >= cert_at) st = ST_CERT;
            } else {
              st = ST_EVENT + EV_CHECK;
            }
          } else if (exited_volume(vol, ray.origin)) {
            st = ST_EVENT + EV_EXIT;
          } else {
            int next_sd;
            bool pending = false;
            uint32_t c = COLD(C_COLOR);
            const bool is_hit = classify_step<USE_GRAD, SMALL, true>(vol, a.tf, ray.origin, c, next_sd, &pending);
            if (is_hit) {
              if (!pending) COLD(C_COLOR) = c;  // the literal route's rule colour
              st = ST_EVENT + (pending ? EV_HIT_COLOR_PENDING : EV_HIT);
            } else if (steps_left == 0) {
              st = ST_EVENT + EV_NONE;
            } else {
              sd = next_sd;
              if (sd >= cert_at) st = ST_CERT;
            }
          }
        }
        const int n_cert = __popcll(__ballot(st == ST_CERT));
        if (n_cert != 0 && (n_cert >= cert_min_lanes || __popcll(__ballot(st == ST_MARCH)) < a.step_min_lanes)) {
          if (st == ST_CERT) st = certify_exit(a, ray.origin, ray.direction, steps_left) ? ST_EVENT + EV_EXIT : ST_MARCH;
        }
      } while (__popcll(__ballot(st == ST_MARCH)) >= a.step_min_lanes);
      // lanes still waiting for a certificate when the loop ends: look it up now (the loop's own condition does so only while it runs)
      if (__ballot(st == ST_CERT) != 0ull) {
        if (st == ST_CERT) st = certify_exit(a, ray.origin, ray.direction, steps_left) ? ST_EVENT + EV_EXIT : ST_MARCH;
      }
    }
  }
#undef COLD
}

// ------------------------------------------------------------------------------------------------
// k_env_fixup: one lane per fix-up record; exact lookups, then the reference's arithmetic in its order
template <int MODE>
__global__ __launch_bounds__(256) void k_env_fixup(const RenderArgs a) {
  uint32_t n = a.counters[2];
  if (n > a.fixup_capacity) n = a.fixup_capacity;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const uint32_t *rec = a.fixups + (size_t)k * kFixupDwords;
    const uint32_t hit = rec[0];
    const int64_t entry = (int64_t)(((uint64_t)rec[2] << 32) | (uint64_t)rec[1]);
    const uint32_t gx = rec[3] & 0xFFFFu, gy = rec[3] >> 16;
    uint32_t bv_r = rec[4], bv_g = rec[5], bv_b = rec[6];
    const uint32_t npend = rec[7];
    for (uint32_t q = 0; q < npend && q < 2u; ++q) {
      const uint32_t *e = rec + 8 + 7 * q;
      const float p_r = __uint_as_float(e[0]), p_g = __uint_as_float(e[1]), p_b = __uint_as_float(e[2]);
      const float factor = __uint_as_float(e[3]);
      const f3 d = f3{__uint_as_float(e[4]), __uint_as_float(e[5]), __uint_as_float(e[6])};
      const uint32_t light = sample_environment_map(a.env, a.env_w, a.env_h, d);
      bv_r = f2u((float)bv_r + p_r * (float)(light & 255u) * factor / 1.0f);
      bv_g = f2u((float)bv_g + p_g * (float)((light >> 8) & 255u) * factor / 1.0f);
      bv_b = f2u((float)bv_b + p_b * (float)((light >> 16) & 255u) * factor / 1.0f);
    }
    finish_item<MODE>(a, entry, hit, gx, gy, bv_r, bv_g, bv_b);
  }
}


// ------------------------------------------------------------------------------------------------
// k_ao: compute_ao (ray_marching.cl:104-149) for every primary hit, the launch's passes one after the other in the
// hit's own lane.  The cache entry is one 32-bit word per voxel, samples | occluded << 16 (the reference's 2-ushort view,
// utility.cl:123-159).  The reference updates it with a plain read-modify-write that races between pixels sharing a
// voxel; here the sample is claimed and the occlusion recorded with integer atomics, i.e. the pixels are serialised,
// which is one legal outcome of that race and independent of the order while the count stays below the cap of 100.
template <bool USE_GRAD>
__global__ __launch_bounds__(256) void k_ao(const RenderArgs a) {
  const uint32_t n_hits = a.n_hits_on_device ? a.counters[0] : a.n_hits;
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n_hits) return;
  const VolumePacked vol = make_volume(a);
  const uint4 *src = reinterpret_cast<const uint4 *>(&a.hits[h]);
  const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
  const f3 hit_origin = f3{__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z)};
  const f3 hit_direction = f3{__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y)};
  const f3 normal = f3{__uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(q2.x)};
  const int64_t entry = (int64_t)(((uint64_t)q2.w << 32) | (uint64_t)q2.z);
  const uint32_t gx = q3.x & 0xFFFFu, gy = q3.x >> 16;
  if (entry < 0) return;  // outside the allocation: nothing is recorded
  uint32_t *word = a.cache + entry;
  for (int s = 0; s < a.n_seeds; ++s) {
    // `if (buffer_value.x < 100) buffer_value.x += 1`: claim a sample, give it back if the cap was already reached
    const uint32_t old = atomicAdd(word, 1u);
    uint32_t granted = 1u, occluded = 0u;
    if ((old & 0xFFFFu) >= 100u) {
      atomicSub(word, 1u);
      granted = 0u;
    } else {
      // ray_bounce (utility_ray.cl:100-103), seven unclassified steps, then the occlusion march
      Ray r{hit_origin + hit_direction, hemisphere_direction(gx, gy, normal, a.seeds[s])};
      for (int k = 0; k < 7; ++k) {
        const int sd = (int)(vol.step_i(f2i(r.origin.x), f2i(r.origin.y), f2i(r.origin.z)) & 0x7Fu);
        r.origin = r.origin + r.direction * cl_max((float)sd, 0.5f);
      }
      int ev;
      uint32_t color = 0u;
      march_to_next_event<USE_GRAD>(vol, a.tf, r, ev, color);
      if (ev == EV_HIT) {
        atomicAdd(word, 0x10000u);
        occluded = 1u;
      }
    }
    if (a.contrib_out) {
      uint32_t *q = a.contrib_out + ((size_t)gy * (size_t)a.launch_w + gx) * 4;
      q[0] = occluded; q[1] = 0u; q[2] = 0u; q[3] = granted;
    }
  }
}

hipError_t launch_ao(const RenderArgs &a, hipStream_t s) {
  if (a.n_hits == 0) return hipSuccess;
  const dim3 grid((a.n_hits + 255u) / 256u), block(256);
  if (a.tf.uses_gradient) hipLaunchKernelGGL(k_ao<true>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(k_ao<false>, grid, block, 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// resolve: every pixel of this rank reads its accumulator after the whole pass (ray_marching.cl:82-99)
__global__ __launch_bounds__(64) void k_resolve(const RenderArgs a) {
  const uint32_t slot = blockIdx.x;
  int tx, ty;
  if (!tile_from_slot(a, slot, tx, ty)) return;
  const uint32_t lane = threadIdx.x;
  const uint32_t x = (uint32_t)tx * 8u + (lane & 7u);
  const uint32_t y = (uint32_t)ty * 8u + (lane >> 3);
  if (x >= (uint32_t)a.frame_w || y >= (uint32_t)a.frame_h) return;
  const uint32_t pslot = slot * 64u + lane;
  const uint32_t ps = a.pix_slot[pslot];
  uint32_t out;
  if (!(ps & PIX_HIT)) {
    out = (ps & 0x00FFFFFFu) | (200u << 24);  // miss: environment colour, alpha 200
  } else if (a.shading == CLWH_SHADE_AO) {
    // compute_ao's return value (ray_marching.cl:145-148): {v, v, v} with v = (100 - occluded) * 2; shown with alpha 1
    const HitRec &h = a.hits[ps & ~PIX_HIT];
    const int64_t e = (int64_t)(((uint64_t)(uint32_t)h.entry_hi << 32) | (uint64_t)(uint32_t)h.entry_lo);
    const uint32_t v = e < 0 ? 200u : (100u - (a.cache[e] >> 16)) * 2u;
    out = v | (v << 8) | (v << 16) | (1u << 24);
  } else if (a.mode == CLWH_ACCUM_VOXEL_CACHE) {
    const HitRec &h = a.hits[ps & ~PIX_HIT];
    const int64_t e = (int64_t)(((uint64_t)(uint32_t)h.entry_hi << 32) | (uint64_t)(uint32_t)h.entry_lo);
    if (e < 0) {
      out = 1u << 24;
    } else {
      const uint2 w = *reinterpret_cast<const uint2 *>(a.cache + e * 2);
      out = tone_map_rgba8(w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16);
    }
  } else {
    const float4 acc = a.accum[pslot];
    out = tone_map_rgba8((uint32_t)acc.x, (uint32_t)acc.y, (uint32_t)acc.z, (uint32_t)acc.w);
  }
  a.frame[(size_t)y * a.frame_w + x] = out;
}

// gathered image-space accumulation (all ranks' tile-major buffers back to back) -> RGBA8 frame
__global__ __launch_bounds__(64) void k_accum_resolve(const RenderArgs a, const float4 *__restrict__ accum_all) {
  const int tx = (int)(blockIdx.x % (unsigned)a.tiles_x), ty = (int)(blockIdx.x / (unsigned)a.tiles_x);
  const int owner = (tx + ty) % a.tile_world;
  const size_t slot = (size_t)ty * a.tiles_per_row + (size_t)(tx / a.tile_world);
  const size_t per_rank = (size_t)a.tiles_y * a.tiles_per_row * 64u;
  const uint32_t lane = threadIdx.x;
  const float4 acc = accum_all[(size_t)owner * per_rank + slot * 64u + lane];
  const uint32_t x = (uint32_t)tx * 8u + (lane & 7u), y = (uint32_t)ty * 8u + (lane >> 3);
  if (x >= (uint32_t)a.frame_w || y >= (uint32_t)a.frame_h) return;
  uint32_t out;
  if (acc.w == 0.0f) {
    const f3 cam_o = f3{a.cam_pos[0], a.cam_pos[1], a.cam_pos[2]};
    const f3 cam_d = f3{a.cam_dir[0], a.cam_dir[1], a.cam_dir[2]};
    const Ray vray = generate_ray(cam_o, cam_d, (int)x, (int)y, a.frame_w, a.frame_h);
    const uint32_t e = sample_environment_map(a.env, a.env_w, a.env_h, vray.direction);
    out = (e & 0x00FFFFFFu) | (200u << 24);
  } else {
    out = tone_map_rgba8((uint32_t)acc.x, (uint32_t)acc.y, (uint32_t)acc.z, (uint32_t)acc.w);
  }
  a.frame[(size_t)y * a.frame_w + x] = out;
}

// The multi-GPU form of the same resolve: a rank resolves ITS tiles to RGBA8 first (tile-major, the slot order of its
// accumulation buffer), the ranks exchange 4 bytes per pixel instead of 16, and k_frame_from_tiles puts the tiles in place.
__global__ __launch_bounds__(64) void k_accum_resolve_tiles(const RenderArgs a, const float4 *__restrict__ accum, uint32_t *__restrict__ tiles_out) {
  const uint32_t slot = blockIdx.x, lane = threadIdx.x;
  int tx, ty;
  const bool exists = tile_from_slot(a, slot, tx, ty);  // the last slots of a row may lie beyond the frame
  const uint32_t x = (uint32_t)tx * 8u + (lane & 7u), y = (uint32_t)ty * 8u + (lane >> 3);
  uint32_t out = 0u;
  if (exists && x < (uint32_t)a.frame_w && y < (uint32_t)a.frame_h) {
    const float4 acc = accum[(size_t)slot * 64u + lane];
    if (acc.w == 0.0f) {
      const f3 cam_o = f3{a.cam_pos[0], a.cam_pos[1], a.cam_pos[2]};
      const f3 cam_d = f3{a.cam_dir[0], a.cam_dir[1], a.cam_dir[2]};
      const Ray vray = generate_ray(cam_o, cam_d, (int)x, (int)y, a.frame_w, a.frame_h);
      const uint32_t e = sample_environment_map(a.env, a.env_w, a.env_h, vray.direction);
      out = (e & 0x00FFFFFFu) | (200u << 24);
    } else {
      out = tone_map_rgba8((uint32_t)acc.x, (uint32_t)acc.y, (uint32_t)acc.z, (uint32_t)acc.w);
    }
  }
  tiles_out[(size_t)slot * 64u + lane] = out;
}

__global__ __launch_bounds__(64) void k_frame_from_tiles(const RenderArgs a, const uint32_t *__restrict__ tiles_all) {
  const int tx = (int)(blockIdx.x % (unsigned)a.tiles_x), ty = (int)(blockIdx.x / (unsigned)a.tiles_x);
  const int owner = (tx + ty) % a.tile_world;
  const size_t slot = (size_t)ty * a.tiles_per_row + (size_t)(tx / a.tile_world);
  const size_t per_rank = (size_t)a.tiles_y * a.tiles_per_row * 64u;
  const uint32_t lane = threadIdx.x;
  const uint32_t x = (uint32_t)tx * 8u + (lane & 7u), y = (uint32_t)ty * 8u + (lane >> 3);
  if (x >= (uint32_t)a.frame_w || y >= (uint32_t)a.frame_h) return;
  a.frame[(size_t)y * a.frame_w + x] = tiles_all[(size_t)owner * per_rank + slot * 64u + lane];
}

// ------------------------------------------------------------------------------------------------
hipError_t launch_repack(const RepackArgs &a, hipStream_t s) {
  const dim3 grid(((unsigned)a.NBX + (unsigned)(kRepackX / 8) - 1u) / (unsigned)(kRepackX / 8), (unsigned)a.NBY, (unsigned)a.NBZ);  // every dimension far below the 2^32 work-item limit
  hipLaunchKernelGGL(k_repack, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_macro_table(const uint32_t *brick_min, int NBX, int NBY, int NBZ, uint8_t *macro8, int X, int Y, int Z, int shift, hipStream_t s) {
  const int M = 1 << shift;
  const int MNX = (X + M - 1) >> shift, MNY = (Y + M - 1) >> shift, MNZ = (Z + M - 1) >> shift;
  uint2 *macro = reinterpret_cast<uint2 *>(macro8);
  const unsigned n = (unsigned)(MNX * MNY * MNZ);
  hipLaunchKernelGGL(k_macro_table, dim3((n + 255u) / 256u), dim3(256), 0, s, brick_min, NBX, NBY, NBZ, macro, MNX, MNY, MNZ, shift);
  const int lines[3] = {MNY * MNZ, MNZ * MNX, MNX * MNY};
  for (int axis = 0; axis < 3; ++axis)
    hipLaunchKernelGGL(k_macro_octants, dim3(((unsigned)lines[axis] + 63u) / 64u), dim3(64), 0, s, macro, MNX, MNY, MNZ, axis);
  hipLaunchKernelGGL(k_macro_bounds, dim3((n + 255u) / 256u), dim3(256), 0, s, macro, MNX, MNY, MNZ);
  return hipGetLastError();
}

hipError_t launch_primary(const RenderArgs &a, hipStream_t s) {
  if (a.tf.uses_gradient)
    hipLaunchKernelGGL(k_primary<true>, dim3(a.num_tile_slots), dim3(64), 0, s, a);
  else
    hipLaunchKernelGGL(k_primary<false>, dim3(a.num_tile_slots), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_bounce(const RenderArgs &a_in, hipStream_t s) {
  RenderArgs a = a_in;
  const uint64_t total = (uint64_t)a.n_hits * (uint64_t)a.n_seeds;
  if (total == 0) return hipSuccess;
  // persistent grid: enough waves to fill the chip (256 CUs x 32 waves), never more than the work
  const uint64_t waves_needed = (total + 63u) / 64u;
  constexpr unsigned wpb = (unsigned)kBounceThreads / 64u;  // waves per block; bounce_max_blocks counts 256-thread blocks
  // (two frame jobs in flight on two streams share the chip: a rank's share of a multi-GPU job -- a few million items -- runs 9 % (4 ranks)
  // to 18 % (8 ranks) faster when each launch takes half the grid, CLWH_TUNE_BLOCKS=1024, which bench.py sets for such runs; alone on the
  // GPU the same launch is 20-30 % slower on half the grid, so the default stays the full chip: profiles/r02_emulate_rank_grid_sweep.txt)
  const unsigned blocks = (unsigned)std::min<uint64_t>((waves_needed + wpb - 1u) / wpb, ((uint64_t)a.bounce_max_blocks * 4u + wpb - 1u) / wpb);
  const dim3 grid(blocks), block(kBounceThreads);
  // Scheduling thresholds (0 = automatic).  A launch with only a few units per wave (one or a few passes) is
  // bound by its longest dependent chain: every wave steps its samples to completion and refills when empty.
  // A long launch is bound by VALU issue: lanes refill at 16 idle and the march phase ends at 16 marching lanes.
  // Measured crossover on the headline scene: between 4 and 8 passes per launch = about 6 units per wave
  // (round-1 sweep, git history: profiles/r01_tune_refill_step_thresholds.txt; re-swept in profiles/r02_tune_k_bounce_knobs.txt).
  // (with the hit count still on the device n_hits is the pixel count, an upper bound; the class follows the estimate: the last
  // camera's count with slack, clwh_render)
  const uint64_t waves_likely = ((uint64_t)a.n_hits_estimate * (uint64_t)a.n_seeds + 63u) / 64u;
  const bool long_launch = a.force_long_launch || waves_likely >= 6u * (uint64_t)blocks * wpb;
  if (a.step_min_lanes <= 0) a.step_min_lanes = long_launch ? 16 : 1;
  if (a.refill_min_lanes <= 0) a.refill_min_lanes = long_launch ? 16 : 64;
  // certificates: a long launch looks them up once 16 lanes of a wave wait for one (8: 4.32, 16: 4.25, 4: 4.42 ms); a short launch
  // is bound by its longest chain of dependent fetches, where the look-up is one more of them: 0.263 ms per pass without, 0.277 with
  a.cert_min_lanes = kCertPhaseMinLanes;
  if (!long_launch) a.cert_min_step = 0;
  if ((uint64_t)(((a.n_hits + 63u) >> 6) + 8u * (1u << a.unit_block_log2)) * (uint64_t)a.n_seeds >= (1ull << 24)) return hipErrorInvalidValue;  // udivmod24
  const bool g = a.tf.uses_gradient != 0;
  // fewer than 2^23 bricks (up to ~1600^3): every step byte has a 32-bit offset -> the march's 32-bit addressing, with
  // the index terms of the three axes in LDS tables
  const bool small = (uint64_t)a.NBX * (uint64_t)a.NBY * (uint64_t)((a.Z + 7) / 8) < (1ull << 23) &&
                     a.X + a.Y + a.Z <= VolumePacked::kPartsMaxEntries;
  const size_t lds_parts_bytes = small ? (size_t)(a.X + a.Y + a.Z) * sizeof(uint32_t) : 0u;
  // CLWH_TUNE_BOUNCE_RAYS=2: long launches run k_bounce2 (two rays per lane; hit index and seed index share a dword: 2^26 hits)
  const bool two_rays = a.bounce_rays == 2 && long_launch && a.n_hits < (1u << 26) && kBounceThreads == 256;
  if (two_rays) a.fixup_capacity = std::min<uint32_t>(a.fixup_capacity, (1u << 27) - 4u);
#define CLVR_LAUNCH_BOUNCE(G, M)                                                                    \
  do {                                                                                              \
    if (two_rays) {                                                                                 \
      if (small) hipLaunchKernelGGL((k_bounce2<G, M, true>), grid, block, lds_parts_bytes, s, a);  \
      else hipLaunchKernelGGL((k_bounce2<G, M, false>), grid, block, 0, s, a);                     \
    } else if (small) hipLaunchKernelGGL((k_bounce<G, M, true>), grid, block, lds_parts_bytes, s, a); \
    else hipLaunchKernelGGL((k_bounce<G, M, false>), grid, block, 0, s, a);                        \
  } while (0)
  if (a.mode == CLWH_ACCUM_VOXEL_CACHE) {
    if (g) CLVR_LAUNCH_BOUNCE(true, CLWH_ACCUM_VOXEL_CACHE);
    else CLVR_LAUNCH_BOUNCE(false, CLWH_ACCUM_VOXEL_CACHE);
  } else {
    if (g) CLVR_LAUNCH_BOUNCE(true, CLWH_ACCUM_IMAGE_SPACE);
    else CLVR_LAUNCH_BOUNCE(false, CLWH_ACCUM_IMAGE_SPACE);
  }
#undef CLVR_LAUNCH_BOUNCE
  return hipGetLastError();
}

// finish the samples whose environment lookups the fast path could not certify
hipError_t launch_env_fixup(const RenderArgs &a, hipStream_t s) {
  if ((uint64_t)a.n_hits * (uint64_t)a.n_seeds == 0) return hipSuccess;  // n_hits is the pixel count when the real one is on the device
  // (the record count is on the device: a grid-stride loop; a 64-seed launch leaves tens of thousands of records of binary64 work:
  // 47.6 us on 64 blocks, 19.6 on 256, 20.2 on 1024)
  const unsigned blocks = a.n_seeds > 1 ? 256u : 64u;
  if (a.mode == CLWH_ACCUM_VOXEL_CACHE)
    hipLaunchKernelGGL(k_env_fixup<CLWH_ACCUM_VOXEL_CACHE>, dim3(blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(k_env_fixup<CLWH_ACCUM_IMAGE_SPACE>, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_commit(const RenderArgs &a, hipStream_t s) {
  if (a.n_hits == 0 || a.mode != CLWH_ACCUM_IMAGE_SPACE) return hipSuccess;
  hipLaunchKernelGGL(k_commit, dim3((a.n_hits + 255u) / 256u), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_vox_keys(const RenderArgs &a, int64_t *keys, uint32_t *iota, uint32_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_vox_keys, dim3((n + 255u) / 256u), dim3(256), 0, s, a, keys, iota, n);
  return hipGetLastError();
}

hipError_t launch_vox_grant(const RenderArgs &a, const int64_t *sorted_keys, const uint32_t *order, uint32_t n, uint32_t *grants, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_vox_grant, dim3((n + 255u) / 256u), dim3(256), 0, s, a, sorted_keys, order, n, grants);
  return hipGetLastError();
}

hipError_t launch_commit_voxel(const RenderArgs &a, hipStream_t s) {
  if (a.n_hits == 0) return hipSuccess;
  hipLaunchKernelGGL(k_commit_voxel, dim3((a.n_hits + 255u) / 256u), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_resolve(const RenderArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_resolve, dim3(a.num_tile_slots), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_accum_resolve_tiles(const RenderArgs &a, const float4 *accum, uint32_t *tiles_out, hipStream_t s) {
  hipLaunchKernelGGL(k_accum_resolve_tiles, dim3((uint32_t)(a.tiles_y * a.tiles_per_row)), dim3(64), 0, s, a, accum, tiles_out);
  return hipGetLastError();
}

hipError_t launch_frame_from_tiles(const RenderArgs &a, const uint32_t *tiles_all, hipStream_t s) {
  hipLaunchKernelGGL(k_frame_from_tiles, dim3((uint32_t)(a.tiles_x * a.tiles_y)), dim3(64), 0, s, a, tiles_all);
  return hipGetLastError();
}

hipError_t launch_accum_resolve(const RenderArgs &a, const float4 *accum_all, hipStream_t s) {
  hipLaunchKernelGGL(k_accum_resolve, dim3((uint32_t)(a.tiles_x * a.tiles_y)), dim3(64), 0, s, a, accum_all);
  return hipGetLastError();
}

}  // namespace clvr
