// render_kernels.hip -- the `render` pass for gfx950 (one sample per pixel per pass).
//
// v0: one work-group = one wave64 = one 8x8 pixel tile (the reference's own work-group shape,
// app/renderer.cpp:145), every lane walks its pixel's whole path (ray_marching.cl:10-101, 152-199),
// volume / SDF in the caller's linear layout.  The frame is NOT read back from the cache while other
// lanes are still adding to it (the reference's race): hit pixels record their cache
// entry and a second kernel resolves them after the pass, which is one legal outcome of that race
// and is deterministic.
#include "render_device.hpp"

namespace clvr {

// blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of tile slots so
// that neighbouring tiles (which walk neighbouring voxels) share one L2
__device__ __forceinline__ uint32_t xcd_contiguous_slot(uint32_t b, uint32_t nblocks) {
  const uint32_t per = nblocks >> 3;
  if (per == 0u || b >= (per << 3)) return b;
  return (b & 7u) * per + (b >> 3);
}

template <class Vol>
__device__ __forceinline__ Vol make_volume(const RenderArgs &a);
template <>
__device__ __forceinline__ VolumeLinear make_volume<VolumeLinear>(const RenderArgs &a) {
  return VolumeLinear{a.volume, a.sdf, a.X, a.Y, a.Z};
}
template <>
__device__ __forceinline__ VolumePacked make_volume<VolumePacked>(const RenderArgs &a) {
  return VolumePacked{a.packed, a.X, a.Y, a.Z, a.NBX, a.NBY};
}

template <bool USE_GRAD, class Vol>
__global__ __launch_bounds__(64) void k_render_v0(const RenderArgs a) {
  const uint32_t slot = xcd_contiguous_slot(blockIdx.x, a.num_blocks);
  int tx, ty;
  if (!tile_from_slot(a, slot, tx, ty)) return;
  const uint32_t lane = threadIdx.x;
  const uint32_t x = (uint32_t)tx * 8u + (lane & 7u);
  const uint32_t y = (uint32_t)ty * 8u + (lane >> 3);
  const size_t pslot = (size_t)slot * 64u + lane;
  const size_t pix = (size_t)y * (size_t)a.launch_w + x;

  const Vol vol = make_volume<Vol>(a);
  const f3 cam_o = f3{a.cam_pos[0], a.cam_pos[1], a.cam_pos[2]};
  const f3 cam_d = f3{a.cam_dir[0], a.cam_dir[1], a.cam_dir[2]};
  const Ray vray = generate_ray(cam_o, cam_d, (int)x, (int)y, a.frame_w, a.frame_h);
  const float dx = (float)a.X, dy = (float)a.Y, dz = (float)a.Z;

  int64_t hit_entry = -1;  // what resolve reads: entry, -1 = miss, -2 = hit outside the cache
  int64_t raw_entry = -1;  // parity output: the entry as computed, valid or not
  uint32_t contrib_r = 0, contrib_g = 0, contrib_b = 0, granted_flag = 0;

  bool cut_ok;
  f3 cut_point;
  if (!(within(vray.origin.x, dx) && within(vray.origin.y, dy) && within(vray.origin.z, dz))) {
    cut_ok = cut_box(dx, dy, dz, vray, cut_point);
  } else {
    cut_ok = true;
    cut_point = vray.origin;
  }

  bool hit = false;
  if (cut_ok) {
    Ray current_ray{cut_point, vray.direction};
    uint32_t current_color = 0u;
    int ev;
    current_ray = march_to_next_event<USE_GRAD>(vol, a.tf, current_ray, ev, current_color);
    if (ev == EV_HIT) {
      hit = true;
      const Ray hit_information = current_ray;
      hit_entry = cache_entry_of(a.X, a.Z, current_ray.origin);
      raw_entry = hit_entry;
      const bool entry_ok = hit_entry >= 0 && hit_entry < a.cache_entries;
      bool granted;
      if (a.mode == CLWH_ACCUM_VOXEL_CACHE)
        granted = entry_ok && cache_take_token(a.cache, hit_entry, 256u);
      else
        granted = true;

      if (granted) {
        const f3 normal = -normalize3(gradient_nn(vol, current_ray.origin));
        float r_energy = (float)(current_color & 255u) / 255.0f;
        float g_energy = (float)((current_color >> 8) & 255u) / 255.0f;
        float b_energy = (float)((current_color >> 16) & 255u) / 255.0f;
        uint32_t bv_r = 0, bv_g = 0, bv_b = 0;

        for (int o = 1; o <= 2; ++o) {  // dist_count = 2
          {
            const float roughness = (float)(current_color >> 24) / 255.0f;
            Ray nr;
            nr.origin = hit_information.origin + hit_information.direction;
            nr.direction = hemisphere_reflective(x, y, normal, a.seed + o, roughness);
            current_ray = nr;
          }
          current_ray.origin = current_ray.origin + normal * 2.0f;
          float atten = fabsf(dot3(current_ray.direction, normal));

          for (int i = 8; i <= 10; ++i) {  // path_length = 3
            current_ray = march_to_next_event<USE_GRAD>(vol, a.tf, current_ray, ev, current_color);
            if (ev == EV_EXIT) {
              const float factor = 8.0f / (float)i;
              const uint32_t light = sample_environment_map(a.env, a.env_w, a.env_h, current_ray.direction);
              // uint += float: promote, add, truncate back
              bv_r = f2u((float)bv_r + atten * r_energy * (float)(light & 255u) * factor / 1.0f);
              bv_g = f2u((float)bv_g + atten * g_energy * (float)((light >> 8) & 255u) * factor / 1.0f);
              bv_b = f2u((float)bv_b + atten * b_energy * (float)((light >> 16) & 255u) * factor / 1.0f);
              break;
            } else if (ev == EV_HIT) {
              const f3 normal2 = -normalize3(gradient_nn(vol, current_ray.origin));
              const float roughness = (float)(current_color >> 24) / 255.0f;
              Ray nr;
              nr.origin = current_ray.origin + current_ray.direction;
              nr.direction = hemisphere_reflective(x, y, normal2, a.seed + o + i, roughness);
              current_ray = nr;
              current_ray.origin = current_ray.origin + normal2 * 2.0f;
              atten *= fabsf(dot3(current_ray.direction, normal2));
              r_energy *= (float)(current_color & 255u) / 255.0f;
              g_energy *= (float)((current_color >> 8) & 255u) / 255.0f;
              b_energy *= (float)((current_color >> 16) & 255u) / 255.0f;
            }
          }
        }
        bv_r /= 2u; bv_g /= 2u; bv_b /= 2u;
        contrib_r = bv_r & 0xFFFFu; contrib_g = bv_g & 0xFFFFu; contrib_b = bv_b & 0xFFFFu;
        granted_flag = 1u;
        if (a.mode == CLWH_ACCUM_VOXEL_CACHE) {
          cache_add(a.cache, hit_entry, bv_r, bv_g, bv_b, 0u);
        } else {
          float4 acc = a.accum[pslot];
          acc.x += (float)contrib_r; acc.y += (float)contrib_g; acc.z += (float)contrib_b; acc.w += 1.0f;
          a.accum[pslot] = acc;
        }
      }
      if (a.mode == CLWH_ACCUM_VOXEL_CACHE && !entry_ok) hit_entry = -2;  // hit, but nothing to read
      if (a.mode == CLWH_ACCUM_IMAGE_SPACE) hit_entry = 0;                 // resolve reads accum, not the cache
    }
  }

  if (!hit) {
    // miss: environment colour, alpha 200 (ray_marching.cl:172-178, 188-195)
    const uint32_t e = sample_environment_map(a.env, a.env_w, a.env_h, vray.direction);
    const uint32_t color = (e & 0x00FFFFFFu) | (200u << 24);
    if (a.frame && x < (uint32_t)a.frame_w && y < (uint32_t)a.frame_h) a.frame[(size_t)y * a.frame_w + x] = color;
    if (a.mode == CLWH_ACCUM_IMAGE_SPACE)
      a.accum[pslot] = make_float4((float)(e & 255u), (float)((e >> 8) & 255u), (float)((e >> 16) & 255u), 0.0f);
  }
  a.hit_slot[pslot] = hit_entry;
  if (a.hit_index_out) a.hit_index_out[pix] = raw_entry;
  if (a.contrib_out) {
    uint32_t *q = a.contrib_out + pix * 4;
    q[0] = contrib_r; q[1] = contrib_g; q[2] = contrib_b; q[3] = granted_flag;
  }
}

// resolve: every hit pixel reads its accumulator after the whole pass (ray_marching.cl:82-99)
__global__ __launch_bounds__(64) void k_resolve(const RenderArgs a) {
  const uint32_t slot = blockIdx.x;
  int tx, ty;
  if (!tile_from_slot(a, slot, tx, ty)) return;
  const uint32_t lane = threadIdx.x;
  const uint32_t x = (uint32_t)tx * 8u + (lane & 7u);
  const uint32_t y = (uint32_t)ty * 8u + (lane >> 3);
  if (x >= (uint32_t)a.frame_w || y >= (uint32_t)a.frame_h) return;
  const size_t pslot = (size_t)slot * 64u + lane;
  const int64_t e = a.hit_slot[pslot];
  if (e == -1) return;  // miss: env colour already written
  uint32_t out;
  if (a.mode == CLWH_ACCUM_VOXEL_CACHE) {
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

// gathered image-space accumulation (all ranks' tile-major buffers, concatenated) -> RGBA8 frame
__global__ __launch_bounds__(64) void k_accum_resolve(const float4 *__restrict__ accum_all, int tile_world,
                                                      int tiles_x, int tiles_y, int tiles_per_row,
                                                      uint32_t *frame, int frame_w, int frame_h) {
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int owner = (tx + ty) % tile_world;
  const size_t slot = (size_t)ty * tiles_per_row + (size_t)(tx / tile_world);
  const size_t per_rank = (size_t)tiles_y * tiles_per_row * 64u;
  const uint32_t lane = threadIdx.x;
  const float4 acc = accum_all[(size_t)owner * per_rank + slot * 64u + lane];
  const uint32_t x = (uint32_t)tx * 8u + (lane & 7u), y = (uint32_t)ty * 8u + (lane >> 3);
  if (x >= (uint32_t)frame_w || y >= (uint32_t)frame_h) return;
  uint32_t out;
  if (acc.w == 0.0f)
    out = (uint32_t)acc.x | ((uint32_t)acc.y << 8) | ((uint32_t)acc.z << 16) | (200u << 24);
  else
    out = tone_map_rgba8((uint32_t)acc.x, (uint32_t)acc.y, (uint32_t)acc.z, (uint32_t)acc.w);
  frame[(size_t)y * frame_w + x] = out;
}

// volume + SDF + transfer function -> packed bricked records; one wave writes one 4x4x4 sub-brick
// (256 contiguous bytes)
__global__ __launch_bounds__(256) void k_repack(const RepackArgs a) {
  const size_t sub_id = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
  const size_t n_sub = (size_t)a.NBX * a.NBY * a.NBZ * 8u;
  if (sub_id >= n_sub) return;
  const size_t brick = sub_id >> 3;
  const unsigned sub = (unsigned)(sub_id & 7u);
  const int bx = (int)(brick % (size_t)a.NBX);
  const int by = (int)((brick / (size_t)a.NBX) % (size_t)a.NBY);
  const int bz = (int)(brick / ((size_t)a.NBX * (size_t)a.NBY));
  const unsigned lane = threadIdx.x & 63u;
  const int x = bx * 8 + (int)(sub & 1u) * 4 + (int)(lane & 3u);
  const int y = by * 8 + (int)((sub >> 1) & 1u) * 4 + (int)((lane >> 2) & 3u);
  const int z = bz * 8 + (int)((sub >> 2) & 1u) * 4 + (int)(lane >> 4);
  uint32_t r = 0u;
  if (x < a.X && y < a.Y && z < a.Z) {
    const size_t i = ((size_t)z * (size_t)a.Y + (size_t)y) * (size_t)a.X + (size_t)x;
    const int value = a.volume[i];
    const int sd = a.sdf[i];
    unsigned cls = 0u;
    if (!a.tf.uses_gradient) {
      // first matching rule wins; a terminal rule (`return (cond);`) ends the evaluation
      for (int k = 0; k < a.tf.n; ++k) {
        const TfRuleDev &rule = a.tf.rules[k];
        if (value >= rule.v_lo && value <= rule.v_hi) { cls = (unsigned)k + 1u; break; }
        if (rule.flags & TF_TERMINAL) break;
      }
    }
    r = ((uint32_t)value & 0xFFFFu) | (((uint32_t)sd & 0xFFu) << 16) | (cls << 24);
  }
  a.packed[sub_id * 64u + lane] = r;
}

hipError_t launch_repack(const RepackArgs &a, hipStream_t s) {
  const size_t n_sub = (size_t)a.NBX * a.NBY * a.NBZ * 8u;
  hipLaunchKernelGGL(k_repack, dim3((unsigned)((n_sub + 3u) / 4u)), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_render_v0(const RenderArgs &a, hipStream_t s) {
  if (a.packed) {
    if (a.tf.uses_gradient)
      hipLaunchKernelGGL((k_render_v0<true, VolumePacked>), dim3(a.num_blocks), dim3(64), 0, s, a);
    else
      hipLaunchKernelGGL((k_render_v0<false, VolumePacked>), dim3(a.num_blocks), dim3(64), 0, s, a);
  } else {
    if (a.tf.uses_gradient)
      hipLaunchKernelGGL((k_render_v0<true, VolumeLinear>), dim3(a.num_blocks), dim3(64), 0, s, a);
    else
      hipLaunchKernelGGL((k_render_v0<false, VolumeLinear>), dim3(a.num_blocks), dim3(64), 0, s, a);
  }
  return hipGetLastError();
}

hipError_t launch_resolve(const RenderArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_resolve, dim3(a.num_blocks), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_accum_resolve(const float4 *accum_all, int32_t tile_world, int32_t width, int32_t height,
                                uint32_t *frame, int32_t frame_w, int32_t frame_h, hipStream_t s) {
  const int tiles_x = width / 8, tiles_y = height / 8;
  const int tiles_per_row = (tiles_x + tile_world - 1) / tile_world;
  hipLaunchKernelGGL(k_accum_resolve, dim3((uint32_t)(tiles_x * tiles_y)), dim3(64), 0, s, accum_all, tile_world,
                     tiles_x, tiles_y, tiles_per_row, frame, frame_w, frame_h);
  return hipGetLastError();
}

}  // namespace clvr
