// render_device.hpp -- device-side building blocks of the `render` pass, written for gfx950.
//
// What each block computes follows the reference kernel sources (cited per function, paths
// relative to the reference's opencl_kernels/); how it is organised (accessor templates, runtime
// transfer-function table, 64-bit cache indices, tile ownership) is this project's design.
#pragma once

#include "clwh_internal.hpp"
#include "device_math.hpp"
#include "env_fast.hpp"
#include "packed_volume.hpp"

namespace clvr {

struct Ray {
  f3 origin;
  f3 direction;
};

enum Event : int { EV_NONE = 0, EV_HIT = 1, EV_EXIT = 2 };  // utility_ray.cl:119-123

// ------------------------------------------------------------------------------------------------
// transfer function (the generated is_event_gen: app/ui.cpp:160-168, app/tf_part.cpp:55-79).
// color: r | g<<8 | b<<16 | a<<24, only overwritten by a matching rule that assigns *color.
__device__ __forceinline__ bool tf_eval(const TfDev &tf, int value_in, int gradient_in, uint32_t &color) {
  const int value = (int)(short)value_in;
  const int gradient = (int)(short)gradient_in;
  for (int k = 0; k < tf.n; ++k) {
    const TfRuleDev &r = tf.rules[k];
    bool m = value >= r.v_lo && value <= r.v_hi;
    if (r.flags & TF_USE_GRADIENT) m = m && gradient >= r.g_lo && gradient <= r.g_hi;
    if (m) {
      if (r.flags & TF_WRITES_COLOR) color = r.color;
      return true;
    }
    if (r.flags & TF_TERMINAL) return false;
  }
  return false;
}

// ------------------------------------------------------------------------------------------------
// utility_filter.cl:2-35: central differences v(p+e_k) - v(p-e_k), no 1/2 factor, border texel = 0.
// The reference adds +-1 to the float position and samples at floor() of the sum; those six texels are the voxel's
// own neighbours unless an addition rounds across an integer (p.x = 255.99999 + 1 -> 257.0).  Such positions are
// detected exactly and take the literal six taps from the caller's image; every other position inside the volume
// reads the voxel's precomputed differences from its hit record: one 8-byte load, one line.
// Only the +1 taps can be irregular.  Every caller passes a position inside the volume (0 <= p < dimension < 2^24, or
// -0.0), and for such p the binary32 difference p - 1 always floors to floor(p) - 1: for p >= 2 it is exact (p - 1 stays a
// multiple of ulp(p)), for 1 <= p < 2 it is exact by Sterbenz' lemma, and for 0 <= p < 1 it lies in [-1, -2^-24] whatever
// the rounding, whose floor is -1.  p + 1 on the other hand rounds UP to the next integer when frac(p) is within half
// an ulp of 1 (0.99999997 + 1 = 2.0).
__device__ __forceinline__ bool taps_are_voxel_neighbours(f3 p) {
  return floorf(p.x + 1.0f) == floorf(p.x) + 1.0f && floorf(p.y + 1.0f) == floorf(p.y) + 1.0f &&
         floorf(p.z + 1.0f) == floorf(p.z) + 1.0f;
}

__device__ __forceinline__ f3 gradient_literal(const VolumePacked &v, f3 p) {
  const int dx = v.value_at(p.x + 1.0f, p.y + 0.0f, p.z + 0.0f) - v.value_at(p.x - 1.0f, p.y - 0.0f, p.z - 0.0f);
  const int dy = v.value_at(p.x + 0.0f, p.y + 1.0f, p.z + 0.0f) - v.value_at(p.x - 0.0f, p.y - 1.0f, p.z - 0.0f);
  const int dz = v.value_at(p.x + 0.0f, p.y + 0.0f, p.z + 1.0f) - v.value_at(p.x - 0.0f, p.y - 0.0f, p.z - 1.0f);
  return f3{(float)dx, (float)dy, (float)dz};
}

// true when the hit record of floor(p) answers for position p: p inside the volume (false for NaN) and its taps regular
__device__ __forceinline__ bool hit_record_serves(const VolumePacked &v, f3 p) {
  const bool inside = p.x >= 0.0f && p.y >= 0.0f && p.z >= 0.0f && p.x < (float)v.X && p.y < (float)v.Y && p.z < (float)v.Z;
  return inside && taps_are_voxel_neighbours(p);
}

template <int SMALL = 0>
__device__ __forceinline__ f3 gradient_nn(const VolumePacked &v, f3 p) {
  if (hit_record_serves(v, p)) {
    int gx, gy, gz;
    VolumePacked::hit_gradient(v.template hit_record<SMALL>(p.x, p.y, p.z), gx, gy, gz);
    return f3{(float)gx, (float)gy, (float)gz};
  }
  return gradient_literal(v, p);
}

// ------------------------------------------------------------------------------------------------
// camera (utility_ray.cl:69-89) and box entry (utility_ray.cl:19-66, 92-97)
__device__ __forceinline__ Ray generate_ray(f3 cam_origin, f3 cam_dir, int x, int y, int x_total, int y_total) {
  const f3 up = f3{0.0f, 1.0f, 0.0f};
  const f3 cam_side = normalize3(cross3(up, cam_dir));
  f3 cam_up = normalize3(cross3(cam_dir, cam_side));
  if (cam_up.y < 0.0f) cam_up = -cam_up;
  const float x_f = (float)(x - x_total / 2);
  const float y_f = (float)(y - y_total / 2);
  const float aspect_ratio = (float)x_total / (float)y_total;
  const float x_offset = x_f / (float)x_total * aspect_ratio;
  const float y_offset = y_f / (float)y_total;
  const f3 point_on_plane = (cam_dir + cam_side * x_offset) + cam_up * y_offset;
  return Ray{cam_origin, normalize3(point_on_plane)};
}

__device__ __forceinline__ float minimum_cut(float dim, float o, float d) {
  const float a = (dim - o) / d;
  const float b = (-o) / d;
  if (a <= 0.0f || b <= 0.0f) return 0.0f;
  return cl_min(a, b);
}
__device__ __forceinline__ bool within(float v, float dim) { return v <= dim && v >= 0.0f; }

// literal restatement of the reference's three-plane test (later axes override earlier ones)
__device__ __forceinline__ bool cut_box(float dx, float dy, float dz, Ray shot, f3 &cut_point) {
  bool res = false;
  f3 cp = f3{0.0f, 0.0f, 0.0f};
  const f3 xc = shot.origin + shot.direction * minimum_cut(dx, shot.origin.x, shot.direction.x);
  const f3 yc = shot.origin + shot.direction * minimum_cut(dy, shot.origin.y, shot.direction.y);
  const f3 zc = shot.origin + shot.direction * minimum_cut(dz, shot.origin.z, shot.direction.z);
  if (within(xc.y, dy) && within(xc.z, dz)) { res = true; cp = xc; }
  if (within(yc.x, dx) && within(yc.z, dz)) { res = true; cp = yc; }
  if (within(zc.x, dx) && within(zc.y, dy)) { res = true; cp = zc; }
  cut_point = cp;
  return res;
}

// ------------------------------------------------------------------------------------------------
// sampling (utility_sampling.cl:40-50) and bounce (utility_ray.cl:106-109)
__device__ __forceinline__ f3 hemisphere_reflective(uint32_t gx, uint32_t gy, f3 normal, int seed, float roughness) {
  const uint32_t useed = (uint32_t)seed + (gx + 1u) * (gy + 1u);
  const int rx = (int)hash_u32(useed * 0x182205bdu);
  const int ry = (int)hash_u32(useed * 0xe8d052f3u);
  const int rz = (int)hash_u32(useed * 0xf1981dcfu);
  // C signed remainder: range [-3071, 1023] (biased, as in the reference)
  const f3 direction = f3{(float)((rx % 2048) - 1024), (float)((ry % 2048) - 1024), (float)((rz % 2048) - 1024)};
  const float decider = dot3(direction, normal);
  const f3 correct_direction = normalize3(direction * decider);
  return normalize3(normal * (1.0f - roughness) + correct_direction * roughness);
}

// utility_sampling.cl:25-36 get_hemisphere_direction (compute_ao's bounce): no roughness blend
__device__ __forceinline__ f3 hemisphere_direction(uint32_t gx, uint32_t gy, f3 normal, int seed) {
  const uint32_t useed = (uint32_t)seed + (gx + 1u) * (gy + 1u);
  const int rx = (int)hash_u32(useed * 0x182205bdu);
  const int ry = (int)hash_u32(useed * 0xe8d052f3u);
  const int rz = (int)hash_u32(useed * 0xf1981dcfu);
  const f3 direction = f3{(float)((rx % 2048) - 1024), (float)((ry % 2048) - 1024), (float)((rz % 2048) - 1024)};
  const float decider = dot3(direction, normal);
  return normalize3(direction * decider);
}

// ------------------------------------------------------------------------------------------------
// environment map (utility_environment_map.cl:3-13): equirectangular, normalised coords,
// clamp-to-edge, nearest
__device__ __forceinline__ uint32_t sample_environment_map(const uint32_t *__restrict__ env, int w, int h, f3 d) {
  float u = cr_atan2f(d.x, d.z);
  float v = cr_asinf(-d.y);
  u = u * 0.1591549431f;
  v = v * 0.318309886f;
  u = u + 0.5f;
  v = v + 0.5f;
  int i = f2i(floorf(u * (float)w));
  int j = f2i(floorf(v * (float)h));
  i = min(max(i, 0), w - 1);
  j = min(max(j, 0), h - 1);
  return env[(size_t)j * (size_t)w + (size_t)i];
}

// the same lookup through the certified fast path (env_fast.hpp); false = undecided, use the exact one
__device__ __forceinline__ bool sample_environment_map_fast(const uint32_t *__restrict__ env, int w, int h, f3 d,
                                                            uint32_t &texel) {
  int32_t i, j;
  if (!env_texel_fast(d.x, d.y, d.z, w, h, i, j)) return false;
  texel = env[(size_t)j * (size_t)w + (size_t)i];
  return true;
}

// ------------------------------------------------------------------------------------------------
// marching (utility_ray.cl:112-168)
template <class Vol>
__device__ __forceinline__ bool exited_volume(const Vol &v, f3 q) {
  const bool exited_max = ((float)v.X < q.x) | ((float)v.Y < q.y) | ((float)v.Z < q.z);
  const bool exited_min = (q.x < 0.0f) | (q.y < 0.0f) | (q.z < 0.0f);
  return exited_max | exited_min;
}

// Transfer functions that read `gradient`: the class byte is baked from the gradient at the voxel's INTEGER position;
// positions whose taps are not the voxel's neighbours (taps_are_voxel_neighbours above) take the literal 7-fetch route.

// the colour of a Hit found through the step byte: the voxel's class (in its hit record) names the transfer-function rule
template <int SMALL = 0>
__device__ __forceinline__ void hit_color(const VolumePacked &v, const TfDev &tf, f3 pos, uint32_t &color) {
  const unsigned cls = VolumePacked::hit_class(v.template hit_record<SMALL>(pos.x, pos.y, pos.z));  // a step-byte Hit is inside the volume
  const TfRuleDev &rule = tf.rules[cls - 1u];
  if (rule.flags & TF_WRITES_COLOR) color = rule.color;
}

// what a Hit needs from memory in ONE 8-byte load: the rule colour (when `want_color`, i.e. the Hit was found through the
// step byte and its colour is still pending) and the gradient for the normal
template <int SMALL = 0>
__device__ __forceinline__ f3 hit_gradient_and_color(const VolumePacked &v, const TfDev &tf, f3 pos, bool want_color, uint32_t &color) {
  const bool regular = hit_record_serves(v, pos);
  if (regular || want_color) {
    const uint2 r = v.template hit_record<SMALL>(pos.x, pos.y, pos.z);  // either condition implies pos is inside the volume
    if (want_color) {
      const TfRuleDev &rule = tf.rules[VolumePacked::hit_class(r) - 1u];
      if (rule.flags & TF_WRITES_COLOR) color = rule.color;
    }
    if (regular) {
      int gx, gy, gz;
      VolumePacked::hit_gradient(r, gx, gy, gz);
      return f3{(float)gx, (float)gy, (float)gz};
    }
  }
  return gradient_literal(v, pos);
}

// one march step's classification on the packed volume: returns true on a Hit (and updates `color`), otherwise
// `next_sd` is the SDF value for the next step.
// DEFER_COLOR: a Hit found through the step byte returns with `color_pending` set instead of fetching the hit record
// for the rule's colour -- the persistent bounce kernel does that in its event phase together with the normal's gradient
// (hit_gradient_and_color), where it costs one pass per event phase rather than one per march iteration in which any
// lane happens to hit.
template <bool USE_GRAD, int SMALL = 0, bool DEFER_COLOR = false>
__device__ __forceinline__ bool classify_step(const VolumePacked &v, const TfDev &tf, f3 pos, uint32_t &color, int &next_sd,
                                              bool *color_pending = nullptr) {
  // both callers have just tested !exited_volume(pos): every coordinate is >= 0 (or -0.0), <= its dimension, or NaN.
  // `inside`: the position has a voxel; otherwise (coordinate == dimension, NaN) the reference reads the border texel 0
  const bool inside = pos.x < (float)v.X && pos.y < (float)v.Y && pos.z < (float)v.Z;
  if (USE_GRAD && !tf.opaque && (tf.literal_gradient_taps || !inside || !taps_are_voxel_neighbours(pos))) {
    // the reference's own seven fetches, from the caller's images (utility_ray.cl:126-138)
    const int value = v.value_at(pos.x, pos.y, pos.z);
    const int gradient = (int)(short)f2i(length3(gradient_literal(v, pos)));
    const int sd = v.sdf_at_f(pos.x, pos.y, pos.z);
    next_sd = sd > 0 ? sd : 0;
    return tf_eval(tf, value, gradient, color);
  }
  if (!inside) {
    // the border texel: value 0, SDF 0; a Hit when a rule contains 0 (its colour needs no memory access)
    next_sd = 0;
    if (tf.border_class == 0) return false;
    const TfRuleDev &rule = tf.rules[tf.border_class - 1];
    if (rule.flags & TF_WRITES_COLOR) color = rule.color;
    return true;
  }
  const unsigned q = v.template step_marched<SMALL>(pos.x, pos.y, pos.z);
  next_sd = (int)(q & 0x7Fu);
  if (!(q & 0x80u)) return false;
  if (DEFER_COLOR) {
    *color_pending = true;
    return true;
  }
  hit_color<SMALL>(v, tf, pos, color);
  return true;
}

// The same march on the packed volume: one 1-byte gather per step.  The byte fetched at floor(new
// origin) classifies the new position and carries the SDF value of the NEXT step: inside the volume
// (0 <= coord <= dim, which `!exited_volume` guarantees) trunc == floor, coord == dim reads the border
// on both sides, and a NaN origin stays NaN whatever the step length.  Only the first SDF read of a march
// is at an arbitrary origin and keeps the truncating int-coordinate read.
template <bool USE_GRAD>
__device__ __forceinline__ Ray march_to_next_event(const VolumePacked &v, const TfDev &tf, Ray cur, int &event,
                                                   uint32_t &color) {
  int ev = EV_NONE;
  int sd = (int)(v.step_i(f2i(cur.origin.x), f2i(cur.origin.y), f2i(cur.origin.z)) & 0x7Fu);
  for (int i = 0; i < 70; ++i) {
    const float step_size = cl_max((float)sd, 0.5f);
    cur.origin = cur.origin + cur.direction * step_size;
    if (exited_volume(v, cur.origin)) { ev = EV_EXIT; break; }
    if (classify_step<USE_GRAD>(v, tf, cur.origin, color, sd)) { ev = EV_HIT; break; }
  }
  event = ev;
  return cur;
}

// ------------------------------------------------------------------------------------------------
// voxel cache (utility.cl:20-54, 93-105): entry = 4 x u16 {r,g,b,count} seen as 2 x u32
__device__ __forceinline__ int64_t cache_entry_of(int X, int Z, f3 pos) {
  const int64_t x = f2i(pos.x), y = f2i(pos.y), z = f2i(pos.z);
  return (int64_t)X * (int64_t)Z * y + (int64_t)X * z + x;  // y-major, then z, then x
}

__device__ __forceinline__ bool cache_take_token(uint32_t *cache, int64_t e, uint32_t max_tokens) {
  uint32_t *words = cache + e * 2;
  const uint32_t w1 = __hip_atomic_load(words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int16_t w = (int16_t)(w1 >> 16);
  if ((uint32_t)(int32_t)w > max_tokens) return false;
  const int32_t t = (int32_t)atomicAdd(words + 1, 0x00010000u);
  if ((uint32_t)(t >> 16) < max_tokens) return true;
  atomicSub(words + 1, 0x00010000u);
  return false;
}

__device__ __forceinline__ void cache_add(uint32_t *cache, int64_t e, uint32_t r, uint32_t g, uint32_t b, uint32_t a) {
  uint32_t *words = cache + e * 2;
  atomicAdd(words, (r & 0xFFFFu) + ((g & 0xFFFFu) << 16));
  atomicAdd(words + 1, (b & 0xFFFFu) + ((a & 0xFFFFu) << 16));
}

// ray_marching.cl:82-99: integer mean, tone curve, truncation; returns packed RGBA8 with alpha 1
__device__ __forceinline__ uint32_t tone_map_rgba8(uint32_t sr, uint32_t sg, uint32_t sb, uint32_t count) {
  if (count == 0u) return 1u << 24;
  const uint32_t r = sr / count, g = sg / count, b = sb / count;
  const float inv_gamma = 1.0f / 1.77777777f;
  const float brightness = 4.0f;
  float fx = (float)r / 255.0f, fy = (float)g / 255.0f, fz = (float)b / 255.0f;
  fx = cr_powf(fx * brightness, inv_gamma) * 255.0f;
  fy = cr_powf(fy * brightness, inv_gamma) * 255.0f;
  fz = cr_powf(fz * brightness, inv_gamma) * 255.0f;
  const uint32_t ur = min(f2u(fx), 255u), ug = min(f2u(fy), 255u), ub = min(f2u(fz), 255u);
  return ur | (ug << 8) | (ub << 16) | (1u << 24);
}

// ------------------------------------------------------------------------------------------------
// image-tile ownership: 8x8 pixel tiles, owner = (tx + ty) % world; a rank's tiles are numbered
// slot = ty * tiles_per_row + tx / world  (tile-major accumulation / scratch layout)
__device__ __forceinline__ bool tile_from_slot(const RenderArgs &a, uint32_t slot, int &tx, int &ty) {
  ty = (int)(slot / (uint32_t)a.tiles_per_row);
  const int k = (int)(slot % (uint32_t)a.tiles_per_row);
  int first = (a.tile_rank - ty) % a.tile_world;
  if (first < 0) first += a.tile_world;
  tx = first + k * a.tile_world;
  return tx < a.tiles_x && ty < a.tiles_y;
}

}  // namespace clvr
