/*
 * oracle/orc_render.c -- CPU restatement of the reference `render` kernel and everything it calls.
 * TEST INFRASTRUCTURE ONLY (see orc.h).  PARITY UNPINNED for this file: the reference ships no
 * golden vector for `render`; every function below cites the reference lines it restates.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } f3;
typedef struct { f3 origin, direction; } ray_t;
typedef struct { int32_t x, y, z, w; } i4;
typedef struct { uint32_t x, y, z, w; } u4;
enum { EV_NONE = 0, EV_HIT = 1, EV_EXIT = 2 }; /* utility_ray.cl:119-123 */

typedef struct {
  const orc_render_params *p;
  uint64_t c[ORC_N_COUNTERS];
  uint32_t gx, gy; /* get_global_id(0), get_global_id(1) */
  int tf_uses_gradient;
  /* instrumentation only (orc_render_params.locality): never read by the algorithm */
  uint64_t loc[ORC_LOC_TOTAL];
  int in_bounce;
  int32_t hit_v[3];   /* voxel of the sample's primary hit */
  int32_t prev_v[3];  /* voxel of the ray's previous step-byte fetch */
  int have_prev;
} ctx_t;

static inline int32_t iabs32(int32_t v) { return v < 0 ? -v : v; }
/* one step-byte fetch of the bounce phase at integer voxel (x, y, z) */
static inline void loc_fetch(ctx_t *c, int32_t x, int32_t y, int32_t z) {
  const orc_render_params *p = c->p;
  if (!p->locality || !c->in_bounce) return;
  if (x < 0 || y < 0 || z < 0 || x >= p->X || y >= p->Y || z >= p->Z) { c->have_prev = 0; return; }
  c->loc[ORC_LOC_FETCHES]++;
  if (c->have_prev) {
    if ((x >> 2) == (c->prev_v[0] >> 2) && (y >> 2) == (c->prev_v[1] >> 2) && (z >> 2) == (c->prev_v[2] >> 2)) c->loc[ORC_LOC_SAME_SUB4]++;
    if ((x >> 3) == (c->prev_v[0] >> 3) && (y >> 3) == (c->prev_v[1] >> 3) && (z >> 3) == (c->prev_v[2] >> 3)) c->loc[ORC_LOC_SAME_BRICK8]++;
  }
  int32_t dx = iabs32(x - c->hit_v[0]), dy = iabs32(y - c->hit_v[1]), dz = iabs32(z - c->hit_v[2]);
  int32_t cheb = dx > dy ? dx : dy;
  if (dz > cheb) cheb = dz;
  if (cheb <= 8) c->loc[ORC_LOC_NEAR_8]++;
  if (cheb <= 16) c->loc[ORC_LOC_NEAR_16]++;
  if (cheb <= 32) c->loc[ORC_LOC_NEAR_32]++;
  if (cheb <= 64) c->loc[ORC_LOC_NEAR_64]++;
  if (p->uniform4) {
    const int64_t sx = (p->X + 3) / 4, sy = (p->Y + 3) / 4;
    if (p->uniform4[((int64_t)(z >> 2) * sy + (y >> 2)) * sx + (x >> 2)]) c->loc[ORC_LOC_UNIFORM4]++;
  }
  c->prev_v[0] = x; c->prev_v[1] = y; c->prev_v[2] = z;
  c->have_prev = 1;
}

/* ---- fixed semantics of implementation-defined OpenCL built-ins (orc.h header) ---- */
static inline float f_min(float a, float b) { return (b < a) ? b : a; }
static inline float f_max(float a, float b) { return (a < b) ? b : a; }
static inline int32_t f2i(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return INT32_MAX;
  if (v <= -2147483648.0f) return INT32_MIN;
  return (int32_t)v;
}
static inline uint32_t f2u(float v) {
  if (v != v) return 0u;
  if (v >= 4294967296.0f) return UINT32_MAX;
  if (v <= 0.0f) return 0u;
  return (uint32_t)v;
}
static inline float cr_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
static inline float cr_asinf(float v) { return (float)asin((double)v); }
static inline float cr_powf(float a, float b) { return (float)pow((double)a, (double)b); }

static inline f3 v_add(f3 a, f3 b) { f3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline f3 v_scale(f3 a, float s) { f3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline f3 v_neg(f3 a) { f3 r = {-a.x, -a.y, -a.z}; return r; }
static inline float v_dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline f3 v_cross(f3 a, f3 b) {
  f3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
  return r;
}
static inline float v_length(f3 a) { return sqrtf(v_dot(a, a)); }
static inline f3 v_normalize(f3 a) {
  float l = v_length(a);
  f3 r = {a.x / l, a.y / l, a.z / l};
  return r;
}

/* ---- image reads ---- */
static inline int64_t lin(const orc_render_params *p, int32_t x, int32_t y, int32_t z) {
  return ((int64_t)z * p->Y + y) * p->X + x;
}
static inline int in_range(const orc_render_params *p, int32_t x, int32_t y, int32_t z) {
  return x >= 0 && y >= 0 && z >= 0 && x < p->X && y < p->Y && z < p->Z;
}
/* read_imagei(volume, smp, float4): unnormalised, nearest, CLK_ADDRESS_CLAMP (border 0) */
static inline int32_t vol_read_f(ctx_t *c, float fx, float fy, float fz) {
  const orc_render_params *p = c->p;
  c->c[ORC_N_VOL]++;
  float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
  if (!(gx >= 0.0f && gy >= 0.0f && gz >= 0.0f && gx < (float)p->X && gy < (float)p->Y &&
        gz < (float)p->Z))
    return 0;
  return p->volume[lin(p, (int32_t)gx, (int32_t)gy, (int32_t)gz)];
}
/* read_imagei(sdf, smp, int4) */
static inline int32_t sdf_read_i(ctx_t *c, i4 q) {
  const orc_render_params *p = c->p;
  c->c[ORC_N_SDF]++;
  loc_fetch(c, q.x, q.y, q.z);
  if (!in_range(p, q.x, q.y, q.z)) return 0;
  return p->sdf[lin(p, q.x, q.y, q.z)];
}

/* utility.cl:13-16 make_int: {x, y, z, z}, truncation toward zero */
static inline i4 make_int(f3 v) {
  i4 r = {f2i(v.x), f2i(v.y), f2i(v.z), f2i(v.z)};
  return r;
}

/* generated is_event_gen (app/ui.cpp:160-168, app/tf_part.cpp:55-79, tests/sdf/sdf_test.cpp:22).
 * Arguments are converted to `short` at the call (utility_ray.cl:134). */
static inline int tf_eval(const orc_tf *tf, int32_t value_in, int32_t gradient_in, i4 *color) {
  int16_t value = (int16_t)value_in;
  int16_t gradient = (int16_t)gradient_in;
  for (int k = 0; k < tf->n; ++k) {
    const orc_tf_rule *r = &tf->rules[k];
    int m = value >= r->v_lo && value <= r->v_hi;
    if (r->use_gradient) m = m && gradient >= r->g_lo && gradient <= r->g_hi;
    if (m) {
      if (r->writes_color) {
        color->x = r->color[0]; color->y = r->color[1];
        color->z = r->color[2]; color->w = r->color[3];
      }
      return 1;
    }
    if (r->terminal) return 0;
  }
  return 0;
}

int orc_tf_eval(const orc_tf *tf, int32_t value, int32_t gradient, int32_t color[4]) {
  i4 c = {color[0], color[1], color[2], color[3]};
  int r = tf_eval(tf, value, gradient, &c);
  color[0] = c.x; color[1] = c.y; color[2] = c.z; color[3] = c.w;
  return r;
}

/* utility_filter.cl:2-35 gradient_prewitt_nn: central differences, no 1/2 factor */
static inline f3 gradient_nn(ctx_t *c, f3 p) {
  int32_t dx = 0, dy = 0, dz = 0;
  dx += vol_read_f(c, p.x + 1.0f, p.y + 0.0f, p.z + 0.0f);
  dx -= vol_read_f(c, p.x - 1.0f, p.y - 0.0f, p.z - 0.0f);
  dy += vol_read_f(c, p.x + 0.0f, p.y + 1.0f, p.z + 0.0f);
  dy -= vol_read_f(c, p.x - 0.0f, p.y - 1.0f, p.z - 0.0f);
  dz += vol_read_f(c, p.x + 0.0f, p.y + 0.0f, p.z + 1.0f);
  dz -= vol_read_f(c, p.x - 0.0f, p.y - 0.0f, p.z - 1.0f);
  f3 r = {(float)dx, (float)dy, (float)dz};
  return r;
}

/* utility_sampling.cl:13-21 */
uint32_t orc_hash(uint32_t seed) {
  seed = (seed ^ 61u) ^ (seed >> 16);
  seed <<= 3;
  seed ^= (seed >> 4);
  seed *= 0xDEADBEEFu;
  seed ^= (seed >> 15);
  return seed;
}

/* utility_sampling.cl:40-50 get_hemisphere_direction_reflective */
static inline f3 hemisphere_reflective(uint32_t gx, uint32_t gy, f3 normal, int32_t seed,
                                       float roughness) {
  uint32_t useed = (uint32_t)seed + (gx + 1u) * (gy + 1u);
  int32_t rx = (int32_t)orc_hash(useed * 0x182205bdu);
  int32_t ry = (int32_t)orc_hash(useed * 0xe8d052f3u);
  int32_t rz = (int32_t)orc_hash(useed * 0xf1981dcfu);
  f3 direction = {(float)((rx % 2048) - 1024), (float)((ry % 2048) - 1024),
                  (float)((rz % 2048) - 1024)};
  float decider = v_dot(direction, normal);
  f3 correct_direction = v_normalize(v_scale(direction, decider));
  return v_normalize(v_add(v_scale(normal, 1.0f - roughness), v_scale(correct_direction, roughness)));
}

void orc_hemisphere_reflective(const float normal[3], int32_t seed, uint32_t gx, uint32_t gy,
                               float roughness, float out[3]) {
  f3 n = {normal[0], normal[1], normal[2]};
  f3 r = hemisphere_reflective(gx, gy, n, seed, roughness);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* utility_ray.cl:106-109 ray_bounce_fake_reflectance */
static inline ray_t bounce_fake_reflectance(ctx_t *c, ray_t cur, f3 normal, int32_t seed,
                                            float roughness) {
  ray_t r;
  r.origin = v_add(cur.origin, cur.direction);
  r.direction = hemisphere_reflective(c->gx, c->gy, normal, seed, roughness);
  return r;
}

/* utility_ray.cl:69-89 generate_ray */
static inline ray_t generate_ray(f3 cam_origin, f3 cam_dir, int32_t x, int32_t y, int32_t x_total,
                                 int32_t y_total) {
  const f3 up = {0.0f, 1.0f, 0.0f};
  const f3 cam_side = v_normalize(v_cross(up, cam_dir));
  f3 cam_up = v_normalize(v_cross(cam_dir, cam_side));
  if (cam_up.y < 0.0f) cam_up = v_neg(cam_up);

  const float x_f = (float)(x - x_total / 2);
  const float y_f = (float)(y - y_total / 2);
  const float aspect_ratio = (float)x_total / (float)y_total;
  const float x_offset = x_f / (float)x_total * aspect_ratio;
  const float y_offset = y_f / (float)y_total;

  const f3 point_on_plane =
      v_add(v_add(cam_dir, v_scale(cam_side, x_offset)), v_scale(cam_up, y_offset));
  ray_t ret = {cam_origin, v_normalize(point_on_plane)};
  return ret;
}

void orc_generate_ray(const float cam_pos[3], const float cam_dir[3], int32_t x, int32_t y,
                      int32_t x_total, int32_t y_total, float out_dir[3]) {
  f3 o = {cam_pos[0], cam_pos[1], cam_pos[2]}, d = {cam_dir[0], cam_dir[1], cam_dir[2]};
  ray_t r = generate_ray(o, d, x, y, x_total, y_total);
  out_dir[0] = r.direction.x; out_dir[1] = r.direction.y; out_dir[2] = r.direction.z;
}

/* utility_ray.cl:19-35: cut_min_eval + MINIMUM_CUT */
static inline float minimum_cut(float dim, float o, float d) {
  float a = (dim - o) / d;
  float b = (-o) / d;
  if (a <= 0.0f || b <= 0.0f) return 0.0f;
  return f_min(a, b);
}
static inline int limits(float v, float dim) { return v <= dim && v >= 0.0f; }

/* utility_ray.cl:37-66 cut: literal restatement (not a slab test) */
static inline int cut(float dx, float dy, float dz, ray_t shot, f3 *cut_point) {
  int res = 0;
  f3 cp = {0.0f, 0.0f, 0.0f};
  float tx = minimum_cut(dx, shot.origin.x, shot.direction.x);
  f3 xc = v_add(shot.origin, v_scale(shot.direction, tx));
  float ty = minimum_cut(dy, shot.origin.y, shot.direction.y);
  f3 yc = v_add(shot.origin, v_scale(shot.direction, ty));
  float tz = minimum_cut(dz, shot.origin.z, shot.direction.z);
  f3 zc = v_add(shot.origin, v_scale(shot.direction, tz));
  if (limits(xc.y, dy) && limits(xc.z, dz)) { res = 1; cp = xc; }
  if (limits(yc.x, dx) && limits(yc.z, dz)) { res = 1; cp = yc; }
  if (limits(zc.x, dx) && limits(zc.y, dy)) { res = 1; cp = zc; }
  *cut_point = cp;
  return res;
}

int orc_cut(int32_t X, int32_t Y, int32_t Z, const float origin[3], const float dir[3],
            float out_point[3]) {
  ray_t s = {{origin[0], origin[1], origin[2]}, {dir[0], dir[1], dir[2]}};
  f3 cp;
  int r = cut((float)X, (float)Y, (float)Z, s, &cp);
  out_point[0] = cp.x; out_point[1] = cp.y; out_point[2] = cp.z;
  return r;
}

/* utility_ray.cl:112-117 exited_volume (strict comparisons: position == dim is still inside) */
static inline int exited_volume(const orc_render_params *p, f3 q) {
  int exited_max = ((float)p->X < q.x) | ((float)p->Y < q.y) | ((float)p->Z < q.z);
  int exited_min = (q.x < 0.0f) | (q.y < 0.0f) | (q.z < 0.0f);
  return exited_max | exited_min;
}

/* utility_ray.cl:126-138 get_event_and_value */
static inline int get_event_and_value(ctx_t *c, f3 position, i4 *value_at_event) {
  if (exited_volume(c->p, position)) return EV_EXIT;
  int32_t gradient = 0;
  if (c->tf_uses_gradient) {
    /* the six taps are only observable when the generated TF reads `gradient` */
    float g = v_length(gradient_nn(c, position));
    gradient = (int16_t)f2i(g); /* float -> short at the call */
  }
  int32_t value = vol_read_f(c, position.x, position.y, position.z);
  if (tf_eval(c->p->tf, value, gradient, value_at_event)) return EV_HIT;
  return EV_NONE;
}

/* utility_ray.cl:148-154 march */
static inline ray_t march(ctx_t *c, ray_t cur) {
  float signed_distance = (float)sdf_read_i(c, make_int(cur.origin));
  float step_size = f_max(signed_distance, 0.5f);
  ray_t r = {v_add(cur.origin, v_scale(cur.direction, step_size)), cur.direction};
  c->c[ORC_N_STEP]++;
  if (c->p->locality && c->in_bounce) {
    c->loc[ORC_LOC_STEPS]++;
    if (step_size <= 1.0f) c->loc[ORC_LOC_STEP_LE_1]++;
    if (step_size <= 2.0f) c->loc[ORC_LOC_STEP_LE_2]++;
    if (step_size <= 8.0f) c->loc[ORC_LOC_STEP_LE_8]++;
    if (step_size <= 32.0f) c->loc[ORC_LOC_STEP_LE_32]++;
  }
  return r;
}

/* instrumentation only: can the march from `pos` along `d` be PROVEN to leave the volume without a Hit within `budget`
 * steps, from the macro-cell table alone?  (3-D DDA over the cells, binary64) */
static int certify_exit(const orc_render_params *p, f3 pos, f3 d, int budget) {
  const double M = (double)p->macro_m;
  const int nx = (p->X + p->macro_m - 1) / p->macro_m, ny = (p->Y + p->macro_m - 1) / p->macro_m,
            nz = (p->Z + p->macro_m - 1) / p->macro_m;
  const double o[3] = {pos.x, pos.y, pos.z}, dir[3] = {d.x, d.y, d.z}, dim[3] = {(double)p->X, (double)p->Y, (double)p->Z};
  const int n[3] = {nx, ny, nz};
  int cell[3], step[3];
  double t_next[3], t_delta[3], t_exit = 1e30;
  for (int k = 0; k < 3; ++k) {
    if (!(o[k] >= 0.0 && o[k] <= dim[k]) || dir[k] != dir[k]) return 0;
    cell[k] = (int)(o[k] / M);
    if (cell[k] >= n[k]) cell[k] = n[k] - 1;
    if (dir[k] > 0.0) {
      step[k] = 1; t_delta[k] = M / dir[k]; t_next[k] = ((cell[k] + 1) * M - o[k]) / dir[k];
      const double te = (dim[k] - o[k]) / dir[k];
      if (te < t_exit) t_exit = te;
    } else if (dir[k] < 0.0) {
      step[k] = -1; t_delta[k] = -M / dir[k]; t_next[k] = (cell[k] * M - o[k]) / dir[k];
      const double te = (0.0 - o[k]) / dir[k];
      if (te < t_exit) t_exit = te;
    } else {
      step[k] = 0; t_delta[k] = 1e30; t_next[k] = 1e30;
    }
  }
  if (!(t_exit < 1e29)) return 0;
  if (p->cert_mode >= 5) {
    /* pyramid variants (estimates for a table with direction bins finer than an octant): the region every march from this cell can
     * cross when its direction's dominant axis is a and its two minor slopes |d_b / d_a|, |d_c / d_a| lie in the bin [lo, hi] of width
     * 1 / B (B = cert_mode - 4): in the slab n cells further along a, the cells q = floor(lo max(n - 1, 0)) .. ceil(1 + hi (n + 1)) - 1
     * along b (likewise c).  Step bound as in mode 3: the diagonal to the corner over the region's smallest SDF value. */
    const int B = p->cert_mode - 4;
    int a = 0;
    if (fabs(dir[1]) > fabs(dir[a])) a = 1;
    if (fabs(dir[2]) > fabs(dir[a])) a = 2;
    if (dir[a] == 0.0) return 0;
    const int b = (a + 1) % 3, cc = (a + 2) % 3;
    const int sa = dir[a] > 0.0 ? 1 : -1, sb = dir[b] >= 0.0 ? 1 : -1, sc = dir[cc] >= 0.0 ? 1 : -1;
    double slope_b = fabs(dir[b] / dir[a]), slope_c = fabs(dir[cc] / dir[a]);
    int bin_b = (int)(slope_b * B), bin_c = (int)(slope_c * B);
    if (bin_b >= B) bin_b = B - 1;
    if (bin_c >= B) bin_c = B - 1;
    const double lob = (double)bin_b / B, hib = (double)(bin_b + 1) / B, loc_ = (double)bin_c / B, hic = (double)(bin_c + 1) / B;
    int region_min = 127;
    for (int nn = 0;; ++nn) {
      const int ia = cell[a] + sa * nn;
      if (ia < 0 || ia >= n[a]) break;
      const int m1 = nn > 0 ? nn - 1 : 0;
      const int qb0 = (int)floor(lob * m1), qb1 = (int)ceil(1.0 + hib * (nn + 1)) - 1;
      const int qc0 = (int)floor(loc_ * m1), qc1 = (int)ceil(1.0 + hic * (nn + 1)) - 1;
      for (int qb = qb0; qb <= qb1; ++qb) {
        const int ib = cell[b] + sb * qb;
        if (ib < 0 || ib >= n[b]) continue;
        for (int qc = qc0; qc <= qc1; ++qc) {
          const int ic = cell[cc] + sc * qc;
          if (ic < 0 || ic >= n[cc]) continue;
          int idx[3];
          idx[a] = ia; idx[b] = ib; idx[cc] = ic;
          const int fm = p->macro_free_min[((int64_t)idx[2] * ny + idx[1]) * nx + idx[0]];
          if (fm < p->cert_min_free) return 0;
          if (fm < region_min) region_min = fm;
        }
      }
    }
    double diag2 = 0.0;
    for (int k = 0; k < 3; ++k) {
      const double e = dir[k] > 0.0 ? dim[k] - cell[k] * M : (dir[k] < 0.0 ? (cell[k] + 1) * M : 0.0);
      diag2 += e * e;
    }
    return (int)(sqrt(diag2) / (double)region_min) + 4 <= budget;
  }
  if (p->cert_mode != 0) {
    /* box / octant variants: no walk, a block of cells must be free with SDF values >= cert_min_free */
    int lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
      int far_cell;
      if (p->cert_mode >= 2) far_cell = dir[k] > 0.0 ? n[k] - 1 : (dir[k] < 0.0 ? 0 : cell[k]);
      else {
        double e = o[k] + (t_exit + 2.0) * dir[k];
        far_cell = (int)(e / M);
        if (e < 0.0) far_cell = 0;
        if (far_cell >= n[k]) far_cell = n[k] - 1;
      }
      lo[k] = cell[k] < far_cell ? cell[k] : far_cell;
      hi[k] = cell[k] < far_cell ? far_cell : cell[k];
    }
    int region_min = 127;
    for (int z = lo[2]; z <= hi[2]; ++z)
      for (int y = lo[1]; y <= hi[1]; ++y)
        for (int x = lo[0]; x <= hi[0]; ++x) {
          const int fm = p->macro_free_min[((int64_t)z * ny + y) * nx + x];
          if (fm < p->cert_min_free) return 0;
          if (fm < region_min) region_min = fm;
        }
    if (p->cert_mode == 3) {
      /* no t_exit: the longest path inside the box between the cell and the corner is its diagonal, every step is at least
       * region_min long; mode 4: the same with the step count compared against a fixed budget (a table bit) */
      double diag2 = 0.0;
      for (int k = 0; k < 3; ++k) {
        const double e = dir[k] > 0.0 ? dim[k] - cell[k] * M : (dir[k] < 0.0 ? (cell[k] + 1) * M : 0.0);
        diag2 += e * e;
      }
      return (int)(sqrt(diag2) / (double)region_min) + 4 <= budget;
    }
    if (p->cert_mode == 4) return (int)(t_exit / (double)region_min) + 4 <= budget;
    return (int)(t_exit / (double)p->cert_min_free) + 3 <= budget;
  }
  int s_min = 127;
  for (int guard = 0; guard < 4096; ++guard) {
    const int fm = p->macro_free_min[((int64_t)cell[2] * ny + cell[1]) * nx + cell[0]];
    if (fm == 0) return 0;
    if (fm < s_min) s_min = fm;
    int a = 0;
    if (t_next[1] < t_next[a]) a = 1;
    if (t_next[2] < t_next[a]) a = 2;
    if (t_next[a] > t_exit + 2.0) break;  /* past the boundary (2 voxels of slack for the rounding of the real march) */
    cell[a] += step[a];
    if (cell[a] < 0 || cell[a] >= n[a]) break;
    t_next[a] += t_delta[a];
  }
  /* every step of the march is at least s_min long: ceil(t_exit / s_min) + 2 steps certainly leave the volume */
  return (int)(t_exit / (double)s_min) + 3 <= budget;
}

/* utility_ray.cl:157-168 march_to_next_event */
static inline ray_t march_to_next_event(ctx_t *c, ray_t cur, int *event_type, i4 *value_at_event) {
  int internal_event = EV_NONE;
  int certified = 0;  /* instrumentation only */
  const int instrument = c->p->locality && c->p->macro_free_min && c->in_bounce;
  for (int i = 0; i < 70; ++i) {
    cur = march(c, cur);
    if (certified) c->loc[ORC_LOC_CERT_SAVED]++;
    internal_event = get_event_and_value(c, cur.origin, value_at_event);
    if (internal_event != EV_NONE) break;
    if (instrument && !certified) {
      const orc_render_params *p = c->p;
      const i4 q = make_int(cur.origin);
      const int sd = in_range(p, q.x, q.y, q.z) ? p->sdf[lin(p, q.x, q.y, q.z)] : 0;
      if (sd >= p->cert_t) {
        c->loc[ORC_LOC_CERT_TRIED]++;
        if (certify_exit(p, cur.origin, cur.direction, 70 - (i + 1))) {
          certified = 1;
          c->loc[ORC_LOC_CERT_GRANTED]++;
        }
      }
    }
  }
  if (certified && internal_event != EV_EXIT) c->loc[ORC_LOC_CERT_WRONG]++;
  *event_type = internal_event;
  return cur;
}

/* utility_environment_map.cl:3-13 sample_environment_map */
static inline void env_texel(f3 d, int32_t w, int32_t h, int32_t *oi, int32_t *oj) {
  float u = cr_atan2f(d.x, d.z);
  float v = cr_asinf(-d.y);
  u = u * 0.1591549431f;
  v = v * 0.318309886f;
  u = u + 0.5f;
  v = v + 0.5f;
  int32_t i = f2i(floorf(u * (float)w));
  int32_t j = f2i(floorf(v * (float)h));
  if (i < 0) i = 0;
  if (i > w - 1) i = w - 1;
  if (j < 0) j = 0;
  if (j > h - 1) j = h - 1;
  *oi = i; *oj = j;
}
void orc_env_texel(const float dir[3], int32_t env_w, int32_t env_h, int32_t out_ij[2]) {
  f3 d = {dir[0], dir[1], dir[2]};
  env_texel(d, env_w, env_h, &out_ij[0], &out_ij[1]);
}
static inline u4 sample_environment_map(ctx_t *c, f3 dir) {
  const orc_render_params *p = c->p;
  int32_t i, j;
  env_texel(dir, p->env_w, p->env_h, &i, &j);
  c->c[ORC_N_ENV]++;
  const uint8_t *t = p->env + ((int64_t)j * p->env_w + i) * 4;
  u4 r = {t[0], t[1], t[2], t[3]};
  return r;
}

/* ---- voxel cache (utility.cl:20-54, 93-121) ---- */
int64_t orc_cache_len(int32_t X, int32_t Y, int32_t Z) {
  return ((int64_t)X * Z * Y + (int64_t)X * Z + X + 1) * 4;
}
static inline int64_t cache_entry(const orc_render_params *p, i4 pos) {
  /* utility.cl:21 -- y-major, then z, then x; 64-bit here (the reference overflows int above ~812^3) */
  return (int64_t)p->X * p->Z * pos.y + (int64_t)p->X * pos.z + pos.x;
}
static inline int cache_entry_valid(const orc_render_params *p, int64_t e) {
  return e >= 0 && (e + 1) * 4 <= orc_cache_len(p->X, p->Y, p->Z);
}
/* utility.cl:20-31 atomic_allow_write_max */
static inline int atomic_allow_write_max(ctx_t *c, int64_t e, uint32_t max) {
  uint16_t *bv = c->p->cache + e * 4;
  int32_t *buffer = (int32_t *)bv;
  int16_t w = (int16_t)__atomic_load_n(&bv[3], __ATOMIC_RELAXED);
  if ((uint32_t)(int32_t)w > max) return 0;
  c->c[ORC_N_TOK]++;
  int32_t t = __atomic_fetch_add(buffer + 1, 0x00010000, __ATOMIC_RELAXED);
  if ((uint32_t)(t >> 16) < max) return 1;
  c->c[ORC_N_TOK]++;
  __atomic_fetch_sub(buffer + 1, 0x00010000, __ATOMIC_RELAXED);
  return 0;
}
/* utility.cl:39-54 atomic_buffer_volume_add4 */
static inline void atomic_buffer_volume_add4(ctx_t *c, int64_t e, u4 v) {
  int32_t *buffer = (int32_t *)(c->p->cache + e * 4);
  uint32_t r = v.x & 0xFFFFu, g = v.y & 0xFFFFu, b = v.z & 0xFFFFu, a = v.w & 0xFFFFu;
  uint32_t low = r + (g << 16), high = b + (a << 16);
  c->c[ORC_N_ADD]++;
  __atomic_fetch_add(buffer, (int32_t)low, __ATOMIC_RELAXED);
  __atomic_fetch_add(buffer + 1, (int32_t)high, __ATOMIC_RELAXED);
}
/* utility.cl:93-105 buffer_volume_read4 */
static inline u4 buffer_volume_read4(const orc_render_params *p, int64_t e) {
  const uint16_t *bv = p->cache + e * 4;
  u4 r = {bv[0], bv[1], bv[2], bv[3]};
  return r;
}

/* ray_marching.cl:82-99: integer mean, tone curve, truncation */
static inline u4 tone_map(u4 bv) {
  if (bv.w == 0) { u4 z = {0, 0, 0, 1}; return z; } /* unreachable in the reference: count >= 1 */
  uint32_t r = bv.x / bv.w, g = bv.y / bv.w, b = bv.z / bv.w;
  const float inv_gamma = 1.0f / 1.77777777f;
  const float brightness = 4.0f;
  float fx = (float)r / 255.0f, fy = (float)g / 255.0f, fz = (float)b / 255.0f;
  fx = cr_powf(fx * brightness, inv_gamma);
  fy = cr_powf(fy * brightness, inv_gamma);
  fz = cr_powf(fz * brightness, inv_gamma);
  fx = fx * 255.0f; fy = fy * 255.0f; fz = fz * 255.0f;
  u4 out = {f2u(fx), f2u(fy), f2u(fz), 1};
  return out;
}

/* write_imageui on CL_UNSIGNED_INT8: saturate */
static inline void frame_write(const orc_render_params *p, uint32_t x, uint32_t y, u4 c) {
  if (!p->frame || (int32_t)x >= p->frame_w || (int32_t)y >= p->frame_h) return;
  uint8_t *px = p->frame + ((int64_t)y * p->frame_w + x) * 4;
  px[0] = c.x > 255 ? 255 : (uint8_t)c.x;
  px[1] = c.y > 255 ? 255 : (uint8_t)c.y;
  px[2] = c.z > 255 ? 255 : (uint8_t)c.z;
  px[3] = c.w > 255 ? 255 : (uint8_t)c.w;
}

/* ray_marching.cl:10-101 compute_light.  Returns w == 0 on a miss. */
static u4 compute_light(ctx_t *c, ray_t surface_ray, int32_t random_seed, int64_t *hit_entry,
                        u4 *contribution) {
  const orc_render_params *p = c->p;
  const int dist_count = 2;
  const int path_length = 3;

  ray_t current_ray = surface_ray;
  i4 current_color = {0, 0, 0, 0};
  ray_t hit_information;
  memset(&hit_information, 0, sizeof hit_information);
  u4 buffer_value = {0, 0, 0, 0};
  const int information_dev = 1;

  int ray_event;
  const uint64_t sdf0 = c->c[ORC_N_SDF], vol0 = c->c[ORC_N_VOL];
  current_ray = march_to_next_event(c, surface_ray, &ray_event, &current_color);
  c->c[ORC_N_SDF_PRIMARY] += c->c[ORC_N_SDF] - sdf0;
  c->c[ORC_N_VOL_PRIMARY] += c->c[ORC_N_VOL] - vol0;
  if (ray_event != EV_HIT) {
    u4 z = {0, 0, 0, 0};
    return z;
  }
  c->c[ORC_N_HIT]++;
  hit_information = current_ray;
  const int64_t entry = cache_entry(p, make_int(current_ray.origin));
  *hit_entry = entry;

  int granted;
  if (p->mode == ORC_MODE_VOXEL_CACHE)
    granted = cache_entry_valid(p, entry) && atomic_allow_write_max(c, entry, 256u * information_dev);
  else
    granted = 1; /* image-space mode: no token, every sample contributes */

  if (granted) {
    const uint64_t vol1 = c->c[ORC_N_VOL];
    const f3 normal = v_neg(v_normalize(gradient_nn(c, current_ray.origin)));
    c->c[ORC_N_VOL_PRIMARY] += c->c[ORC_N_VOL] - vol1;
    float r_energy = (float)current_color.x / 255.0f;
    float g_energy = (float)current_color.y / 255.0f;
    float b_energy = (float)current_color.z / 255.0f;

    {
      i4 hv = make_int(hit_information.origin);
      c->hit_v[0] = hv.x; c->hit_v[1] = hv.y; c->hit_v[2] = hv.z;
      c->in_bounce = 1;
    }
    const uint64_t item_start = c->c[ORC_N_STEP];
    for (int o = 1; o <= dist_count; ++o) {
      const uint64_t ray_start = c->c[ORC_N_STEP];
      c->have_prev = 0;
      current_ray = bounce_fake_reflectance(c, hit_information, normal, random_seed + o,
                                            ((float)current_color.w) / 255.0f);
      current_ray.origin = v_add(current_ray.origin, v_scale(normal, 2.0f));
      float atten = fabsf(v_dot(current_ray.direction, normal));

      for (int i = 8; i <= 7 + path_length; ++i) {
        current_ray = march_to_next_event(c, current_ray, &ray_event, &current_color);
        if (ray_event == EV_EXIT) {
          float factor = 8.0f / (float)i;
          u4 light_map = sample_environment_map(c, current_ray.direction);
          /* uint += float : operands promoted to float, sum truncated back to uint */
          buffer_value.x = f2u((float)buffer_value.x +
                               atten * r_energy * (float)light_map.x * factor / (float)information_dev);
          buffer_value.y = f2u((float)buffer_value.y +
                               atten * g_energy * (float)light_map.y * factor / (float)information_dev);
          buffer_value.z = f2u((float)buffer_value.z +
                               atten * b_energy * (float)light_map.z * factor / (float)information_dev);
          break;
        } else if (ray_event == EV_HIT) {
          c->c[ORC_N_HIT_BOUNCE]++;
          const f3 normal2 = v_neg(v_normalize(gradient_nn(c, current_ray.origin)));
          current_ray = bounce_fake_reflectance(c, current_ray, normal2, random_seed + o + i,
                                                ((float)current_color.w) / 255.0f);
          current_ray.origin = v_add(current_ray.origin, v_scale(normal2, 2.0f));
          atten *= fabsf(v_dot(current_ray.direction, normal2));
          r_energy *= (float)current_color.x / 255.0f;
          g_energy *= (float)current_color.y / 255.0f;
          b_energy *= (float)current_color.z / 255.0f;
        }
      }
      if (p->locality) {
        const uint64_t n = c->c[ORC_N_STEP] - ray_start;
        c->loc[ORC_LOC_HIST_RAY + (n > 255 ? 255 : n)]++;
      }
    }
    if (p->locality) {
      const uint64_t n = c->c[ORC_N_STEP] - item_start;
      c->loc[ORC_LOC_HIST_ITEM + (n > 511 ? 511 : n)]++;
    }
    c->in_bounce = 0;
    buffer_value.x /= (uint32_t)dist_count;
    buffer_value.y /= (uint32_t)dist_count;
    buffer_value.z /= (uint32_t)dist_count;
    buffer_value.w /= (uint32_t)dist_count;
    contribution->x = buffer_value.x & 0xFFFFu;
    contribution->y = buffer_value.y & 0xFFFFu;
    contribution->z = buffer_value.z & 0xFFFFu;
    contribution->w = 1;
    if (p->mode == ORC_MODE_VOXEL_CACHE) {
      atomic_buffer_volume_add4(c, entry, buffer_value);
    } else {
      float *a = p->accum + ((int64_t)c->gy * p->launch_w + c->gx) * 4;
      c->c[ORC_N_ADD]++;
      a[0] += (float)contribution->x;
      a[1] += (float)contribution->y;
      a[2] += (float)contribution->z;
      a[3] += 1.0f;
    }
  }

  if (p->mode == ORC_MODE_VOXEL_CACHE) {
    if (!cache_entry_valid(p, entry)) { u4 z = {0, 0, 0, 1}; return z; }
    c->c[ORC_N_READ]++;
    buffer_value = buffer_volume_read4(p, entry);
  } else {
    const float *a = p->accum + ((int64_t)c->gy * p->launch_w + c->gx) * 4;
    c->c[ORC_N_READ]++;
    buffer_value.x = (uint32_t)a[0]; buffer_value.y = (uint32_t)a[1];
    buffer_value.z = (uint32_t)a[2]; buffer_value.w = (uint32_t)a[3];
  }
  return tone_map(buffer_value);
}


/* ---- ambient occlusion: the reference's alternate shading function ------------------------------------------
 * utility_sampling.cl:25-36 get_hemisphere_direction: like the reflective variant without the roughness blend */
static inline f3 hemisphere_direction(uint32_t gx, uint32_t gy, f3 normal, int32_t seed) {
  uint32_t useed = (uint32_t)seed + (gx + 1u) * (gy + 1u);
  int32_t rx = (int32_t)orc_hash(useed * 0x182205bdu);
  int32_t ry = (int32_t)orc_hash(useed * 0xe8d052f3u);
  int32_t rz = (int32_t)orc_hash(useed * 0xf1981dcfu);
  f3 direction = {(float)((rx % 2048) - 1024), (float)((ry % 2048) - 1024),
                  (float)((rz % 2048) - 1024)};
  float decider = v_dot(direction, normal);
  return v_normalize(v_scale(direction, decider));
}

/* utility.cl:123-135 / :147-159 buffer_volume_read(f) / buffer_volume_write(f): the 2-channel view of the cache,
 * idx = (X*Z*y + X*z + x) * 2 ushorts = {samples, occluded} per voxel */
int64_t orc_ao_cache_len(int32_t X, int32_t Y, int32_t Z) {
  return ((int64_t)X * Z * Y + (int64_t)X * Z + X + 1) * 2;
}
static inline int ao_entry_valid(const orc_render_params *p, int64_t e) {
  return e >= 0 && (e + 1) * 2 <= orc_ao_cache_len(p->X, p->Y, p->Z);
}

/* ray_marching.cl:104-149 compute_ao.  The reference's read-modify-write of the cache entry is not atomic; this
 * restatement runs the pixels one after the other, which is one legal outcome of that race, and the entry is
 * order-independent while its sample count stays below the cap of 100.  Returns {v, v, v, 0} with
 * v = 2 * (100 - occluded); note w == 0: plugged into render() as it stands (:186-195) every AO pixel would be
 * painted with the environment colour, so *shade carries the value for the caller's resolve instead. */
static u4 compute_ao(ctx_t *c, ray_t surface_ray, int32_t random_seed, int64_t *hit_entry, u4 *contribution) {
  const orc_render_params *p = c->p;
  ray_t current_ray = surface_ray;
  ray_t hit_information;
  memset(&hit_information, 0, sizeof hit_information);
  i4 color = {0, 0, 0, 0};
  int ray_event;
  const uint64_t sdf0 = c->c[ORC_N_SDF], vol0 = c->c[ORC_N_VOL];
  current_ray = march_to_next_event(c, surface_ray, &ray_event, &color);
  c->c[ORC_N_SDF_PRIMARY] += c->c[ORC_N_SDF] - sdf0;
  c->c[ORC_N_VOL_PRIMARY] += c->c[ORC_N_VOL] - vol0;
  if (ray_event != EV_HIT) {
    u4 z = {0, 0, 0, 0};
    return z; /* `return 0.0f;` */
  }
  c->c[ORC_N_HIT]++;
  const int64_t entry = cache_entry(p, make_int(current_ray.origin));
  *hit_entry = entry;
  if (!ao_entry_valid(p, entry)) { /* outside the allocation: nothing recorded, nothing occluded */
    u4 z = {200, 200, 200, 0};
    return z;
  }
  uint16_t *bv = p->cache + entry * 2;
  uint32_t samples = bv[0], occluded = bv[1];
  c->c[ORC_N_READ]++;
  hit_information = current_ray;
  if (samples < 100u) {
    samples += 1u;
    const uint64_t vol1 = c->c[ORC_N_VOL];
    const f3 normal = v_neg(v_normalize(gradient_nn(c, current_ray.origin)));
    c->c[ORC_N_VOL_PRIMARY] += c->c[ORC_N_VOL] - vol1;
    /* utility_ray.cl:100-103 ray_bounce */
    ray_t b = {v_add(current_ray.origin, current_ray.direction), hemisphere_direction(c->gx, c->gy, normal, random_seed)};
    current_ray = b;
    for (int k = 0; k < 7; ++k) current_ray = march(c, current_ray); /* seven unclassified steps */
    march_to_next_event(c, current_ray, &ray_event, &color);
    if (ray_event == EV_HIT) occluded += 1u;
    bv[0] = (uint16_t)samples;
    bv[1] = (uint16_t)occluded;
    c->c[ORC_N_ADD]++;
    contribution->x = (ray_event == EV_HIT) ? 1u : 0u;
    contribution->w = 1u;
  }
  c->c[ORC_N_READ]++;
  const uint32_t v = (100u - bv[1]) * 2u;
  u4 r = {v, v, v, 0};
  (void)hit_information;
  return r;
}

/* ray_marching.cl:152-199 render, one work-item */
static void render_pixel(ctx_t *c, uint32_t x, uint32_t y) {
  const orc_render_params *p = c->p;
  c->gx = x; c->gy = y;
  const int64_t pix = (int64_t)y * p->launch_w + x;
  if (p->hit_index) p->hit_index[pix] = -1;
  if (p->contrib) memset(p->contrib + pix * 4, 0, 4 * sizeof(uint32_t));

  u4 zero = {0, 0, 0, 0};
  frame_write(p, x, y, zero);

  f3 cam_o = {p->cam_pos[0], p->cam_pos[1], p->cam_pos[2]};
  f3 cam_d = {p->cam_dir[0], p->cam_dir[1], p->cam_dir[2]};
  ray_t vray = generate_ray(cam_o, cam_d, (int32_t)x, (int32_t)y, p->frame_w, p->frame_h);

  int cut_ok;
  f3 cut_point;
  const float dx = (float)p->X, dy = (float)p->Y, dz = (float)p->Z;
  if (!(limits(vray.origin.x, dx) && limits(vray.origin.y, dy) && limits(vray.origin.z, dz))) {
    cut_ok = cut(dx, dy, dz, vray, &cut_point);
  } else {
    cut_ok = 1;
    cut_point = vray.origin;
  }

  if (!cut_ok) {
    u4 e = sample_environment_map(c, vray.direction);
    c->c[ORC_N_ENV_PRIMARY]++;
    u4 color = {e.x, e.y, e.z, 200};
    frame_write(p, x, y, color);
    return;
  }

  ray_t surface_ray = {cut_point, vray.direction};
  int64_t hit_entry = -1;
  u4 contribution = {0, 0, 0, 0};
  u4 f;
  if (p->shading == ORC_SHADE_AO) {
    f = compute_ao(c, surface_ray, p->seed, &hit_entry, &contribution);
    /* compute_ao's result has w == 0 (see there); a hit pixel is shown as {v, v, v, 1} like compute_light's */
    if (hit_entry != -1) f.w = 1;
  } else {
    f = compute_light(c, surface_ray, p->seed, &hit_entry, &contribution);
  }
  if (f.w == 0) {
    u4 e = sample_environment_map(c, vray.direction);
    c->c[ORC_N_ENV_PRIMARY]++;
    u4 color = {e.x, e.y, e.z, 200};
    frame_write(p, x, y, color);
    return;
  }
  if (p->hit_index) p->hit_index[pix] = hit_entry;
  if (p->contrib) {
    uint32_t *q = p->contrib + pix * 4;
    q[0] = contribution.x; q[1] = contribution.y; q[2] = contribution.z; q[3] = contribution.w;
  }
  frame_write(p, x, y, f);
}

static int tf_uses_gradient(const orc_tf *tf) {
  for (int k = 0; k < tf->n; ++k)
    if (tf->rules[k].use_gradient) return 1;
  return 0;
}

static inline int tile_owned(const orc_render_params *p, uint32_t x, uint32_t y) {
  if (p->tile_world <= 1) return 1;
  return (int32_t)(((x >> 3) + (y >> 3)) % (uint32_t)p->tile_world) == p->tile_rank;
}

int orc_render(const orc_render_params *p) {
  if (!p || !p->volume || !p->sdf || !p->env || !p->tf) return -1;
  if ((p->mode == ORC_MODE_VOXEL_CACHE || p->shading == ORC_SHADE_AO) && !p->cache) return -1;
  if (p->mode == ORC_MODE_IMAGE_SPACE && !p->accum) return -1;
  const int uses_g = tf_uses_gradient(p->tf);
  uint64_t total[ORC_N_COUNTERS];
  memset(total, 0, sizeof total);
  int nthreads = p->threads > 0 ? p->threads : 1;
  (void)nthreads;
#pragma omp parallel num_threads(nthreads)
  {
    ctx_t c;
    memset(&c, 0, sizeof c);
    c.p = p;
    c.tf_uses_gradient = uses_g;
#pragma omp for schedule(dynamic, 4)
    for (int32_t y = 0; y < p->launch_h; ++y)
      for (int32_t x = 0; x < p->launch_w; ++x)
        if (tile_owned(p, (uint32_t)x, (uint32_t)y)) render_pixel(&c, (uint32_t)x, (uint32_t)y);
#pragma omp critical
    {
      for (int k = 0; k < ORC_N_COUNTERS; ++k) total[k] += c.c[k];
      if (p->locality)
        for (int k = 0; k < ORC_LOC_TOTAL; ++k) p->locality[k] += c.loc[k];
    }
  }
  if (p->counters)
    for (int k = 0; k < ORC_N_COUNTERS; ++k) p->counters[k] += total[k];
  return 0;
}

int orc_resolve(const orc_render_params *p) {
  if (!p || !p->hit_index || !p->frame) return -1;
  for (int32_t y = 0; y < p->launch_h; ++y)
    for (int32_t x = 0; x < p->launch_w; ++x) {
      const int64_t pix = (int64_t)y * p->launch_w + x;
      if (!tile_owned(p, (uint32_t)x, (uint32_t)y)) continue;
      const int64_t e = p->hit_index[pix];
      if (e < 0) continue;
      u4 bv;
      if (p->shading == ORC_SHADE_AO) {
        const uint32_t v = ao_entry_valid(p, e) ? (100u - p->cache[e * 2 + 1]) * 2u : 200u;
        u4 shade = {v, v, v, 1};
        frame_write(p, (uint32_t)x, (uint32_t)y, shade);
        continue;
      }
      if (p->mode == ORC_MODE_VOXEL_CACHE) {
        if (!cache_entry_valid(p, e)) continue;
        bv = buffer_volume_read4(p, e);
      } else {
        const float *a = p->accum + pix * 4;
        bv.x = (uint32_t)a[0]; bv.y = (uint32_t)a[1]; bv.z = (uint32_t)a[2]; bv.w = (uint32_t)a[3];
      }
      frame_write(p, (uint32_t)x, (uint32_t)y, tone_map(bv));
    }
  return 0;
}

/* buffer_reset.cl:3-13 */
void orc_buffer_reset(uint16_t *cache, int32_t X, int32_t Y, int32_t Z) {
  memset(cache, 0, (size_t)orc_cache_len(X, Y, Z) * sizeof(uint16_t));
}

/* app/common.hpp:5-12, 48-56: Position3D(alpha, beta, gamma=0, base={1,0,0}); double math, stored to
 * float, then normalised with sqrtf of a double sum and float division (common.hpp:44-56) */
void orc_camera_direction(double alpha, double beta, float out[3]) {
  const double gamma = 0.0;
  const float base[3] = {1.0f, 0.0f, 0.0f};
  float val[3];
  val[0] = (float)((cos(alpha) * cos(beta)) * base[0] +
                   (cos(alpha) * sin(beta) - sin(alpha) * cos(gamma)) * base[1] +
                   (cos(alpha) * sin(beta) * cos(gamma) + sin(alpha) * sin(gamma)) * base[2]);
  val[1] = (float)((-sin(beta)) * base[0] + (cos(beta) * sin(gamma)) * base[1] +
                   (cos(beta) * cos(gamma)) * base[2]);
  val[2] = (float)((sin(alpha) * cos(beta)) * base[0] +
                   (sin(alpha) * sin(beta) * sin(gamma) + cos(alpha) * cos(gamma)) * base[1] +
                   (sin(alpha) * sin(beta) * cos(gamma) - cos(alpha) * sin(gamma)) * base[2]);
  /* length(): sqrtf(pow(v0,2)+pow(v1,2)+pow(v2,2)) returned as double; operator/(float) */
  double len = (double)sqrtf((float)(pow(val[0], 2) + pow(val[1], 2) + pow(val[2], 2)));
  float flen = (float)len;
  out[0] = val[0] / flen;
  out[1] = val[1] / flen;
  out[2] = val[2] / flen;
}
