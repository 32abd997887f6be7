/*
 * oracle/orc.h -- CPU restatement of the reference's volumetric path-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liborc.so, and there only as the checker /
 * reported baseline.  The product (libclwhip.so) never links or calls anything here.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - SDF builder (orc_sdf_build): PINNED bit-exactly by the reference's own golden vector
 *     tests/sdf/values.x for tests/sdf/testdata.nrrd  (tests/test_oracle_sdf.py).
 *   - render / compute_light / RNG / cut / env lookup / voxel cache: PARITY UNPINNED --
 *     the reference holds no golden image or known-answer vector for them and its OpenCL
 *     kernels cannot run in this image (no OpenCL CPU runtime).  Every function cites the
 *     reference file:line it restates so the restatement can be audited by reading.
 *
 * Implementation-defined OpenCL behaviour is fixed as follows (DESIGN.md "Semantics"):
 *   - integer images + CLK_FILTER_LINEAR  -> nearest: texel = floor(coord), out of range
 *     (or NaN) -> border 0 for CLK_ADDRESS_CLAMP; env map: floor(u*w) clamped to edge.
 *   - no FMA contraction anywhere; every float op is a single IEEE-754 binary32 operation
 *     in the order written.
 *   - normalize(v) = v / sqrtf((v.x*v.x + v.y*v.y) + v.z*v.z)  (three IEEE divisions).
 *   - atan2, asin, pow: correctly rounded binary32 results, obtained by evaluating in
 *     binary64 and rounding once.
 *   - float -> int / uint conversions truncate toward zero, saturate, NaN -> 0.
 *   - min(a,b) = (b < a) ? b : a ;  max(a,b) = (a < b) ? b : a.
 */
#ifndef ORC_H
#define ORC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_TF_MAX_RULES 16

/* One transfer-function rule:  value in [v_lo, v_hi]  (&& gradient in [g_lo, g_hi] if use_gradient).
 * The reference builds OpenCL-C source (app/ui.cpp:160-168, app/tf_part.cpp:55-79); comparing a
 * short against printed decimal literals is a comparison of integers against reals, so the
 * thresholds are pre-rounded to the equivalent inclusive integer bounds by the caller. */
typedef struct {
  int32_t v_lo, v_hi;
  int32_t g_lo, g_hi;
  int32_t use_gradient;   /* rule reads `gradient` */
  int32_t writes_color;   /* rule assigns *color (rectangle form) or not (`return (value > 800);`) */
  int32_t terminal;       /* `return (cond);` form: evaluation stops here whether or not it matched */
  int32_t color[4];       /* r,g,b,a(=roughness) in 0..255 */
} orc_tf_rule;

typedef struct {
  int32_t n;
  orc_tf_rule rules[ORC_TF_MAX_RULES];
} orc_tf;

/* counters[]: exact texel / atomic counts for the algorithmic-bytes model */
enum {
  ORC_N_SDF = 0,   /* SDF texel fetches (1 B each) */
  ORC_N_VOL = 1,   /* volume texel fetches (2 B each) */
  ORC_N_ENV = 2,   /* env texel fetches (4 B each) */
  ORC_N_TOK = 3,   /* token atomics (4 B RMW each; the undo sub counts too) */
  ORC_N_ADD = 4,   /* cache adds (2 x 4 B) */
  ORC_N_READ = 5,  /* cache reads (8 B) */
  ORC_N_HIT = 6,   /* pixel-samples whose primary march ended in Hit */
  ORC_N_STEP = 7,  /* march steps */
  /* the part of N_SDF / N_VOL / N_ENV that belongs to the camera-dependent, seed-independent phase: the primary
   * march (ray_marching.cl:33), the primary hit's normal (:42) and the environment lookups of miss pixels
   * (:172-178, :188-195).  The MI355X build performs it once per camera (k_primary); the rest is the bounce phase. */
  ORC_N_SDF_PRIMARY = 8,
  ORC_N_VOL_PRIMARY = 9,
  ORC_N_ENV_PRIMARY = 10,
  ORC_N_HIT_BOUNCE = 11, /* Hit events of the distribution rays (what k_bounce pays one 8-byte hit record for) */
  ORC_N_COUNTERS = 12
};

enum { ORC_MODE_VOXEL_CACHE = 0, ORC_MODE_IMAGE_SPACE = 1 };
/* which of the reference's two shading functions `render` calls (ray_marching.cl:186 calls compute_light; compute_ao,
 * :104-149, is kept in the file as the alternate the authors swapped in by editing that line) */
enum { ORC_SHADE_LIGHT = 0, ORC_SHADE_AO = 1 };

typedef struct {
  const int16_t *volume;      /* x-fastest [z][y][x] */
  int32_t X, Y, Z;
  const int8_t *sdf;          /* same layout */
  const uint8_t *env;         /* RGBA8, row-major */
  int32_t env_w, env_h;
  uint16_t *cache;            /* voxel cache, orc_cache_len(X,Y,Z) ushorts */
  uint8_t *frame;             /* RGBA8 frame image, frame_w x frame_h */
  int32_t frame_w, frame_h;   /* image dims (what get_image_width(frame) returns) */
  int32_t launch_w, launch_h; /* NDRange global size */
  float cam_pos[3];
  float cam_dir[3];
  int32_t seed;
  const orc_tf *tf;
  int32_t mode;
  float *accum;               /* image-space mode: float4 per pixel {r,g,b,count} */
  int64_t *hit_index;         /* optional, per pixel: cache entry index (voxel units) or -1 */
  uint32_t *contrib;          /* optional, per pixel: {r,g,b,granted} of this pass */
  uint64_t *counters;         /* optional, ORC_N_COUNTERS totals, accumulated */
  int32_t tile_rank, tile_world; /* image-tile partition: 8x8 tiles, owner = (tx+ty) % world */
  int32_t threads;            /* OpenMP threads (1 = deterministic sequential pixel order) */
  int32_t shading;            /* ORC_SHADE_LIGHT (default) or ORC_SHADE_AO */
  /* optional instrumentation of the bounce phase's march steps (tools/step_locality.py): */
  uint64_t *locality;         /* ORC_LOC_COUNT totals, accumulated; NULL = off */
  const uint8_t *uniform4;    /* optional, one byte per 4x4x4 sub-brick [z/4][y/4][x/4]: 1 = all 64 step bytes equal */
  /* optional, for the same instrumentation: the "exit certificate" experiment (tools/exit_certificate.py).  One byte per
   * macro cell of macro_m^3 voxels [z][y][x]: 0 = the cell (dilated by 2 voxels) holds an event voxel or a non-positive SDF
   * value, else the smallest SDF value in it.  After a bounce-phase step whose next SDF value is >= cert_t the oracle walks the
   * cells from the new position to the volume's boundary: all non-zero and few enough steps left => the march must end in
   * Exit_volume without a Hit, and every later step of that march is counted as avoidable. */
  const uint8_t *macro_free_min;
  int32_t macro_m, cert_t;
  int32_t cert_mode;          /* 0: walk the cells the ray crosses; 1: every cell of the box spanned by the ray's cell and its exit cell;
                                 2: every cell of the octant region from the ray's cell to the volume corner it heads for;
                                 3: as 2, the step count bounded by the region's diagonal / its smallest SDF value (no exit distance:
                                    what k_bounce's table holds); 4: as 2, bounded by exit distance / the region's smallest SDF value */
  int32_t cert_min_free;      /* modes 1-4: a cell counts as free only if its smallest SDF value is at least this */
} orc_render_params;

/* locality[]: where the step fetches of the distribution rays fall (every fetch of a step byte = one SDF texel read) */
enum {
  ORC_LOC_FETCHES = 0,      /* step-byte fetches of the bounce phase that land inside the volume */
  ORC_LOC_SAME_SUB4 = 1,    /* ... in the same 4x4x4 sub-brick (one 64-B line of step bytes) as the ray's previous fetch */
  ORC_LOC_SAME_BRICK8 = 2,  /* ... in the same 8x8x8 brick */
  ORC_LOC_NEAR_8 = 3,       /* ... within Chebyshev distance 8 / 16 / 32 / 64 voxels of the sample's PRIMARY hit */
  ORC_LOC_NEAR_16 = 4,
  ORC_LOC_NEAR_32 = 5,
  ORC_LOC_NEAR_64 = 6,
  ORC_LOC_UNIFORM4 = 7,     /* ... in a sub-brick whose 64 step bytes are all equal (uniform4 given) */
  ORC_LOC_STEP_LE_1 = 8,    /* march steps of the bounce phase by length: <= 1, <= 2, <= 8, <= 32 voxels (cumulative) */
  ORC_LOC_STEP_LE_2 = 9,
  ORC_LOC_STEP_LE_8 = 10,
  ORC_LOC_STEP_LE_32 = 11,
  ORC_LOC_STEPS = 12,       /* march steps of the bounce phase */
  ORC_LOC_CERT_TRIED = 13,  /* exit certificates attempted / granted; steps (= step-byte fetches) that followed a granted one */
  ORC_LOC_CERT_GRANTED = 14,
  ORC_LOC_CERT_SAVED = 15,
  ORC_LOC_COUNT = 16,
  /* followed, in the same array, by two histograms of march steps (the dependent chain a lane walks):
   * [ORC_LOC_HIST_RAY + n]  distribution rays with n steps (n clamped to 255; a ray is at most 3 marches = 210 steps)
   * [ORC_LOC_HIST_ITEM + n] samples (both distribution rays) with n steps (n clamped to 511) */
  ORC_LOC_HIST_RAY = 16,
  ORC_LOC_HIST_ITEM = 16 + 256,
  ORC_LOC_CERT_WRONG = 16 + 256 + 512, /* granted certificates whose march did NOT end in Exit_volume (must stay 0) */
  ORC_LOC_TOTAL = 16 + 256 + 512 + 8
};

/* number of ushorts the voxel cache needs so that the reference's latent one-row overrun
 * (position == dim, opencl_kernels/utility.cl:21 with utility_ray.cl:112-117) stays in bounds */
int64_t orc_cache_len(int32_t X, int32_t Y, int32_t Z);
/* ushorts the 2-channel {samples, occluded} view of the cache needs (compute_ao, utility.cl:123-159); half of the above */
int64_t orc_ao_cache_len(int32_t X, int32_t Y, int32_t Z);

/* opencl_kernels/ray_marching.cl:152-199 for every work-item of the NDRange */
int orc_render(const orc_render_params *p);

/* deterministic resolve: for every pixel with hit_index >= 0 read the cache (or accum) and apply
 * ray_marching.cl:82-99; miss pixels keep what orc_render wrote (env colour, alpha 200) */
int orc_resolve(const orc_render_params *p);

/* app/signed_distance_field.cpp:7-35 + opencl_kernels/signed_distance_field.cl
 * layer_counts (optional, >= 130 ints) receives the atomic counter after each launch */
int orc_sdf_build(const int16_t *volume, int32_t X, int32_t Y, int32_t Z, const orc_tf *tf,
                  int8_t *sdf_out, int32_t *n_launches, int32_t *layer_counts);

/* opencl_kernels/buffer_reset.cl:3-13 */
/* volume_filter.cl:5-11 + utility_filter.cl:38-62: 5x5x5 bilateral filter of a short volume (orc_filter.c) */
void orc_bilateral_filter(const int16_t *volume, int32_t X, int32_t Y, int32_t Z, int16_t *out);

void orc_buffer_reset(uint16_t *cache, int32_t X, int32_t Y, int32_t Z);

/* app/common.hpp:5-12 Position3D(alpha, beta, 0, {1,0,0}) */
void orc_camera_direction(double alpha, double beta, float out[3]);

/* single-function probes for unit tests */
uint32_t orc_hash(uint32_t seed);
void orc_hemisphere_reflective(const float normal[3], int32_t seed, uint32_t gx, uint32_t gy,
                               float roughness, float out[3]);
void orc_generate_ray(const float cam_pos[3], const float cam_dir[3], int32_t x, int32_t y,
                      int32_t x_total, int32_t y_total, float out_dir[3]);
int orc_cut(int32_t X, int32_t Y, int32_t Z, const float origin[3], const float dir[3],
            float out_point[3]);
void orc_env_texel(const float dir[3], int32_t env_w, int32_t env_h, int32_t out_ij[2]);
int orc_tf_eval(const orc_tf *tf, int32_t value, int32_t gradient, int32_t color[4]);

#ifdef __cplusplus
}
#endif
#endif
