/*
 * clwh.h -- C ABI of libclwhip.so: the thin HIP shim that replaces the reference's
 * `opencl_wrapper` (clw_*) layer for the volumetric path-tracing hot path on MI355X (gfx950).
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * repository).  Conventions:
 *   - plain C: opaque handles, pointers and sizes; no exceptions, no callbacks;
 *   - every function returns an int status (CLWH_OK == 0); clwh_strerror() names it;
 *   - a handle is owned by whoever created it and is released explicitly; host pointers are
 *     only borrowed for the duration of a call;
 *   - one in-order HIP stream per context; push/pull are blocking like the reference's
 *     CL_TRUE transfers; launches are asynchronous and stream-ordered;
 *   - single host thread per context (the reference's ui::run loop).
 * The fail-hard convention of the reference (print + exit(1), clw_helper.hpp:293-309) lives in
 * the header-only C++ wrappers (include/clw_*.hpp), not in this ABI.
 */
#ifndef CLWH_H
#define CLWH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clwh_ctx clwh_ctx;       /* replaces clw_context        (opencl_wrapper/include/clw_context.hpp:5-28) */
typedef struct clwh_mem clwh_mem;       /* replaces cl_mem of clw_vector / clw_image */
typedef struct clwh_kernel clwh_kernel; /* replaces cl_program + cl_kernel of clw_function */

enum clwh_status {
  CLWH_OK = 0,
  CLWH_ERR_INVALID_VALUE = 1,
  CLWH_ERR_NO_DEVICE = 2,
  CLWH_ERR_OUT_OF_MEMORY = 3,
  CLWH_ERR_HIP = 4,            /* a HIP runtime call failed; clwh_last_hip_error() has the code */
  CLWH_ERR_UNKNOWN_KERNEL = 5, /* (file, entry) is not one of the precompiled hot-path kernels */
  CLWH_ERR_TF_UNSUPPORTED = 6, /* the transfer-function source neither parses as rules nor compiles with hiprtc */
  CLWH_ERR_BAD_ARGS = 7,       /* wrong number / kind of kernel arguments */
  CLWH_ERR_BAD_NDRANGE = 8,    /* global not a multiple of local, zero size, ... */
  CLWH_ERR_SIZE_MISMATCH = 9,
  CLWH_ERR_INTERNAL_OVERFLOW = 10 /* a device-side work buffer overflowed; results of the last render are incomplete */
};

/* element kinds of an image channel: (signedness, sizeof) as clw_image picks them
 * (opencl_wrapper/include/clw_image.hpp:68-110) */
enum clwh_elem_kind {
  CLWH_ELEM_S8 = 0, CLWH_ELEM_S16 = 1, CLWH_ELEM_S32 = 2,
  CLWH_ELEM_U8 = 3, CLWH_ELEM_U16 = 4, CLWH_ELEM_U32 = 5,
  CLWH_ELEM_F32 = 6
};

enum clwh_mem_flags { CLWH_MEM_READ_WRITE = 0, CLWH_MEM_READ_ONLY = 1 };

/* ---- context: clw_context::clw_context / ~clw_context (opencl_wrapper/src/clw_context.cpp:38-83).
 * The reference takes platform[0]/device[0]; here the HIP device ordinal is explicit. */
int clwh_ctx_create(int device, clwh_ctx **out);
/* same, but enqueue on a stream the caller owns (e.g. PyTorch's current stream); NULL = default */
int clwh_ctx_create_on_stream(int device, void *hip_stream, clwh_ctx **out);
int clwh_ctx_destroy(clwh_ctx *ctx);
/* clFinish on the context's queue */
int clwh_ctx_finish(clwh_ctx *ctx);
void *clwh_ctx_stream(clwh_ctx *ctx);
int clwh_ctx_device(clwh_ctx *ctx);

/* ---- memory objects.
 * clwh_mem_create    replaces clCreateBuffer in clw_vector's ctor (clw_vector.hpp:14-33)
 * clwh_image_create  replaces clCreateImage in clw_image's ctor  (clw_image.hpp:18-137);
 *                    dims of 0 count as 1; 1-D/2-D/3-D follows from the dims
 * clwh_mem_wrap      adopts device memory the caller allocated (a torch tensor's data_ptr) or that another
 *                    clwh_mem -- of this or another context -- already names; never freed by the shim.  All objects
 *                    that name the same device pointer share ONE content version: a push, clwh_sdf_build or
 *                    clwh_mem_mark_dirty through any of them invalidates what every context derived from it
 * clwh_mem_push/pull replace the blocking clEnqueueWrite/Read{Buffer,Image}
 *                    (clw_vector.hpp:73-86, clw_image.hpp:202-216); `bytes` must equal the object size
 */
int clwh_mem_create(clwh_ctx *ctx, size_t bytes, int flags, clwh_mem **out);
int clwh_mem_wrap(clwh_ctx *ctx, void *device_ptr, size_t bytes, clwh_mem **out);
int clwh_image_create(clwh_ctx *ctx, const size_t dims[3], int channels, int elem_kind, int flags,
                      clwh_mem **out);
int clwh_image_wrap(clwh_ctx *ctx, void *device_ptr, const size_t dims[3], int channels,
                    int elem_kind, clwh_mem **out);
int clwh_mem_push(clwh_ctx *ctx, clwh_mem *mem, const void *host, size_t bytes);
int clwh_mem_pull(clwh_ctx *ctx, clwh_mem *mem, void *host, size_t bytes);
int clwh_mem_release(clwh_mem *mem);
/* page-lock / unlock a host buffer the caller keeps pushing from or pulling into (the host mirror of a clw_image,
 * e.g. the 8 MiB frame renderer::render_frame pulls after every pass, app/renderer.cpp:150): transfers then run as
 * one DMA instead of being staged through the runtime's bounce buffers.  Optional; never changes results. */
int clwh_host_register(void *host, size_t bytes);
int clwh_host_unregister(void *host);
void *clwh_mem_device_ptr(clwh_mem *mem);
size_t clwh_mem_size(clwh_mem *mem);
/* tell the shim that device code outside it rewrote the object (invalidates derived layouts) */
int clwh_mem_mark_dirty(clwh_mem *mem);

/* ---- kernels.
 * clwh_kernel_get replaces clw_function's ctor (clw_function.hpp:74-111): instead of flattening
 * `#clw_include_once`, prepending `prepend` and JIT-compiling, it looks (file, entry) up in the
 * registry of precompiled HIP kernels and parses `prepend` (the generated `is_event_gen` source,
 * app/ui.cpp:160-168) into a rule table that becomes a launch-time parameter; source outside the rule
 * grammar is compiled with hiprtc into a per-voxel classifier (csrc/tf_jit.cpp), so any `is_event_gen`
 * the reference's JIT would accept is accepted here too (up to 16 distinct colours).
 * Known pairs: ("ray_marching.cl","render"), ("signed_distance_field.cl","create_base_image"),
 * ("signed_distance_field.cl","create_signed_distance_field"), ("buffer_reset.cl","buffer_reset"),
 * ("empty.cl","empty"), and next to the hot path ("reference_volume_figures.cl","fetch_stats"),
 * ("reference_volume_clip.cl","apply_clip"), ("volume_filter.cl","bilateral_filter"),
 * ("histogram.cl","tf_sort_values"), ("histogram.cl","tf_flush_color_frame").
 */
int clwh_kernel_get(clwh_ctx *ctx, const char *file, const char *entry, const char *prepend,
                    clwh_kernel **out);
int clwh_kernel_release(clwh_kernel *k);

enum clwh_arg_kind { CLWH_ARG_MEM = 0, CLWH_ARG_I32 = 1, CLWH_ARG_U32 = 2, CLWH_ARG_F32 = 3,
                     CLWH_ARG_I64 = 4, CLWH_ARG_U64 = 5, CLWH_ARG_F64 = 6 };
typedef struct clwh_arg {
  int32_t kind;
  int32_t reserved;
  union {
    clwh_mem *mem;
    int32_t i32;
    uint32_t u32;
    float f32;
    int64_t i64;
    uint64_t u64;
    double f64;
  } v;
} clwh_arg;

/* clwh_launch replaces clw_function::execute (clw_function.hpp:217-244): zeros in global/local
 * count as 1; global must be a multiple of local (the reference asserts it); the argument list is
 * the reference kernel's own, in order (clSetKernelArg by position, clw_function.hpp:152-167). */
int clwh_launch(clwh_kernel *k, const size_t global[3], const size_t local[3], const clwh_arg *args,
                int nargs);

/* ---- hot-path entry points beyond the generic launch (used by the C++ host mirror and bench).
 *
 * clwh_render: one `render` pass = one sample per pixel (app/renderer.cpp:145-148 +
 * opencl_kernels/ray_marching.cl:152-199), with the options the MI355X design adds.
 */
enum clwh_accum_mode {
  CLWH_ACCUM_VOXEL_CACHE = 0, /* reference-exact world-space cache (utility.cl:20-54) */
  CLWH_ACCUM_IMAGE_SPACE = 1  /* per-pixel float4 {r,g,b,count}, no token cap (multi-GPU tiles) */
};
/* Which of the reference's two shading functions `render` runs.  ray_marching.cl:186 calls compute_light; compute_ao
 * (:104-149) is the alternate the authors kept in the file.  AO needs accum_mode CLWH_ACCUM_VOXEL_CACHE: it keeps
 * {samples, occluded} as 2 ushorts per voxel in buffer_volume (utility.cl:123-159; at least clwh_cache_len()/2
 * ushorts), capped at 100 samples per voxel.  compute_ao's return value has w == 0, which the reference's render()
 * would overpaint with the environment colour (:187-195); here an AO hit pixel resolves to {v, v, v, 1},
 * v = 2 * (100 - occluded), the value compute_ao returns. */
enum clwh_shading { CLWH_SHADE_LIGHT = 0, CLWH_SHADE_AO = 1 };
typedef struct clwh_render_desc {
  clwh_mem *frame;          /* RGBA8 2-D image; its dims are what get_image_width/height(frame) return */
  clwh_mem *volume;         /* S16 3-D image */
  clwh_mem *sdf;            /* S8 3-D image */
  clwh_mem *env;            /* RGBA8 2-D image */
  clwh_mem *buffer_volume;  /* voxel cache, clwh_cache_len() ushorts (mode 0) */
  float cam_pos[3];
  float cam_dir[3];
  int32_t seed;             /* the pass's random seed (renderer.cpp:142) when n_seeds == 0 */
  uint32_t width, height;   /* NDRange global size (multiples of 8) */
  int32_t accum_mode;
  clwh_mem *accum;          /* mode 1: float4 per pixel, tile-major (see clwh_accum_len) */
  int32_t tile_rank, tile_world; /* image-tile partition: 8x8 tiles, owner = (tx + ty) % world */
  int32_t write_frame;      /* 1: resolve the frame (deterministic, after all adds of the pass) */
  clwh_mem *hit_index;      /* optional int64 per pixel, row-major: cache entry or -1 (parity tests) */
  clwh_mem *contrib;        /* optional uint32[4] per pixel, row-major: r,g,b,granted (parity tests; one seed) */
  int32_t n_seeds;          /* > 0: render n_seeds passes (<= CLWH_MAX_SEEDS) in ONE launch, seeds[] below; the
                               result equals n_seeds consecutive single-seed calls (image-space mode always; voxel
                               cache mode while no voxel reaches the 256-token cap).  ceil(hit pixels / 64) x
                               n_seeds must stay below 2^24 (64 seeds: 16.7 M hit pixels), else
                               CLWH_ERR_INVALID_VALUE: use fewer seeds per launch */
  int32_t seeds[64];
  int32_t shading;          /* enum clwh_shading; 0 (zero-initialised descriptors) = compute_light */
  int32_t resolve_only;     /* 1: no pass; only resolve `frame` from buffer_volume / accum for this camera (the read-back
                               half of compute_light, ray_marching.cl:82-99) -- e.g. after the cache was updated by an
                               exchange between ranks */
} clwh_render_desc;
#define CLWH_MAX_SEEDS 64
int clwh_render(clwh_kernel *render_kernel, const clwh_render_desc *desc);

/* number of ushorts of a voxel cache for an X*Y*Z volume (the reference allocates X*Y*Z*4,
 * app/renderer.cpp:29-30, and can index one row past it, utility.cl:21 with utility_ray.cl:112-117;
 * the shim's kernels require the padded length) */
int64_t clwh_cache_len(uint32_t X, uint32_t Y, uint32_t Z);
/* number of float4 of a rank's tile-major accumulation buffer */
int64_t clwh_accum_len(uint32_t width, uint32_t height, int32_t tile_world);
/* resolve the RGBA8 frame from the tile-major accumulation buffers of all ranks laid back to back
 * (what the RCCL all-gather produces): hit pixels (count > 0) get ray_marching.cl:82-99 applied to
 * their sums; the others get the environment colour of their camera ray, alpha 200
 * (ray_marching.cl:172-178), which is why the camera and the env map are arguments. */
int clwh_accum_resolve(clwh_ctx *ctx, clwh_mem *accum_all_ranks, int32_t tile_world, uint32_t width,
                       uint32_t height, clwh_mem *frame_rgba8, clwh_mem *env, const float cam_pos[3],
                       const float cam_dir[3]);
/* The same resolve for several GPUs with a quarter of the exchange: a rank first resolves ITS tiles to RGBA8
 * (`tiles_rgba8`: clwh_accum_len(width, height, tile_world) 32-bit pixels, tile-major in the slot order of its accumulation
 * buffer), the ranks all-gather those 4-byte pixels instead of 16-byte sums, and clwh_frame_from_tiles puts all ranks' tiles,
 * laid back to back, into the row-major frame.  Pixel for pixel the result of clwh_accum_resolve (the resolve of
 * ray_marching.cl:82-99 / :172-178 is per pixel). */
int clwh_accum_resolve_tiles(clwh_ctx *ctx, clwh_mem *accum, int32_t tile_rank, int32_t tile_world, uint32_t width,
                             uint32_t height, clwh_mem *tiles_rgba8, clwh_mem *env, const float cam_pos[3],
                             const float cam_dir[3]);
int clwh_frame_from_tiles(clwh_ctx *ctx, clwh_mem *tiles_all_ranks, int32_t tile_world, uint32_t width, uint32_t height,
                          clwh_mem *frame_rgba8);
/* drop what the context derived from its inputs; the next clwh_render rebuilds it.  Needed only to time the
 * rebuild or to render "the first frame after a camera move" again (memory rewritten behind the shim's back
 * is what clwh_mem_mark_dirty is for).
 *   CLWH_DERIVED_SCENE   step bytes + hit records + exit-certificate table (function of volume, SDF, transfer
 *                        function: flush-time data).  There is ONE copy per device: contexts that render the same
 *                        (volume content, SDF content, transfer function) share it -- 9 bytes per voxel, 72 GiB at
 *                        2048^3 -- and it is freed when the last of them lets go.  Invalidating makes THIS context
 *                        rebuild; contexts already sharing the old copy keep it until their own inputs change.
 *   CLWH_DERIVED_CAMERA  primary hits (function of the camera and of the scene), per context */
enum clwh_derived { CLWH_DERIVED_SCENE = 1, CLWH_DERIVED_CAMERA = 2 };
int clwh_ctx_invalidate_derived(clwh_ctx *ctx, int what);
/* which copy of the derived scene data the context renders from (after its last clwh_render): a process-wide unique id
 * of the content (0: none yet), its size in bytes, and how many contexts hold it right now */
int clwh_ctx_scene_info(clwh_ctx *ctx, uint64_t *scene_id, uint64_t *bytes, int32_t *holders);

/* clwh_sdf_build replaces the host loop of signed_distance_field::signed_distance_field
 * (app/signed_distance_field.cpp:7-35): base image + all propagation layers, no host round trip
 * per layer.  `sdf` is an S8 3-D image of the volume's dims.  n_launches (optional) receives the
 * number of create_signed_distance_field layers the reference's loop would have run. */
int clwh_sdf_build(clwh_ctx *ctx, clwh_mem *volume, const char *tf_source, clwh_mem *sdf,
                   int32_t *n_launches);

/* buffer_reset.cl:3-13 / app/renderer.cpp:32-35 */
int clwh_buffer_reset(clwh_ctx *ctx, clwh_mem *buffer_volume);

/* ---- the reference's world-space accumulation fed with contributions computed elsewhere (the pixels of other
 * ranks): utility.cl:20-54 (token + add) with the 256-token rule of ray_marching.cl:28,39 applied to the GLOBAL
 * count ("reference-exact voxel-cache mode").  The reference hands a voxel's tokens to whichever
 * work-items reach the atomic first; across GPUs the exchange fixes one legal outcome of that race instead: the
 * contributions to a voxel are taken in the order the caller lists them -- (rank, pixel) -- while the voxel's count
 * is below 256 and dropped afterwards, so every rank that applies the same list to its replica of buffer_volume ends
 * with the same bytes, equal to the single-GPU cache while the count stays below the cap.
 *   clwh_cache_exchange_plan         once per camera.  entries: int64[n] on the device, the cache entry (what
 *                                    clwh_render_desc.hit_index reports) of every contributing pixel of every rank
 *                                    in (rank, pixel) order; entries outside the cache are ignored.  Blocking
 *                                    (one stable sort).
 *   clwh_cache_apply_contributions   once per pass.  rgb: int32[n][rgb_stride] on the device, the pass's contribution
 *                                    of each listed pixel (clwh_render_desc.contrib rows, gathered), same order.
 *                                    One kernel on the context's stream, no host round trip. */
typedef struct clwh_exchange_plan clwh_exchange_plan;
int clwh_cache_exchange_plan(clwh_ctx *ctx, clwh_mem *entries, uint64_t n, clwh_exchange_plan **out);
int clwh_cache_apply_contributions(clwh_ctx *ctx, clwh_exchange_plan *plan, clwh_mem *buffer_volume, clwh_mem *rgb,
                                   int32_t rgb_stride);
int clwh_cache_exchange_plan_release(clwh_exchange_plan *plan);

/* ---- display hand-off without a host readback.
 * Replaces clw_foreign_memory::acquire / release (opencl_wrapper/include/clw_foreign_memory.hpp:40-46:
 * clEnqueueAcquireGLObjects / clEnqueueReleaseGLObjects on a cl_mem made from a GL texture).  The frame the
 * render kernels write is any device memory adopted with clwh_image_wrap -- the pointer a graphics-interop
 * mapping (hipGraphicsResourceGetMappedPointer on a registered GL buffer) or another framework's allocator hands
 * out -- so the frame never crosses PCIe.  Ordering between the context's stream and the consumer's stream is
 * by HIP events, no host synchronisation:
 *   clwh_ctx_acquire_from(ctx, s)  work queued on the context's stream after this call waits for everything
 *                                  queued on stream `s` so far (the display has finished with the previous frame)
 *   clwh_ctx_release_to(ctx, s)    work queued on stream `s` after this call waits for everything queued on the
 *                                  context's stream so far (the frame is complete before the display reads it)
 * `s` is a hipStream_t (NULL = the default stream). */
int clwh_ctx_acquire_from(clwh_ctx *ctx, void *hip_stream);
int clwh_ctx_release_to(clwh_ctx *ctx, void *hip_stream);

/* ---- transfer function: the parsed form of the generated `is_event_gen` source */
#define CLWH_TF_MAX_RULES 16
typedef struct clwh_tf_rule {
  int32_t v_lo, v_hi;   /* value    in [v_lo, v_hi] */
  int32_t g_lo, g_hi;   /* gradient in [g_lo, g_hi] when use_gradient */
  int32_t use_gradient;
  int32_t writes_color;
  int32_t terminal;
  int32_t color[4];
} clwh_tf_rule;
typedef struct clwh_tf {
  int32_t n;
  clwh_tf_rule rules[CLWH_TF_MAX_RULES];
} clwh_tf;
int clwh_tf_parse(const char *source, clwh_tf *out);

/* ---- diagnostics */
/* self-test of the library's float -> integer conversions (DESIGN.md "Semantics": truncate, saturate, NaN -> 0), which the kernels
 * perform with single hardware instructions: converts n floats on the device both ways -- the instruction and the definition
 * written out -- into int32 / uint32 arrays of 2 n elements each ([0, n): instruction, [n, 2n): definition) */
int clwh_debug_float_conversions(clwh_ctx *ctx, clwh_mem *floats_in, uint64_t n, clwh_mem *i32_out, clwh_mem *u32_out);
/* self-test of the wave-wide minimum k_repack takes over a sub-brick's 64 voxels for the exit-certificate table (cross-lane DPP
 * operations): n (a multiple of 64) values, one wave per 64; u32_out[0, n/64) = that minimum, [n/64, 2n/64) = the same by a shuffle loop */
int clwh_debug_wave_min(clwh_ctx *ctx, clwh_mem *u32_in, uint64_t n, clwh_mem *u32_out);
const char *clwh_strerror(int status);
int clwh_last_hip_error(void);
const char *clwh_version(void);
/* per-context timing of the dominant kernel of clwh_render: when enabled, every clwh_render records
 * a HIP event pair on the context's stream around that kernel (no host sync per pass);
 * clwh_ctx_timing_read waits for them, returns the summed duration and launch count, and resets. */
int clwh_ctx_set_timing(clwh_ctx *ctx, int enabled);
int clwh_ctx_timing_read(clwh_ctx *ctx, float *total_ms, int32_t *launches);
/* the same per kernel of the render path: ms[k] / launches[k] for k < n, k = enum clwh_timer (HIP events on the
 * context's stream around each launch; what bench.py's per-kernel roofline is computed from) */
enum clwh_timer {
  CLWH_TIMER_BOUNCE = 0,   /* k_bounce (the dominant kernel; what clwh_ctx_timing_read returns) */
  CLWH_TIMER_PRIMARY = 1,  /* k_primary */
  CLWH_TIMER_FIXUP = 2,    /* k_env_fixup + k_commit */
  CLWH_TIMER_RESOLVE = 3,  /* k_resolve / k_accum_resolve */
  CLWH_TIMER_REPACK = 4,   /* k_repack */
  CLWH_TIMER_AO = 5,       /* k_ao */
  CLWH_TIMER_COUNT = 6
};
int clwh_ctx_timing_read_all(clwh_ctx *ctx, float *ms, int32_t *launches, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* CLWH_H */
