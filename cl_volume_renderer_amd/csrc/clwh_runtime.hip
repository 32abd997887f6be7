// clwh_runtime.hip -- host side of libclwhip.so: the C ABI declared in include/clwh.h.
//
// Replaces the reference's opencl_wrapper (clw_context / clw_vector / clw_image / clw_function)
// with a thin layer over the HIP runtime: one in-order stream per context, hipMalloc'ed objects,
// a registry of precompiled gfx950 kernels keyed by the reference's (file, entry) names, and the
// transfer-function source parsed into a launch-time table instead of being JIT-compiled.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "clwh_internal.hpp"

using namespace clvr;

static thread_local int g_last_hip_error = 0;

// Content versions are drawn from one process-wide counter, so (device pointer, version) identifies a
// content uniquely even when an object is released and another one is allocated at the same address.
// The version lives in a cell shared by every clwh_mem of the same device pointer (the owner and its wraps in
// other contexts): whoever rewrites the memory -- a push, clwh_sdf_build, clwh_mem_mark_dirty -- is seen by all of
// them, so no context keeps rendering from derived data of the old content.
static std::atomic<uint64_t> g_content_version{0};
static std::mutex g_registry_mutex;  // guards the two process-wide registries below (contexts may live on different host threads)
static std::map<const void *, std::weak_ptr<VersionCell>> g_version_cells;
void clwh_touch(clwh_mem *m) { m->cell->v.store(++g_content_version, std::memory_order_relaxed); }
static inline void touch(clwh_mem *m) { clwh_touch(m); }

static std::shared_ptr<VersionCell> version_cell_of(const void *dptr) {
  std::lock_guard<std::mutex> lock(g_registry_mutex);
  auto it = g_version_cells.find(dptr);
  if (it != g_version_cells.end())
    if (auto live = it->second.lock()) return live;
  auto cell = std::make_shared<VersionCell>();
  g_version_cells[dptr] = cell;
  if (g_version_cells.size() > 4096) {  // forget the cells nobody holds any more
    for (auto i = g_version_cells.begin(); i != g_version_cells.end();)
      i = i->second.expired() ? g_version_cells.erase(i) : std::next(i);
  }
  return cell;
}

// ---- the per-device registry of derived scene data (PackedScene, clwh_internal.hpp)
static std::vector<std::weak_ptr<PackedScene>> g_packed_scenes;
static std::atomic<uint64_t> g_packed_generation{0};

PackedScene::~PackedScene() {
  // the last context let go; another context that dropped its reference earlier may still have kernels in flight on its
  // own stream that read this memory
  (void)hipSetDevice(device);
  (void)hipDeviceSynchronize();
  if (data) (void)hipFree(data);
  if (ready) (void)hipEventDestroy(ready);
}

#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) {                             \
      g_last_hip_error = (int)_e;                       \
      return _e == hipErrorOutOfMemory ? CLWH_ERR_OUT_OF_MEMORY : CLWH_ERR_HIP; \
    }                                                   \
  } while (0)

static size_t elem_size(int kind) {
  switch (kind) {
    case CLWH_ELEM_S8: case CLWH_ELEM_U8: return 1;
    case CLWH_ELEM_S16: case CLWH_ELEM_U16: return 2;
    case CLWH_ELEM_S32: case CLWH_ELEM_U32: case CLWH_ELEM_F32: return 4;
    default: return 0;
  }
}

static void tf_to_dev(const clwh_tf &tf, TfDev &d) {
  std::memset(&d, 0, sizeof d);
  d.n = tf.n;
  for (int k = 0; k < tf.n; ++k) {
    const clwh_tf_rule &r = tf.rules[k];
    TfRuleDev &o = d.rules[k];
    o.v_lo = r.v_lo; o.v_hi = r.v_hi; o.g_lo = r.g_lo; o.g_hi = r.g_hi;
    o.flags = (r.use_gradient ? TF_USE_GRADIENT : 0u) | (r.writes_color ? TF_WRITES_COLOR : 0u) |
              (r.terminal ? TF_TERMINAL : 0u);
    o.color = ((uint32_t)r.color[0] & 255u) | (((uint32_t)r.color[1] & 255u) << 8) |
              (((uint32_t)r.color[2] & 255u) << 16) | (((uint32_t)r.color[3] & 255u) << 24);
    if (r.use_gradient) d.uses_gradient = 1;
  }
  // the border texel's class (see TfDev::border_class); tables that read `gradient` classify such positions literally
  for (int k = 0; k < tf.n && !d.uses_gradient; ++k) {
    const clwh_tf_rule &r = tf.rules[k];
    if (r.v_lo <= 0 && 0 <= r.v_hi) { d.border_class = k + 1; break; }
    if (r.terminal) break;
  }
}

// ---- hiprtc fallback (tf_jit.cpp): compile once per source text and context
static int jit_for_source(clwh_ctx *ctx, const char *source, std::shared_ptr<JitTf> &out) {
  auto it = ctx->jit_cache.find(source);
  if (it != ctx->jit_cache.end()) {
    out = it->second;
    return CLWH_OK;
  }
  auto j = std::make_shared<JitTf>();
  j->source = source;
  std::string log;
  const int rc = tf_jit_compile(source, j->code, log);
  if (rc != CLWH_OK) {
    if (!log.empty()) std::fprintf(stderr, "clwhip: transfer-function source is neither in the rule grammar nor compilable:\n%s\n", log.c_str());
    return rc;
  }
  ctx->jit_cache[source] = j;
  out = j;
  return CLWH_OK;
}

// class byte per voxel + colour palette of an opaque TF for `volume`; cached per (volume content, source)
static int ensure_classes(clwh_ctx *ctx, const std::shared_ptr<JitTf> &jit, const clwh_mem *volume, TfDev &tf_out,
                          const uint8_t **cls_out) {
  if (ctx->jit_cls && ctx->jit_vol == volume->dptr && ctx->jit_vol_ver == volume->version() && ctx->jit_source == jit->source) {
    tf_out = ctx->jit_tf;
    *cls_out = ctx->jit_cls;
    return CLWH_OK;
  }
  if (!jit->module) {
    HIP_TRY(hipModuleLoadData(&jit->module, jit->code.data()));
    HIP_TRY(hipModuleGetFunction(&jit->classify, jit->module, "clvr_tf_classify"));
  }
  const size_t voxels = volume->dims[0] * volume->dims[1] * volume->dims[2];
  if (ctx->jit_cls_bytes < voxels) {
    if (ctx->jit_cls) {
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      HIP_TRY(hipFree(ctx->jit_cls));
      ctx->jit_cls = nullptr;
      ctx->jit_cls_bytes = 0;
    }
    HIP_TRY(hipMalloc((void **)&ctx->jit_cls, voxels));
    ctx->jit_cls_bytes = voxels;
  }
  constexpr int kColors = CLWH_TF_MAX_RULES;
  if (!ctx->jit_palette) HIP_TRY(hipMalloc((void **)&ctx->jit_palette, (kColors + 2) * sizeof(unsigned long long)));  // colours, error word, border class
  HIP_TRY(hipMemsetAsync(ctx->jit_palette, 0xFF, kColors * sizeof(unsigned long long), ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->jit_palette + kColors, 0, 2 * sizeof(unsigned long long), ctx->stream));
  const void *vol = volume->dptr;
  int X = (int)volume->dims[0], Y = (int)volume->dims[1], Z = (int)volume->dims[2], max_colors = kColors;
  unsigned char *cls = ctx->jit_cls;
  unsigned long long *palette = ctx->jit_palette;
  int *error = reinterpret_cast<int *>(ctx->jit_palette + kColors);
  int *border = reinterpret_cast<int *>(ctx->jit_palette + kColors + 1);
  void *args[] = {&vol, &X, &Y, &Z, &cls, &palette, &max_colors, &error, &border};
  const size_t blocks = std::min<size_t>((voxels + 255u) / 256u, (size_t)1u << 23);  // the classifier strides over the rest
  HIP_TRY(hipModuleLaunchKernel(jit->classify, (unsigned)blocks, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr));
  unsigned long long host[kColors + 2];
  HIP_TRY(hipMemcpyAsync(host, ctx->jit_palette, sizeof host, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx->jit_vol = nullptr;  // invalid until fully built
  if ((int)(host[kColors] & 0xFFFFFFFFull) != 0) return CLWH_ERR_TF_UNSUPPORTED;  // more distinct colours than the table holds
  TfDev t;
  std::memset(&t, 0, sizeof t);
  t.opaque = 1;
  for (int k = 0; k < kColors && host[k] != ~0ull; ++k) {
    t.rules[k].flags = (host[k] >> 32) & 1ull ? TF_WRITES_COLOR : 0u;
    t.rules[k].color = (uint32_t)(host[k] & 0xFFFFFFFFull);
    t.n = k + 1;
  }
  t.border_class = (int32_t)(host[kColors + 1] & 0xFFull);  // is_event_gen(0, 0): the border texel (TfDev::border_class)
  ctx->jit_tf = t;
  ctx->jit_vol = volume->dptr;
  ctx->jit_vol_ver = volume->version();
  ctx->jit_source = jit->source;
  tf_out = t;
  *cls_out = ctx->jit_cls;
  return CLWH_OK;
}

extern "C" {

// ------------------------------------------------------------------------------------------------
// context

int clwh_ctx_create_on_stream(int device, void *hip_stream, clwh_ctx **out) {
  if (!out) return CLWH_ERR_INVALID_VALUE;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_last_hip_error = (int)e;
    return CLWH_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count) return CLWH_ERR_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));
  clwh_ctx *c = new (std::nothrow) clwh_ctx();
  if (!c) return CLWH_ERR_OUT_OF_MEMORY;
  c->device = device;
  c->stream = (hipStream_t)hip_stream;
  c->own_stream = false;
  if (const char *e = std::getenv("CLWH_TUNE_STEP")) c->tune_step_min_lanes = std::max(0, std::min(64, std::atoi(e)));
  if (const char *e = std::getenv("CLWH_TUNE_REFILL")) c->tune_refill_min_lanes = std::max(0, std::min(64, std::atoi(e)));
  if (const char *e = std::getenv("CLWH_TUNE_LITERAL_GRADIENT")) c->tune_literal_gradient = std::atoi(e) != 0;
  if (const char *e = std::getenv("CLWH_TUNE_AFFINITY")) c->tune_unit_affinity = std::atoi(e);
  if (const char *e = std::getenv("CLWH_TUNE_QUEUES")) c->tune_unit_queues = std::max(1, std::min(8, std::atoi(e)));
  if (const char *e = std::getenv("CLWH_TUNE_GROUP")) c->tune_unit_group = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("CLWH_TUNE_CHUNK_BLOCK_LOG2")) c->tune_unit_block_log2 = std::max(0, std::min(8, std::atoi(e)));
  if (const char *e = std::getenv("CLWH_TUNE_BLOCKS")) c->tune_bounce_max_blocks = (uint32_t)std::max(1, std::atoi(e));
  if (const char *e = std::getenv("CLWH_TUNE_MACRO_SHIFT")) c->tune_macro_shift = std::atoi(e);
  if (const char *e = std::getenv("CLWH_TUNE_LONG_LAUNCH")) c->tune_force_long_launch = std::atoi(e) != 0;
  if (const char *e = std::getenv("CLWH_TUNE_BOUNCE_RAYS")) c->tune_bounce_rays = std::atoi(e) == 2 ? 2 : 1;
  if (const char *e = std::getenv("CLWH_TUNE_SDF")) c->tune_sdf_front = std::strcmp(e, "front") == 0;
  if (const char *e = std::getenv("CLWH_TUNE_SDFBIT_WAVES")) c->tune_sdfbit_waves = std::atoi(e) == 16 ? 16 : 8;
  if (const char *e = std::getenv("CLWH_TUNE_SDFBIT_GRID")) c->tune_sdfbit_grid = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("CLWH_TUNE_SDFBIT_REC")) c->tune_sdfbit_rec_lds = std::strcmp(e, "lds") == 0 ? 1 : 0;
  if (const char *e = std::getenv("CLWH_TUNE_CERT")) c->tune_cert_min_step = std::max(0, std::min(127, std::atoi(e)));
  *out = c;
  return CLWH_OK;
}

int clwh_ctx_create(int device, clwh_ctx **out) {
  int rc = clwh_ctx_create_on_stream(device, nullptr, out);
  if (rc != CLWH_OK) return rc;
  hipStream_t s;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    delete *out;
    *out = nullptr;
    return CLWH_ERR_HIP;
  }
  (*out)->stream = s;
  (*out)->own_stream = true;
  return CLWH_OK;
}

int clwh_ctx_destroy(clwh_ctx *ctx) {
  if (!ctx) return CLWH_ERR_INVALID_VALUE;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->pix_slot) (void)hipFree(ctx->pix_slot);
  if (ctx->hits) (void)hipFree(ctx->hits);
  if (ctx->render_counters) (void)hipFree(ctx->render_counters);
  if (ctx->sticky_flags) (void)hipFree(ctx->sticky_flags);
  if (ctx->fixups) (void)hipFree(ctx->fixups);
  if (ctx->delta) (void)hipFree(ctx->delta);
  if (ctx->vox_plan) (void)hipFree(ctx->vox_plan);
  if (ctx->vox_temp) (void)hipFree(ctx->vox_temp);
  if (ctx->jit_cls) (void)hipFree(ctx->jit_cls);
  if (ctx->jit_palette) (void)hipFree(ctx->jit_palette);
  for (auto &kv : ctx->jit_cache)
    if (kv.second->module) (void)hipModuleUnload(kv.second->module);
  if (ctx->bilateral_weights) (void)hipFree(ctx->bilateral_weights);
  if (ctx->sdf_counters) (void)hipFree(ctx->sdf_counters);
  if (ctx->sdf_flags) (void)hipFree(ctx->sdf_flags);
  if (ctx->sdf_bits) (void)hipFree(ctx->sdf_bits);
  ctx->scene.reset();
  if (ctx->host_n_hits) (void)hipHostFree(ctx->host_n_hits);
  if (ctx->n_hits_event) (void)hipEventDestroy(ctx->n_hits_event);
  if (ctx->handoff_event) (void)hipEventDestroy(ctx->handoff_event);
  for (hipEvent_t e : ctx->ev_begin) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_end) (void)hipEventDestroy(e);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return CLWH_OK;
}

// a render may have overflowed its fix-up buffer (never with sane env-map sizes); the flag is read at
// the next synchronisation point and reported instead of handing out incomplete results
static int check_device_flags(clwh_ctx *ctx) {
  if (!ctx->fixup_overflow_pending || !ctx->sticky_flags) return CLWH_OK;
  uint32_t flag = 0;
  HIP_TRY(hipMemcpyAsync(&flag, ctx->sticky_flags, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx->fixup_overflow_pending = false;
  if (!flag) return CLWH_OK;
  // the flag is sticky on the device (no render resets it): cleared here, once it has been reported
  HIP_TRY(hipMemsetAsync(ctx->sticky_flags, 0, sizeof(uint32_t), ctx->stream));
  return CLWH_ERR_INTERNAL_OVERFLOW;
}

int clwh_ctx_finish(clwh_ctx *ctx) {
  if (!ctx) return CLWH_ERR_INVALID_VALUE;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return check_device_flags(ctx);
}

static int order_streams(clwh_ctx *ctx, hipStream_t first, hipStream_t then) {
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ctx->handoff_event) HIP_TRY(hipEventCreateWithFlags(&ctx->handoff_event, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ctx->handoff_event, first));
  HIP_TRY(hipStreamWaitEvent(then, ctx->handoff_event, 0));
  return CLWH_OK;
}

int clwh_ctx_acquire_from(clwh_ctx *ctx, void *hip_stream) {
  if (!ctx) return CLWH_ERR_INVALID_VALUE;
  return order_streams(ctx, (hipStream_t)hip_stream, ctx->stream);
}

int clwh_ctx_release_to(clwh_ctx *ctx, void *hip_stream) {
  if (!ctx) return CLWH_ERR_INVALID_VALUE;
  return order_streams(ctx, ctx->stream, (hipStream_t)hip_stream);
}

void *clwh_ctx_stream(clwh_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int clwh_ctx_device(clwh_ctx *ctx) { return ctx ? ctx->device : -1; }

int clwh_ctx_set_timing(clwh_ctx *ctx, int enabled) {
  if (!ctx) return CLWH_ERR_INVALID_VALUE;
  ctx->timing = enabled != 0;
  ctx->ev_used = 0;
  return CLWH_OK;
}

int clwh_ctx_timing_read_all(clwh_ctx *ctx, float *ms, int32_t *launches, int32_t n) {
  if (!ctx || !ms || !launches || n <= 0) return CLWH_ERR_INVALID_VALUE;
  for (int k = 0; k < n; ++k) { ms[k] = 0.0f; launches[k] = 0; }
  HIP_TRY(hipSetDevice(ctx->device));
  for (size_t i = 0; i < ctx->ev_used; ++i) {
    HIP_TRY(hipEventSynchronize(ctx->ev_end[i]));
    float t = 0.0f;
    HIP_TRY(hipEventElapsedTime(&t, ctx->ev_begin[i], ctx->ev_end[i]));
    const int k = ctx->ev_which[i];
    if (k >= 0 && k < n) { ms[k] += t; launches[k] += 1; }
  }
  ctx->ev_used = 0;
  return CLWH_OK;
}

int clwh_ctx_timing_read(clwh_ctx *ctx, float *total_ms, int32_t *launches) {
  if (!ctx || !total_ms || !launches) return CLWH_ERR_INVALID_VALUE;
  float ms[CLWH_TIMER_COUNT];
  int32_t n[CLWH_TIMER_COUNT];
  const int rc = clwh_ctx_timing_read_all(ctx, ms, n, CLWH_TIMER_COUNT);
  *total_ms = ms[CLWH_TIMER_BOUNCE];
  *launches = n[CLWH_TIMER_BOUNCE];
  return rc;
}

// a (begin, end) event pair for one launch of kernel `which`; null events when timing is off
struct TimedLaunch {
  clwh_ctx *ctx;
  hipEvent_t b = nullptr, e = nullptr;
  int begin(clwh_ctx *c, int which) {
    ctx = c;
    if (!c->timing) return CLWH_OK;
    if (c->ev_used == c->ev_begin.size()) {
      hipEvent_t x, y;
      HIP_TRY(hipEventCreate(&x));
      HIP_TRY(hipEventCreate(&y));
      c->ev_begin.push_back(x);
      c->ev_end.push_back(y);
      c->ev_which.push_back(0);
    }
    b = c->ev_begin[c->ev_used];
    e = c->ev_end[c->ev_used];
    c->ev_which[c->ev_used] = which;
    c->ev_used++;
    HIP_TRY(hipEventRecord(b, c->stream));
    return CLWH_OK;
  }
  int end() {
    if (e) HIP_TRY(hipEventRecord(e, ctx->stream));
    return CLWH_OK;
  }
};

// ------------------------------------------------------------------------------------------------
// memory objects

static int mem_new(clwh_ctx *ctx, void *dptr, size_t bytes, bool owned, clwh_mem **out) {
  clwh_mem *m = new (std::nothrow) clwh_mem();
  if (!m) return CLWH_ERR_OUT_OF_MEMORY;
  m->ctx = ctx;
  m->dptr = dptr;
  m->bytes = bytes;
  m->owned = owned;
  m->cell = version_cell_of(dptr);
  // a new version also for a wrap of memory another clwh_mem already names: the caller may have rewritten it since, and
  // the address may have been reused by its allocator -- the owner's derived data is rebuilt once per wrap
  touch(m);
  *out = m;
  return CLWH_OK;
}

int clwh_mem_create(clwh_ctx *ctx, size_t bytes, int flags, clwh_mem **out) {
  if (!ctx || !out || bytes == 0) return CLWH_ERR_INVALID_VALUE;
  *out = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  void *p = nullptr;
  HIP_TRY(hipMalloc(&p, bytes));
  int rc = mem_new(ctx, p, bytes, true, out);
  if (rc != CLWH_OK) {
    (void)hipFree(p);
    return rc;
  }
  (*out)->flags = flags;
  return CLWH_OK;
}

int clwh_mem_wrap(clwh_ctx *ctx, void *device_ptr, size_t bytes, clwh_mem **out) {
  if (!ctx || !out || !device_ptr || bytes == 0) return CLWH_ERR_INVALID_VALUE;
  *out = nullptr;
  return mem_new(ctx, device_ptr, bytes, false, out);
}

static int image_describe(clwh_mem *m, const size_t dims_in[3], int channels, int elem_kind) {
  const size_t es = elem_size(elem_kind);
  if (es == 0 || !(channels == 1 || channels == 2 || channels == 4)) return CLWH_ERR_INVALID_VALUE;
  m->is_image = true;
  for (int k = 0; k < 3; ++k) m->dims[k] = dims_in[k] == 0 ? 1 : dims_in[k];
  m->channels = channels;
  m->elem_kind = elem_kind;
  return CLWH_OK;
}

int clwh_image_create(clwh_ctx *ctx, const size_t dims[3], int channels, int elem_kind, int flags, clwh_mem **out) {
  if (!ctx || !out || !dims) return CLWH_ERR_INVALID_VALUE;
  *out = nullptr;
  const size_t es = elem_size(elem_kind);
  if (es == 0 || !(channels == 1 || channels == 2 || channels == 4)) return CLWH_ERR_INVALID_VALUE;
  size_t d[3];
  for (int k = 0; k < 3; ++k) d[k] = dims[k] == 0 ? 1 : dims[k];
  // clw_image.hpp:45-58: width must exceed 1
  if (!(d[0] > 1)) return CLWH_ERR_INVALID_VALUE;
  const size_t bytes = d[0] * d[1] * d[2] * (size_t)channels * es;
  int rc = clwh_mem_create(ctx, bytes, flags, out);
  if (rc != CLWH_OK) return rc;
  return image_describe(*out, d, channels, elem_kind);
}

int clwh_image_wrap(clwh_ctx *ctx, void *device_ptr, const size_t dims[3], int channels, int elem_kind, clwh_mem **out) {
  if (!ctx || !out || !dims || !device_ptr) return CLWH_ERR_INVALID_VALUE;
  *out = nullptr;
  const size_t es = elem_size(elem_kind);
  if (es == 0) return CLWH_ERR_INVALID_VALUE;
  size_t d[3];
  for (int k = 0; k < 3; ++k) d[k] = dims[k] == 0 ? 1 : dims[k];
  int rc = mem_new(ctx, device_ptr, d[0] * d[1] * d[2] * (size_t)channels * es, false, out);
  if (rc != CLWH_OK) return rc;
  rc = image_describe(*out, d, channels, elem_kind);
  if (rc != CLWH_OK) {
    delete *out;
    *out = nullptr;
  }
  return rc;
}

int clwh_mem_push(clwh_ctx *ctx, clwh_mem *mem, const void *host, size_t bytes) {
  if (!ctx || !mem || !host) return CLWH_ERR_INVALID_VALUE;
  if (bytes != mem->bytes) return CLWH_ERR_SIZE_MISMATCH;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(mem->dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  touch(mem);
  return CLWH_OK;
}

int clwh_mem_pull(clwh_ctx *ctx, clwh_mem *mem, void *host, size_t bytes) {
  if (!ctx || !mem || !host) return CLWH_ERR_INVALID_VALUE;
  if (bytes != mem->bytes) return CLWH_ERR_SIZE_MISMATCH;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(host, mem->dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return check_device_flags(ctx);
}

int clwh_mem_release(clwh_mem *mem) {
  if (!mem) return CLWH_ERR_INVALID_VALUE;
  int rc = CLWH_OK;
  if (mem->owned && mem->dptr) {
    (void)hipSetDevice(mem->ctx->device);
    (void)hipStreamSynchronize(mem->ctx->stream);
    hipError_t e = hipFree(mem->dptr);
    if (e != hipSuccess) {
      g_last_hip_error = (int)e;
      rc = CLWH_ERR_HIP;
    }
  }
  delete mem;
  return rc;
}

int clwh_host_register(void *host, size_t bytes) {
  if (!host || bytes == 0) return CLWH_ERR_INVALID_VALUE;
  HIP_TRY(hipHostRegister(host, bytes, hipHostRegisterDefault));
  return CLWH_OK;
}

int clwh_host_unregister(void *host) {
  if (!host) return CLWH_ERR_INVALID_VALUE;
  HIP_TRY(hipHostUnregister(host));
  return CLWH_OK;
}

void *clwh_mem_device_ptr(clwh_mem *mem) { return mem ? mem->dptr : nullptr; }
size_t clwh_mem_size(clwh_mem *mem) { return mem ? mem->bytes : 0; }
int clwh_mem_mark_dirty(clwh_mem *mem) {
  if (!mem) return CLWH_ERR_INVALID_VALUE;
  touch(mem);
  return CLWH_OK;
}

// ------------------------------------------------------------------------------------------------
// kernels

int clwh_kernel_get(clwh_ctx *ctx, const char *file, const char *entry, const char *prepend, clwh_kernel **out) {
  if (!ctx || !file || !entry || !out) return CLWH_ERR_INVALID_VALUE;
  *out = nullptr;
  // the reference passes paths relative to KERNEL_DIR; accept a directory prefix
  const char *base = std::strrchr(file, '/');
  base = base ? base + 1 : file;
  int id = -1;
  bool needs_tf = false;
  if (!std::strcmp(base, "ray_marching.cl") && !std::strcmp(entry, "render")) { id = CLWH_K_RENDER; needs_tf = true; }
  else if (!std::strcmp(base, "signed_distance_field.cl") && !std::strcmp(entry, "create_base_image")) { id = CLWH_K_SDF_BASE; needs_tf = true; }
  else if (!std::strcmp(base, "signed_distance_field.cl") && !std::strcmp(entry, "create_signed_distance_field")) { id = CLWH_K_SDF_LAYER; }
  else if (!std::strcmp(base, "buffer_reset.cl") && !std::strcmp(entry, "buffer_reset")) { id = CLWH_K_BUFFER_RESET; }
  else if (!std::strcmp(base, "empty.cl") && !std::strcmp(entry, "empty")) { id = CLWH_K_EMPTY; }
  else if (!std::strcmp(base, "reference_volume_figures.cl") && !std::strcmp(entry, "fetch_stats")) { id = CLWH_K_FETCH_STATS; }
  else if (!std::strcmp(base, "reference_volume_clip.cl") && !std::strcmp(entry, "apply_clip")) { id = CLWH_K_APPLY_CLIP; }
  else if (!std::strcmp(base, "histogram.cl") && !std::strcmp(entry, "tf_sort_values")) { id = CLWH_K_TF_SORT_VALUES; }
  else if (!std::strcmp(base, "histogram.cl") && !std::strcmp(entry, "tf_flush_color_frame")) { id = CLWH_K_TF_FLUSH_COLOR_FRAME; }
  else if (!std::strcmp(base, "volume_filter.cl") && !std::strcmp(entry, "bilateral_filter")) { id = CLWH_K_BILATERAL_FILTER; }
  if (id < 0) return CLWH_ERR_UNKNOWN_KERNEL;
  clwh_kernel *k = new (std::nothrow) clwh_kernel();
  if (!k) return CLWH_ERR_OUT_OF_MEMORY;
  k->ctx = ctx;
  k->id = id;
  const bool have_src = prepend && prepend[0] != '\0';
  if (needs_tf && !have_src) {
    // the reference would fail to compile: is_event_gen is undeclared
    delete k;
    return CLWH_ERR_TF_UNSUPPORTED;
  }
  if (have_src && needs_tf) {
    int rc = clwh_tf_parse(prepend, &k->tf);
    if (rc == CLWH_ERR_TF_UNSUPPORTED) rc = jit_for_source(ctx, prepend, k->jit);  // general fallback: hiprtc
    if (rc != CLWH_OK) {
      delete k;
      return rc;
    }
    k->has_tf = true;
  }
  *out = k;
  return CLWH_OK;
}

int clwh_kernel_release(clwh_kernel *k) {
  if (!k) return CLWH_ERR_INVALID_VALUE;
  delete k;
  return CLWH_OK;
}

int64_t clwh_cache_len(uint32_t X, uint32_t Y, uint32_t Z) {
  return ((int64_t)X * Z * Y + (int64_t)X * Z + X + 1) * 4;
}

int64_t clwh_accum_len(uint32_t width, uint32_t height, int32_t tile_world) {
  if (tile_world < 1) tile_world = 1;
  const int64_t tiles_x = width / 8, tiles_y = height / 8;
  const int64_t tiles_per_row = (tiles_x + tile_world - 1) / tile_world;
  return tiles_y * tiles_per_row * 64;
}

static bool is_image(const clwh_mem *m, int dims_n, int channels, int elem_kind) {
  if (!m || !m->is_image || m->channels != channels || m->elem_kind != elem_kind) return false;
  if (dims_n == 2) return m->dims[2] == 1;
  return true;
}

// layout of a PackedScene's allocation
struct PackedLayout {
  int X, Y, Z, NBX, NBY, NBZ, mshift, MNX, MNY, MNZ;
  size_t records, n_bricks, off_stepb, off_brick_min, off_macro, bytes;
};
static PackedLayout packed_layout(const clwh_mem *volume, int forced_macro_shift) {
  PackedLayout L;
  L.X = (int)volume->dims[0]; L.Y = (int)volume->dims[1]; L.Z = (int)volume->dims[2];
  L.NBX = (L.X + 7) / 8; L.NBY = (L.Y + 7) / 8; L.NBZ = (L.Z + 7) / 8;
  L.records = (size_t)L.NBX * L.NBY * L.NBZ * 512u;
  L.n_bricks = (size_t)L.NBX * L.NBY * L.NBZ;
  L.mshift = macro_cell_shift(L.X, L.Y, L.Z, forced_macro_shift);  // macro cells of the exit certificates
  const int mcell = 1 << L.mshift;
  L.MNX = (L.X + mcell - 1) >> L.mshift; L.MNY = (L.Y + mcell - 1) >> L.mshift; L.MNZ = (L.Z + mcell - 1) >> L.mshift;
  // hit records, the per-step bytes, the per-brick minima (u32, 16-byte aligned), the macro-cell table
  L.off_stepb = L.records * sizeof(uint2);
  L.off_brick_min = (L.records * (sizeof(uint2) + 1u) + 15u) & ~(size_t)15u;
  L.off_macro = (L.off_brick_min + L.n_bricks * sizeof(uint32_t) + 15u) & ~(size_t)15u;
  L.bytes = L.off_macro + (size_t)L.MNX * L.MNY * L.MNZ * 8u;  // eight octant entries per cell
  return L;
}

static bool packed_matches(const PackedScene &p, int device, const PackedLayout &L, const clwh_mem *volume, const clwh_mem *sdf,
                           const TfDev &tf, const std::string &tf_identity) {
  return !p.stale && p.device == device && p.bytes == L.bytes && p.macro_shift == L.mshift && p.vol == volume->dptr &&
         p.sdf == sdf->dptr && p.vol_ver == volume->version() && p.sdf_ver == sdf->version() &&
         !std::memcmp(&p.tf, &tf, sizeof tf) && p.tf_identity == tf_identity;
}

// the bricked step bytes + hit records (packed_volume.hpp) of (volume, SDF, TF): the context's own entry if it still matches,
// else the entry another context of this device has built (its stream's work is ordered behind the build by an event), else
// built here on the context's stream
static int ensure_packed(clwh_ctx *ctx, const clwh_mem *volume, const clwh_mem *sdf, const TfDev &tf, const uint8_t *cls_in,
                         const std::string &tf_identity) {
  const PackedLayout L = packed_layout(volume, ctx->tune_macro_shift);
  if (ctx->scene && packed_matches(*ctx->scene, ctx->device, L, volume, sdf, tf, tf_identity)) return CLWH_OK;
  std::shared_ptr<PackedScene> entry;
  {
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    for (auto it = g_packed_scenes.begin(); it != g_packed_scenes.end();) {
      std::shared_ptr<PackedScene> live = it->lock();
      if (!live) { it = g_packed_scenes.erase(it); continue; }
      if (!entry && packed_matches(*live, ctx->device, L, volume, sdf, tf, tf_identity)) entry = live;
      ++it;
    }
  }
  if (entry) {
    ctx->scene.reset();  // (frees the old entry if this context was its last holder)
    HIP_TRY(hipStreamWaitEvent(ctx->stream, entry->ready, 0));
    ctx->scene = entry;
    return CLWH_OK;
  }
  // build.  A context that is the only holder of an entry of the right size rebuilds in place (a transfer-function flush
  // does not free and allocate 9 bytes per voxel); the stream orders the rebuild behind the kernels that still read it.
  if (ctx->scene && ctx->scene.use_count() == 1 && ctx->scene->bytes == L.bytes) {
    entry = ctx->scene;
  } else {
    ctx->scene.reset();
    entry = std::make_shared<PackedScene>();
    entry->device = ctx->device;
    HIP_TRY(hipMalloc((void **)&entry->data, L.bytes));
    entry->bytes = L.bytes;
    HIP_TRY(hipEventCreateWithFlags(&entry->ready, hipEventDisableTiming));
  }
  ctx->scene.reset();
  {
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    entry->stale = true;  // not adoptable until described below
  }
  RepackArgs r;
  std::memset(&r, 0, sizeof r);
  r.volume = (const int16_t *)volume->dptr;
  r.sdf = (const int8_t *)sdf->dptr;
  r.X = L.X; r.Y = L.Y; r.Z = L.Z;
  r.NBX = L.NBX; r.NBY = L.NBY; r.NBZ = L.NBZ;
  r.grec = reinterpret_cast<uint2 *>(entry->data);
  r.stepb = entry->data + L.off_stepb;
  r.brick_min = reinterpret_cast<uint32_t *>(entry->data + L.off_brick_min);
  HIP_TRY(hipMemsetAsync(r.brick_min, 0xFF, L.n_bricks * sizeof(uint32_t), ctx->stream));
  r.cls_in = cls_in;
  r.tf = tf;
  {
    TimedLaunch t;
    int trc = t.begin(ctx, CLWH_TIMER_REPACK);
    if (trc != CLWH_OK) return trc;
    HIP_TRY(launch_repack(r, ctx->stream));
    HIP_TRY(launch_macro_table(r.brick_min, L.NBX, L.NBY, L.NBZ, entry->data + L.off_macro, L.X, L.Y, L.Z, L.mshift, ctx->stream));
    trc = t.end();
    if (trc != CLWH_OK) return trc;
  }
  HIP_TRY(hipEventRecord(entry->ready, ctx->stream));
  ctx->scene = entry;
  {
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    entry->vol = volume->dptr;
    entry->sdf = sdf->dptr;
    entry->vol_ver = volume->version();
    entry->sdf_ver = sdf->version();
    entry->tf = tf;
    entry->tf_identity = tf_identity;
    entry->macro_shift = L.mshift;
    entry->generation = ++g_packed_generation;
    entry->stale = false;
    bool listed = false;
    for (auto &w : g_packed_scenes)
      if (w.lock() == entry) listed = true;
    if (!listed) g_packed_scenes.push_back(entry);
  }
  return CLWH_OK;
}

static int grow(clwh_ctx *ctx, void **p, size_t *have, size_t need) {
  if (*have >= need) return CLWH_OK;
  if (*p) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(*p));
    *p = nullptr;
    *have = 0;
  }
  HIP_TRY(hipMalloc(p, need));
  *have = need;
  return CLWH_OK;
}

// fill the parts of RenderArgs that describe the frame / tile partition / camera
static void describe_frame(RenderArgs &a, int32_t launch_w, int32_t launch_h, int32_t frame_w, int32_t frame_h,
                           int32_t rank, int32_t world, const float cam_pos[3], const float cam_dir[3]) {
  a.launch_w = launch_w;
  a.launch_h = launch_h;
  a.frame_w = frame_w;
  a.frame_h = frame_h;
  a.tiles_x = launch_w / 8;
  a.tiles_y = launch_h / 8;
  a.tile_rank = rank;
  a.tile_world = world;
  a.tiles_per_row = (a.tiles_x + world - 1) / world;
  a.num_tile_slots = (uint32_t)a.tiles_y * (uint32_t)a.tiles_per_row;
  for (int q = 0; q < 3; ++q) {
    a.cam_pos[q] = cam_pos[q];
    a.cam_dir[q] = cam_dir[q];
  }
}

int clwh_render(clwh_kernel *k, const clwh_render_desc *d) {
  if (!k || !d || k->id != CLWH_K_RENDER || !k->has_tf) return CLWH_ERR_INVALID_VALUE;
  clwh_ctx *ctx = k->ctx;
  if (!is_image(d->volume, 3, 1, CLWH_ELEM_S16) || !is_image(d->sdf, 3, 1, CLWH_ELEM_S8) ||
      !is_image(d->env, 2, 4, CLWH_ELEM_U8))
    return CLWH_ERR_BAD_ARGS;
  if (d->frame && !is_image(d->frame, 2, 4, CLWH_ELEM_U8)) return CLWH_ERR_BAD_ARGS;
  for (int q = 0; q < 3; ++q)
    if (d->volume->dims[q] != d->sdf->dims[q]) return CLWH_ERR_SIZE_MISMATCH;
  if (d->width == 0 || d->height == 0 || (d->width % 8) != 0 || (d->height % 8) != 0) return CLWH_ERR_BAD_NDRANGE;
  if (d->width > 65535u || d->height > 65535u) return CLWH_ERR_BAD_NDRANGE;  // pixel ids are packed x | y << 16
  if (d->env->dims[0] > 32768u || d->env->dims[1] > 32768u) return CLWH_ERR_INVALID_VALUE;  // env_fast.hpp bracket
  if (d->volume->dims[0] > 0x7fffffffu || d->volume->dims[1] > 0x7fffffffu || d->volume->dims[2] > 0x7fffffffu)
    return CLWH_ERR_INVALID_VALUE;
  const int world = d->tile_world < 1 ? 1 : d->tile_world;
  if (d->tile_rank < 0 || d->tile_rank >= world) return CLWH_ERR_INVALID_VALUE;
  if (d->n_seeds < 0 || d->n_seeds > CLWH_MAX_SEEDS) return CLWH_ERR_INVALID_VALUE;
  if (d->n_seeds > 1 && d->contrib) return CLWH_ERR_BAD_ARGS;  // per-pixel contribution output: one seed
  if (d->shading != CLWH_SHADE_LIGHT && d->shading != CLWH_SHADE_AO) return CLWH_ERR_INVALID_VALUE;
  if (d->shading == CLWH_SHADE_AO && d->accum_mode != CLWH_ACCUM_VOXEL_CACHE) return CLWH_ERR_BAD_ARGS;  // compute_ao lives in buffer_volume

  RenderArgs a;
  std::memset(&a, 0, sizeof a);
  a.X = (int32_t)d->volume->dims[0];
  a.Y = (int32_t)d->volume->dims[1];
  a.Z = (int32_t)d->volume->dims[2];
  a.env = (const uint32_t *)d->env->dptr;
  a.env_w = (int32_t)d->env->dims[0];
  a.env_h = (int32_t)d->env->dims[1];
  // get_image_width/height(frame): the frame IMAGE's dims (the reference allocates 2048x1024 whatever
  // the launch size); without a frame, the launch size
  const int32_t fw = d->frame ? (int32_t)d->frame->dims[0] : (int32_t)d->width;
  const int32_t fh = d->frame ? (int32_t)d->frame->dims[1] : (int32_t)d->height;
  describe_frame(a, (int32_t)d->width, (int32_t)d->height, fw, fh, d->tile_rank, world, d->cam_pos, d->cam_dir);
  a.frame = d->frame ? (uint32_t *)d->frame->dptr : nullptr;
  a.mode = d->accum_mode;
  a.shading = d->shading;
  if (a.mode == CLWH_ACCUM_VOXEL_CACHE) {
    if (!d->buffer_volume || d->buffer_volume->bytes < 8) return CLWH_ERR_BAD_ARGS;
    a.cache = (uint32_t *)d->buffer_volume->dptr;
    // compute_light: 4 ushorts per voxel (utility.cl:21); compute_ao: 2 ushorts per voxel (utility.cl:127)
    a.cache_entries = (int64_t)(d->buffer_volume->bytes / (d->shading == CLWH_SHADE_AO ? 4 : 8));
  } else if (a.mode == CLWH_ACCUM_IMAGE_SPACE) {
    const int64_t need = clwh_accum_len(d->width, d->height, world) * 16;
    if (!d->accum || (int64_t)d->accum->bytes < need) return CLWH_ERR_BAD_ARGS;
    a.accum = (float4 *)d->accum->dptr;
  } else {
    return CLWH_ERR_INVALID_VALUE;
  }
  const size_t npx = (size_t)a.launch_w * (size_t)a.launch_h;
  if (d->hit_index) {
    if (d->hit_index->bytes < npx * 8) return CLWH_ERR_SIZE_MISMATCH;
    a.hit_index_out = (int64_t *)d->hit_index->dptr;
  }
  if (d->contrib) {
    if (d->contrib->bytes < npx * 16) return CLWH_ERR_SIZE_MISMATCH;
    a.contrib_out = (uint32_t *)d->contrib->dptr;
  }
  if (d->n_seeds > 0) {
    a.n_seeds = d->n_seeds;
    for (int q = 0; q < d->n_seeds; ++q) a.seeds[q] = d->seeds[q];
  } else {
    a.n_seeds = 1;
    a.seeds[0] = d->seed;
  }
  const uint8_t *cls_in = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  if (k->jit) {
    int jrc = ensure_classes(ctx, k->jit, d->volume, a.tf, &cls_in);
    if (jrc != CLWH_OK) return jrc;
  } else {
    tf_to_dev(k->tf, a.tf);
  }
  a.tf.literal_gradient_taps = ctx->tune_literal_gradient;
  a.step_min_lanes = ctx->tune_step_min_lanes;
  a.refill_min_lanes = ctx->tune_refill_min_lanes;
  a.force_long_launch = ctx->tune_force_long_launch;
  a.bounce_rays = ctx->tune_bounce_rays;
  a.bounce_max_blocks = ctx->tune_bounce_max_blocks;
  a.unit_group = ctx->tune_unit_group;
  a.unit_block_log2 = ctx->tune_unit_block_log2;
  a.unit_affinity = ctx->tune_unit_affinity;
  a.unit_queues = ctx->tune_unit_queues;

  HIP_TRY(hipSetDevice(ctx->device));
  int rc = ensure_packed(ctx, d->volume, d->sdf, a.tf, cls_in, k->jit ? k->jit->source : std::string());
  if (rc != CLWH_OK) return rc;
  {
    const PackedLayout L = packed_layout(d->volume, ctx->tune_macro_shift);
    const uint8_t *packed = ctx->scene->data;
    a.grec = reinterpret_cast<const uint2 *>(packed);
    a.NBX = L.NBX;
    a.NBY = L.NBY;
    a.stepb = packed + L.off_stepb;
    a.volume_lin = (const int16_t *)d->volume->dptr;
    a.sdf_lin = (const int8_t *)d->sdf->dptr;
    a.macro = packed + L.off_macro;
    a.macro_shift = L.mshift;
    a.MNX = L.MNX; a.MNY = L.MNY; a.MNZ = L.MNZ;
    // An exit certificate proves "this march leaves the volume without a Hit"; a position with a coordinate == dimension or NaN
    // reads the border texel (value 0), so tables under which value 0 can be an event keep marching literally.
    bool zero_may_hit = a.tf.border_class != 0;
    for (int q = 0; q < a.tf.n && a.tf.uses_gradient && !a.tf.opaque; ++q)
      if (a.tf.rules[q].v_lo <= 0 && 0 <= a.tf.rules[q].v_hi) zero_may_hit = true;
    const int cert_auto = std::min(12 << (a.macro_shift - 4), 48);  // re-swept in round 3 with the stronger certificates: 8 / 12 / 16 -> 3.69 / 3.59 / 3.6-3.9 ms
    a.cert_min_step = zero_may_hit ? 0 : (ctx->tune_cert_min_step >= 0 ? ctx->tune_cert_min_step : cert_auto);
  }

  // ---- primary hits of this camera: rebuilt only when something they depend on changed
  const size_t slots = (size_t)a.num_tile_slots * 64u;
  rc = grow(ctx, (void **)&ctx->pix_slot, &ctx->pix_slot_bytes, slots * sizeof(uint32_t));
  if (rc != CLWH_OK) return rc;
  rc = grow(ctx, (void **)&ctx->hits, &ctx->hits_bytes, slots * sizeof(HitRec));
  if (rc != CLWH_OK) return rc;
  if (!ctx->render_counters) HIP_TRY(hipMalloc((void **)&ctx->render_counters, clwh_ctx::kRenderCounters * sizeof(uint32_t)));
  if (!ctx->sticky_flags) {
    HIP_TRY(hipMalloc((void **)&ctx->sticky_flags, 64));
    HIP_TRY(hipMemsetAsync(ctx->sticky_flags, 0, 64, ctx->stream));
  }
  a.sticky_flags = ctx->sticky_flags;
  a.pix_slot = ctx->pix_slot;
  a.hits = ctx->hits;
  a.counters = ctx->render_counters;

  clwh_ctx::PrimaryKey key;
  std::memset(&key, 0, sizeof key);
  for (int q = 0; q < 3; ++q) {
    key.cam_pos[q] = a.cam_pos[q];
    key.cam_dir[q] = a.cam_dir[q];
  }
  key.frame_w = a.frame_w; key.frame_h = a.frame_h;
  key.launch_w = a.launch_w; key.launch_h = a.launch_h;
  key.tile_rank = a.tile_rank; key.tile_world = a.tile_world;
  key.cache_entries = a.cache_entries;
  key.mode = a.mode;
  key.shading = a.shading;
  key.packed_generation = ctx->scene->generation;
  key.env = d->env->dptr;
  key.env_version = d->env->version();
  key.env_w = a.env_w; key.env_h = a.env_h;
  if (!ctx->primary_valid || std::memcmp(&key, &ctx->primary_key, sizeof key) != 0 || a.hit_index_out) {
    ctx->primary_valid = false;
    HIP_TRY(hipMemsetAsync(ctx->render_counters, 0, clwh_ctx::kRenderCounters * sizeof(uint32_t), ctx->stream));
    TimedLaunch t;
    rc = t.begin(ctx, CLWH_TIMER_PRIMARY);
    if (rc != CLWH_OK) return rc;
    HIP_TRY(launch_primary(a, ctx->stream));
    rc = t.end();
    if (rc != CLWH_OK) return rc;
    // the camera's hit count follows on the stream into page-locked memory; nobody waits for it
    if (!ctx->host_n_hits) {
      HIP_TRY(hipHostMalloc((void **)&ctx->host_n_hits, 64, hipHostMallocDefault));
      HIP_TRY(hipEventCreateWithFlags(&ctx->n_hits_event, hipEventDisableTiming));
    }
    HIP_TRY(hipMemcpyAsync(ctx->host_n_hits, ctx->render_counters, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->n_hits_event, ctx->stream));
    ctx->n_hits_in_flight = true;
    ctx->vox_plan_valid = false;
    ctx->primary_n_hits_known = false;
    ctx->primary_key = key;
    ctx->primary_valid = true;
  }
  // No host round trip per camera (the reference's queue is one in-order queue without a readback between camera and pass,
  // app/renderer.cpp:145-150).  The kernels read the hit count from counters[0] themselves; the host only needs a bound of it
  // for grid and buffer sizes, and uses the real count as soon as the copy above has arrived by itself.
  if (ctx->n_hits_in_flight && !ctx->primary_n_hits_known) {
    if (hipEventQuery(ctx->n_hits_event) == hipSuccess) {
      ctx->primary_n_hits = *ctx->host_n_hits;
      ctx->primary_n_hits_known = true;
      ctx->n_hits_in_flight = false;
      ctx->last_known_n_hits = ctx->primary_n_hits;
    } else {
      (void)hipGetLastError();  // hipErrorNotReady is an answer, not a failure: keep it out of the launches' error checks
    }
  }
  const bool ao = d->shading == CLWH_SHADE_AO;
  a.n_hits_on_device = !ctx->primary_n_hits_known;
  a.n_hits = ctx->primary_n_hits_known ? ctx->primary_n_hits : (uint32_t)std::min<size_t>(slots, 0xFFFFFFFFu);  // else: upper bound
  // what the count is LIKELY to be, for the scheduling class of the launch and the size of the fix-up buffer: the count itself, else
  // twice the last camera's (a camera move rarely doubles the hit pixels), never below a quarter of the pixels
  uint32_t n_hits_est = a.n_hits;
  if (!ctx->primary_n_hits_known && ctx->last_known_n_hits != 0)
    n_hits_est = (uint32_t)std::min<uint64_t>(a.n_hits, std::max<uint64_t>(2ull * ctx->last_known_n_hits, slots / 4u));
  a.n_hits_estimate = n_hits_est;
  // the bounce kernel's queue arithmetic needs ceil(hits / 64) x seeds below 2^24 (64 seeds: 16.7 M hit pixels)
  if ((uint64_t)(((a.n_hits + 63u) >> 6) + 8u * (1u << ctx->tune_unit_block_log2)) * (uint64_t)a.n_seeds >= (1ull << 24)) return CLWH_ERR_INVALID_VALUE;

  HIP_TRY(hipMemsetAsync(ctx->render_counters + 1, 0, (clwh_ctx::kRenderCounters - 1) * sizeof(uint32_t), ctx->stream));
  if (a.contrib_out) HIP_TRY(hipMemsetAsync(a.contrib_out, 0, npx * 16, ctx->stream));  // misses contribute nothing

  if (d->resolve_only) {
    if (!a.frame) return CLWH_ERR_BAD_ARGS;
  } else if (ao) {
    // ---- ambient occlusion (compute_ao, ray_marching.cl:104-149): one lane per hit, its passes one after the other
    TimedLaunch t;
    rc = t.begin(ctx, CLWH_TIMER_AO);
    if (rc != CLWH_OK) return rc;
    HIP_TRY(launch_ao(a, ctx->stream));
    rc = t.end();
    if (rc != CLWH_OK) return rc;
  } else {
    // ---- the pass: every (hit, seed) item
    // fix-up records for environment lookups the fast path cannot certify.  Expected rate: 4e-6 x (0.16 w + 0.32 h) per lookup, at
    // most two lookups per item (0.5 % at 4096x2048, 4 % at the 32768 limit); room for three times that, never less than 1/64 of the
    // items -- of the ESTIMATED item count while the camera's hit count is still on its way (an overflow is reported, not silent)
    const double rate = std::min(1.0, std::max(1.0 / 64.0, 3.0 * 8.0e-6 * (0.16 * a.env_w + 0.32 * a.env_h)));
    const size_t fix_cap = std::max<size_t>((size_t)((double)n_hits_est * (double)a.n_seeds * rate), 4096u);
    rc = grow(ctx, (void **)&ctx->fixups, &ctx->fixups_bytes, fix_cap * 128u);
    if (rc != CLWH_OK) return rc;
    a.fixups = ctx->fixups;
    a.fixup_capacity = (uint32_t)std::min<size_t>(ctx->fixups_bytes / 128u, 0x7fffffffu);
    // a voxel-cache launch of several seeds deals its tokens out beforehand (render_kernels.hip "planned voxel-cache launches");
    // one seed per launch -- the reference's call pattern, and the per-pixel contribution output of the parity tests -- keeps the
    // reference's token-per-sample protocol
    const bool planned = a.mode == CLWH_ACCUM_VOXEL_CACHE && a.n_seeds > 1;
    if (a.mode == CLWH_ACCUM_IMAGE_SPACE || planned) {
      // one 64-bit delta per hit; k_commit folds a launch's deltas into the accumulator and leaves them zero for the next launch
      const size_t before = ctx->delta_bytes;
      rc = grow(ctx, (void **)&ctx->delta, &ctx->delta_bytes, std::max<size_t>(slots, 1) * sizeof(unsigned long long));
      if (rc != CLWH_OK) return rc;
      a.delta = ctx->delta;
      if (ctx->delta_bytes != before) HIP_TRY(hipMemsetAsync(ctx->delta, 0, ctx->delta_bytes, ctx->stream));
    }
    if (planned) {
      // keys_in | keys_sorted (int64) | iota | order | grants (u32), `cap` elements each; sorted once per camera
      const size_t cap = slots;
      const uint8_t *plan_before = ctx->vox_plan;
      rc = grow(ctx, (void **)&ctx->vox_plan, &ctx->vox_plan_bytes, cap * (2 * sizeof(int64_t) + 3 * sizeof(uint32_t)));
      if (rc != CLWH_OK) return rc;
      if (ctx->vox_plan != plan_before) ctx->vox_plan_valid = false;
      const size_t have = ctx->vox_plan_bytes / (2 * sizeof(int64_t) + 3 * sizeof(uint32_t));
      int64_t *keys_in = reinterpret_cast<int64_t *>(ctx->vox_plan), *keys = keys_in + have;
      uint32_t *iota = reinterpret_cast<uint32_t *>(keys + have), *order = iota + have, *grants = order + have;
      if (!ctx->vox_plan_valid) {
        const uint32_t n = a.n_hits;  // the count, or its bound (then the tail sorts behind every real hit)
        HIP_TRY(launch_vox_keys(a, keys_in, iota, n, ctx->stream));
        size_t need = 0;
        HIP_TRY(sort_entry_pairs(nullptr, need, keys_in, keys, iota, order, n, 39u, ctx->stream));
        rc = grow(ctx, &ctx->vox_temp, &ctx->vox_temp_bytes, std::max<size_t>(need, 16));
        if (rc != CLWH_OK) return rc;
        size_t tb = ctx->vox_temp_bytes;
        HIP_TRY(sort_entry_pairs(ctx->vox_temp, tb, keys_in, keys, iota, order, n, 39u, ctx->stream));
        ctx->vox_plan_valid = true;
        ctx->vox_plan_n = n;
      }
      HIP_TRY(launch_vox_grant(a, keys, order, ctx->vox_plan_n, grants, ctx->stream));
      a.grants = grants;
    }
    TimedLaunch t;
    rc = t.begin(ctx, CLWH_TIMER_BOUNCE);
    if (rc != CLWH_OK) return rc;
    HIP_TRY(launch_bounce(a, ctx->stream));
    rc = t.end();
    if (rc != CLWH_OK) return rc;
#ifdef CLVR_BOUNCE_STATS  // experiment builds only (CLVR_EXTRA_HIPCC_FLAGS=-DCLVR_BOUNCE_STATS): scheduling statistics of the launch
    {
      uint32_t h[22];
      HIP_TRY(hipMemcpyAsync(h, ctx->render_counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      std::fprintf(stderr, "[bounce stats] items=%llu step_iters=%u avg_march_lanes=%.2f event_phases=%u avg_event_lanes=%.2f "
                   "refills=%u avg_refill=%.2f events start/exit/hit/none=%u/%u/%u/%u cert_phases=%u avg_cert_lanes=%.2f cert_granted=%u\n",
                   (unsigned long long)h[0] * (unsigned long long)a.n_seeds, h[8], h[8] ? (double)h[9] / h[8] : 0.0, h[10],
                   h[10] ? (double)h[11] / h[10] : 0.0, h[12], h[12] ? (double)h[13] / h[12] : 0.0, h[14], h[15], h[16], h[17], h[18],
                   h[18] ? (double)h[19] / h[18] : 0.0, h[20]);
      if (h[21]) std::fprintf(stderr, "[bounce stats] two rays per lane: %u swap points\n", h[21]);
    }
#endif
    TimedLaunch tf;
    rc = tf.begin(ctx, CLWH_TIMER_FIXUP);
    if (rc != CLWH_OK) return rc;
    HIP_TRY(launch_env_fixup(a, ctx->stream));
    HIP_TRY(launch_commit(a, ctx->stream));
    if (planned) HIP_TRY(launch_commit_voxel(a, ctx->stream));
    rc = tf.end();
    if (rc != CLWH_OK) return rc;
    ctx->fixup_overflow_pending = true;
  }
  if ((d->write_frame || d->resolve_only) && a.frame) {
    TimedLaunch t;
    rc = t.begin(ctx, CLWH_TIMER_RESOLVE);
    if (rc != CLWH_OK) return rc;
    HIP_TRY(launch_resolve(a, ctx->stream));
    rc = t.end();
    if (rc != CLWH_OK) return rc;
  }
  if (d->frame) touch(d->frame);
  return CLWH_OK;
}

int clwh_accum_resolve(clwh_ctx *ctx, clwh_mem *accum_all, int32_t tile_world, uint32_t width, uint32_t height,
                       clwh_mem *frame, clwh_mem *env, const float cam_pos[3], const float cam_dir[3]) {
  if (!ctx || !accum_all || !frame || !env || !cam_pos || !cam_dir || tile_world < 1) return CLWH_ERR_INVALID_VALUE;
  if (!is_image(frame, 2, 4, CLWH_ELEM_U8) || !is_image(env, 2, 4, CLWH_ELEM_U8)) return CLWH_ERR_BAD_ARGS;
  if (width == 0 || height == 0 || (width % 8) || (height % 8)) return CLWH_ERR_BAD_NDRANGE;
  const int64_t need = clwh_accum_len(width, height, tile_world) * 16 * tile_world;
  if ((int64_t)accum_all->bytes < need) return CLWH_ERR_SIZE_MISMATCH;
  RenderArgs a;
  std::memset(&a, 0, sizeof a);
  describe_frame(a, (int32_t)width, (int32_t)height, (int32_t)frame->dims[0], (int32_t)frame->dims[1], 0, tile_world,
                 cam_pos, cam_dir);
  a.frame = (uint32_t *)frame->dptr;
  a.env = (const uint32_t *)env->dptr;
  a.env_w = (int32_t)env->dims[0];
  a.env_h = (int32_t)env->dims[1];
  HIP_TRY(hipSetDevice(ctx->device));
  {
    TimedLaunch t;
    int trc = t.begin(ctx, CLWH_TIMER_RESOLVE);
    if (trc != CLWH_OK) return trc;
    HIP_TRY(launch_accum_resolve(a, (const float4 *)accum_all->dptr, ctx->stream));
    trc = t.end();
    if (trc != CLWH_OK) return trc;
  }
  touch(frame);
  return CLWH_OK;
}

int clwh_accum_resolve_tiles(clwh_ctx *ctx, clwh_mem *accum, int32_t tile_rank, int32_t tile_world, uint32_t width, uint32_t height,
                             clwh_mem *tiles_rgba8, clwh_mem *env, const float cam_pos[3], const float cam_dir[3]) {
  if (!ctx || !accum || !tiles_rgba8 || !env || !cam_pos || !cam_dir || tile_world < 1 || tile_rank < 0 || tile_rank >= tile_world)
    return CLWH_ERR_INVALID_VALUE;
  if (!is_image(env, 2, 4, CLWH_ELEM_U8)) return CLWH_ERR_BAD_ARGS;
  if (width == 0 || height == 0 || (width % 8) || (height % 8)) return CLWH_ERR_BAD_NDRANGE;
  const int64_t n = clwh_accum_len(width, height, tile_world);
  if ((int64_t)accum->bytes < n * 16 || (int64_t)tiles_rgba8->bytes < n * 4) return CLWH_ERR_SIZE_MISMATCH;
  RenderArgs a;
  std::memset(&a, 0, sizeof a);
  describe_frame(a, (int32_t)width, (int32_t)height, (int32_t)width, (int32_t)height, tile_rank, tile_world, cam_pos, cam_dir);
  a.env = (const uint32_t *)env->dptr;
  a.env_w = (int32_t)env->dims[0];
  a.env_h = (int32_t)env->dims[1];
  HIP_TRY(hipSetDevice(ctx->device));
  {
    TimedLaunch t;
    int trc = t.begin(ctx, CLWH_TIMER_RESOLVE);
    if (trc != CLWH_OK) return trc;
    HIP_TRY(launch_accum_resolve_tiles(a, (const float4 *)accum->dptr, (uint32_t *)tiles_rgba8->dptr, ctx->stream));
    trc = t.end();
    if (trc != CLWH_OK) return trc;
  }
  touch(tiles_rgba8);
  return CLWH_OK;
}

int clwh_frame_from_tiles(clwh_ctx *ctx, clwh_mem *tiles_all, int32_t tile_world, uint32_t width, uint32_t height, clwh_mem *frame) {
  if (!ctx || !tiles_all || !frame || tile_world < 1) return CLWH_ERR_INVALID_VALUE;
  if (!is_image(frame, 2, 4, CLWH_ELEM_U8)) return CLWH_ERR_BAD_ARGS;
  if (width == 0 || height == 0 || (width % 8) || (height % 8)) return CLWH_ERR_BAD_NDRANGE;
  if ((int64_t)tiles_all->bytes < clwh_accum_len(width, height, tile_world) * 4 * tile_world) return CLWH_ERR_SIZE_MISMATCH;
  const float zero[3] = {0.0f, 0.0f, 0.0f};
  RenderArgs a;
  std::memset(&a, 0, sizeof a);
  describe_frame(a, (int32_t)width, (int32_t)height, (int32_t)frame->dims[0], (int32_t)frame->dims[1], 0, tile_world, zero, zero);
  a.frame = (uint32_t *)frame->dptr;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(launch_frame_from_tiles(a, (const uint32_t *)tiles_all->dptr, ctx->stream));
  touch(frame);
  return CLWH_OK;
}

int clwh_ctx_invalidate_derived(clwh_ctx *ctx, int what) {
  if (!ctx) return CLWH_ERR_INVALID_VALUE;
  if ((what & CLWH_DERIVED_SCENE) && ctx->scene) {
    // this context rebuilds at its next render and nobody adopts the old copy any more; contexts that already share it keep it
    // until their own key changes (memory rewritten behind the shim's back is what clwh_mem_mark_dirty is for: every alias sees it)
    {
      std::lock_guard<std::mutex> lock(g_registry_mutex);
      ctx->scene->stale = true;
    }
    if (ctx->scene.use_count() > 1) ctx->scene.reset();  // the only holder keeps the allocation and rebuilds in place
  }
  if (what & (CLWH_DERIVED_SCENE | CLWH_DERIVED_CAMERA)) ctx->primary_valid = false;
  return CLWH_OK;
}

int clwh_ctx_scene_info(clwh_ctx *ctx, uint64_t *scene_id, uint64_t *bytes, int32_t *holders) {
  if (!ctx) return CLWH_ERR_INVALID_VALUE;
  std::lock_guard<std::mutex> lock(g_registry_mutex);
  if (scene_id) *scene_id = ctx->scene ? ctx->scene->generation : 0;
  if (bytes) *bytes = ctx->scene ? ctx->scene->bytes : 0;
  if (holders) *holders = ctx->scene ? (int32_t)ctx->scene.use_count() : 0;
  return CLWH_OK;
}

int clwh_buffer_reset(clwh_ctx *ctx, clwh_mem *buffer_volume) {
  if (!ctx || !buffer_volume) return CLWH_ERR_INVALID_VALUE;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemsetAsync(buffer_volume->dptr, 0, buffer_volume->bytes, ctx->stream));
  touch(buffer_volume);
  return CLWH_OK;
}

// ------------------------------------------------------------------------------------------------
// SDF

static int sdf_max_iterations(const clwh_mem *v) {
  size_t m = std::max(v->dims[0], std::max(v->dims[1], v->dims[2])) / 2;  // signed_distance_field.cpp:11
  return (int)std::min<size_t>(m, 127);
}

// the byte front: one launch per layer over the active 8x8x8 tiles (sdf_kernels.hip); CLWH_TUNE_SDF=front
static int sdf_build_front(clwh_ctx *ctx, SdfArgs &b, std::vector<int32_t> &settled) {
  const int X = b.X, Y = b.Y, Z = b.Z;
  const int kSlots = (int)settled.size();
  int rc;
  const int TX = (X + 7) / 8, TY = (Y + 7) / 8, TZ = (Z + 7) / 8;
  const size_t n_tiles = (size_t)TX * TY * TZ;
  rc = grow(ctx, (void **)&ctx->sdf_flags, &ctx->sdf_flags_bytes, 4 * n_tiles);
  if (rc != CLWH_OK) return rc;
  HIP_TRY(hipMemsetAsync(ctx->sdf_flags, 0, 4 * n_tiles, ctx->stream));
  uint8_t *flags[3] = {ctx->sdf_flags, ctx->sdf_flags + n_tiles, ctx->sdf_flags + 2 * n_tiles};
  HIP_TRY(launch_sdf_base_front(b, flags[1], TX, TY, ctx->stream));  // layer 1 reads flags[1 % 3]

  SdfFrontArgs a;
  std::memset(&a, 0, sizeof a);
  a.sdf = b.ping;
  a.X = X; a.Y = Y; a.Z = Z;
  a.TX = TX; a.TY = TY; a.TZ = TZ;
  a.max_iterations = b.max_iterations;
  a.counters = ctx->sdf_counters;
  a.tile_done = ctx->sdf_flags + 3 * n_tiles;

  // layers that can still settle a voxel: i + 1 < max_iterations; the host looks at the per-layer
  // counts every kSdfLayersPerCheck launches and stops once a layer settled nothing (nothing can change after it)
#ifndef CLVR_SDF_LAYERS_PER_CHECK
#define CLVR_SDF_LAYERS_PER_CHECK 32  // measured 16 / 32 / 64 / 128: 5.31 / 5.21 / 5.14 / 5.19 ms for the 512^3 build
#endif
  constexpr int kSdfLayersPerCheck = CLVR_SDF_LAYERS_PER_CHECK;
  const int last_layer = a.max_iterations - 2;
  int i = 1;
  bool quiet = false;
  while (i <= last_layer && !quiet) {
    const int chunk_end = std::min(last_layer, i + kSdfLayersPerCheck - 1);
    for (; i <= chunk_end; ++i) {
      a.iteration = i;
      a.flags_cur = flags[i % 3];
      a.flags_next = flags[(i + 1) % 3];
      a.flags_clear = flags[(i + 2) % 3];
      HIP_TRY(launch_sdf_front(a, ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync(settled.data(), ctx->sdf_counters, kSlots * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int j = 1; j < i; ++j)
      if (settled[j] == 0) quiet = true;
  }
  if (i <= 1) {  // no layer ran (max_iterations <= 2): still need the base counts
    HIP_TRY(hipMemcpyAsync(settled.data(), ctx->sdf_counters, kSlots * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return CLWH_OK;
}

// the bit-parallel build: event bits -> seeds + base image -> eight layers per launch on one bit per voxel (sdf_kernels.hip)
static int sdf_build_bits(clwh_ctx *ctx, SdfArgs &b, std::vector<int32_t> &settled) {
  const int X = b.X, Y = b.Y, Z = b.Z;
  const int kSlots = (int)settled.size();
  SdfBitArgs a;
  std::memset(&a, 0, sizeof a);
  a.X = X; a.Y = Y; a.Z = Z;
  a.WP = 2 * ((X + 63) / 64);
  const int waves = ctx->tune_sdfbit_waves;
  sdfbit_block_grid(X, Y, Z, waves, &a.BX, &a.BY, &a.BZ, &a.core_z);
  const size_t words = (size_t)a.WP * (size_t)Y * (size_t)Z;
  const size_t n_blocks = (size_t)a.BX * a.BY * a.BZ;
  if (n_blocks >= (1ull << 31)) return CLWH_ERR_INVALID_VALUE;
  // a voxel D corner moves from the nearest seed settles to D + 1 while D + 1 < max_iterations: max_iterations - 2 layers
  const int total = b.max_iterations - 2;
  const int n_launches = total > 0 ? (total + 7) / 8 : 0;
  // scratch: event bits (x-fastest rows), two reached-set buffers (tiled by region, padded to whole regions), the list of active
  // blocks, per-launch {count, head}, the regions' states and wake stamps
  const size_t tiled = n_blocks * (size_t)(2 * 48 * a.core_z);
  const size_t small = (n_blocks + 2 * (size_t)(n_launches + 1) + 1) & ~(size_t)1;  // even: the planes behind it take 8-byte atomics
  const size_t scratch_words = words + 2 * tiled + small + 7 * tiled;  // ... and the seven bit planes of the layer index
  int rc = grow(ctx, (void **)&ctx->sdf_bits, &ctx->sdf_bits_bytes, scratch_words * sizeof(uint32_t) + 2 * n_blocks);
  if (rc != CLWH_OK) return rc;
  uint32_t *ev = ctx->sdf_bits, *reached[2] = {ctx->sdf_bits + words, ctx->sdf_bits + words + tiled};  // words and tiled are even: 8-byte aligned
  uint32_t *list = ctx->sdf_bits + words + 2 * tiled, *queue = list + n_blocks;
  a.planes = ctx->sdf_bits + words + 2 * tiled + small;
  a.plane_words = tiled;
  HIP_TRY(hipMemsetAsync(a.planes, 0, 7 * tiled * sizeof(uint32_t), ctx->stream));
  a.sdf = b.ping;
  a.ev = ev;
  a.list = list;
  a.state = reinterpret_cast<uint8_t *>(ctx->sdf_bits + scratch_words);
  a.wake = a.state + n_blocks;
  a.presence = ctx->sdf_counters;
  HIP_TRY(hipMemsetAsync(reached[0], 0, 2 * tiled * sizeof(uint32_t), ctx->stream));  // rows nobody ever writes (beyond the volume, never reached) read as empty in both buffers
  HIP_TRY(hipMemsetAsync(queue, 0, 2 * (size_t)(n_launches + 1) * sizeof(uint32_t), ctx->stream));
  HIP_TRY(hipMemsetAsync(a.wake, 0, n_blocks, ctx->stream));
  HIP_TRY(launch_sdfbit_events(b, ev, a.WP, ctx->stream));
  a.r_out = reached[0];
  HIP_TRY(launch_sdfbit_seed(a, ctx->stream));
  a.r_in = reached[0];
  HIP_TRY(launch_sdfbit_state(a, ctx->stream));
#ifdef CLVR_SDFBIT_TIMING
  static unsigned long long *d_timing = nullptr;
  if (!d_timing) HIP_TRY(hipMalloc((void **)&d_timing, 8 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d_timing, 0, 8 * sizeof(unsigned long long), ctx->stream));
  a.timing = d_timing;
#endif
  int t = 0;
  for (int r0 = 0; r0 < total; r0 += 8, ++t) {
    a.r0 = r0;
    a.launch = t;
    a.steps = std::min(8, total - r0);
    a.r_in = reached[t & 1];
    a.r_out = reached[(t + 1) & 1];
    a.list_count = queue + 2 * t;
    a.list_head = queue + 2 * t + 1;
    const bool rec_lds = waves == 8 && ctx->tune_sdfbit_rec_lds != 0;
    HIP_TRY(launch_sdfbit_layers(a, waves, (unsigned)ctx->tune_sdfbit_grid * (waves == 16 ? 1u : (rec_lds ? 3u : 2u)) / 2u, rec_lds, ctx->stream));
  }
  // the values, once: the reached set after the last launch is the one it wrote (regions complete earlier are complete in both)
  HIP_TRY(launch_sdfbit_expand(a, reached[t & 1], b.max_iterations, ctx->stream));
  HIP_TRY(hipMemcpyAsync(settled.data(), ctx->sdf_counters, kSlots * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
#ifdef CLVR_SDFBIT_TIMING
  {
    unsigned long long tm[8];
    HIP_TRY(hipMemcpy(tm, d_timing, sizeof tm, hipMemcpyDeviceToHost));
    const double n = tm[0] ? (double)tm[0] : 1.0;
    std::fprintf(stderr, "sdfbit timing: %llu regions (%llu interior); per region, us: fetch %.2f load %.2f steps %.2f store+values %.2f tail %.2f\n", tm[0], tm[6],
                 tm[1] / n / 100.0, tm[2] / n / 100.0, tm[3] / n / 100.0, tm[4] / n / 100.0, tm[5] / n / 100.0);
  }
#endif
  if (std::getenv("CLWH_DEBUG_SDFBIT")) {  // regions each launch worked on
    std::vector<uint32_t> q(2 * (size_t)(n_launches + 1));
    HIP_TRY(hipMemcpy(q.data(), queue, q.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "sdfbit: %zu regions of 64 x 48 x %d voxels, active per launch:", n_blocks, a.core_z);
    for (int l = 0; l < n_launches; ++l) std::fprintf(stderr, " %u", q[2 * l]);
    std::fprintf(stderr, "\n");
  }
  return CLWH_OK;
}

int clwh_sdf_build(clwh_ctx *ctx, clwh_mem *volume, const char *tf_source, clwh_mem *sdf, int32_t *n_launches) {
  if (!ctx || !volume || !tf_source || !sdf) return CLWH_ERR_INVALID_VALUE;
  if (!is_image(volume, 3, 1, CLWH_ELEM_S16) || !is_image(sdf, 3, 1, CLWH_ELEM_S8)) return CLWH_ERR_BAD_ARGS;
  for (int q = 0; q < 3; ++q)
    if (volume->dims[q] != sdf->dims[q]) return CLWH_ERR_SIZE_MISMATCH;
  if (volume->dims[1] > 65535 || volume->dims[2] > 65535) return CLWH_ERR_INVALID_VALUE;
  HIP_TRY(hipSetDevice(ctx->device));
  clwh_tf tf;
  TfDev tfdev;
  const uint8_t *cls_in = nullptr;
  int rc = clwh_tf_parse(tf_source, &tf);
  if (rc == CLWH_OK) {
    tf_to_dev(tf, tfdev);
  } else if (rc == CLWH_ERR_TF_UNSUPPORTED) {
    std::shared_ptr<JitTf> jit;
    rc = jit_for_source(ctx, tf_source, jit);
    if (rc == CLWH_OK) rc = ensure_classes(ctx, jit, volume, tfdev, &cls_in);
  }
  if (rc != CLWH_OK) return rc;

  const int X = (int)volume->dims[0], Y = (int)volume->dims[1], Z = (int)volume->dims[2];
  constexpr int kSlots = 160;
  if (!ctx->sdf_counters) HIP_TRY(hipMalloc((void **)&ctx->sdf_counters, kSlots * sizeof(int32_t)));
  HIP_TRY(hipMemsetAsync(ctx->sdf_counters, 0, kSlots * sizeof(int32_t), ctx->stream));
  std::vector<int32_t> settled(kSlots, 0);
  const int max_iterations = sdf_max_iterations(volume);
  SdfArgs b;
  std::memset(&b, 0, sizeof b);
  b.volume = (const int16_t *)volume->dptr;
  b.X = X; b.Y = Y; b.Z = Z;
  b.ping = (int8_t *)sdf->dptr;
  b.max_iterations = max_iterations;
  b.counters = ctx->sdf_counters;  // [0] != 0: some |v| == 1
  b.tf = tfdev;
  b.cls_in = cls_in;
  if (ctx->tune_sdf_front) {
    rc = sdf_build_front(ctx, b, settled);
  } else {
    rc = sdf_build_bits(ctx, b, settled);
  }
  if (rc != CLWH_OK) return rc;
  if (n_launches) {
    // what the reference's host loop would have run (app/signed_distance_field.cpp:22-32): its counter at
    // layer i counts the voxels holding i plus those settling to i+1 (< max); it stops at the first odd
    // layer whose counter is zero, or at the bound
    const int bound = max_iterations + (max_iterations % 2) + 1;
    int launches = bound;
    for (int j = 1; j <= bound; ++j) {
      const int64_t holding = (j == 1) ? settled[0] : (j - 1 < kSlots ? settled[j - 1] : 0);
      const int64_t settling = j < kSlots ? settled[j] : 0;
      const bool holding_counts = j < max_iterations;  // a voxel holding j is rewritten only while j < max
      if ((j & 1) && (holding_counts ? holding : 0) + settling == 0) {
        launches = j;
        break;
      }
    }
    *n_launches = launches;
  }
  touch(sdf);
  return CLWH_OK;
}

// ------------------------------------------------------------------------------------------------
// generic launch: the reference kernels' own argument lists, by position

static void normalise3(const size_t in[3], size_t out[3]) {
  for (int k = 0; k < 3; ++k) out[k] = (in && in[k]) ? in[k] : 1;
}

int clwh_launch(clwh_kernel *k, const size_t global_in[3], const size_t local_in[3], const clwh_arg *args, int nargs) {
  if (!k || !global_in || !local_in || (nargs > 0 && !args)) return CLWH_ERR_INVALID_VALUE;
  size_t g[3], l[3];
  normalise3(global_in, g);
  normalise3(local_in, l);
  for (int q = 0; q < 3; ++q)
    if (g[q] < l[q] || (g[q] % l[q]) != 0) return CLWH_ERR_BAD_NDRANGE;  // clw_function.hpp:232-237
  clwh_ctx *ctx = k->ctx;
  HIP_TRY(hipSetDevice(ctx->device));

  auto is_mem = [&](int i) { return args[i].kind == CLWH_ARG_MEM && args[i].v.mem != nullptr; };
  auto as_f32 = [&](int i, float &o) {
    if (args[i].kind == CLWH_ARG_F32) { o = args[i].v.f32; return true; }
    if (args[i].kind == CLWH_ARG_F64) { o = (float)args[i].v.f64; return true; }
    return false;
  };
  auto as_i32 = [&](int i, int32_t &o) {
    switch (args[i].kind) {
      case CLWH_ARG_I32: o = args[i].v.i32; return true;
      case CLWH_ARG_U32: o = (int32_t)args[i].v.u32; return true;
      case CLWH_ARG_I64: o = (int32_t)args[i].v.i64; return true;
      case CLWH_ARG_U64: o = (int32_t)args[i].v.u64; return true;
      default: return false;
    }
  };

  switch (k->id) {
    case CLWH_K_EMPTY:
      return CLWH_OK;

    case CLWH_K_RENDER: {
      // render(frame, volume, sdf, env, buffer_volume, 6 x float, int seed)  ray_marching.cl:152
      if (nargs != 12) return CLWH_ERR_BAD_ARGS;
      for (int i = 0; i < 5; ++i)
        if (!is_mem(i)) return CLWH_ERR_BAD_ARGS;
      clwh_render_desc d;
      std::memset(&d, 0, sizeof d);
      d.frame = args[0].v.mem;
      d.volume = args[1].v.mem;
      d.sdf = args[2].v.mem;
      d.env = args[3].v.mem;
      d.buffer_volume = args[4].v.mem;
      for (int q = 0; q < 3; ++q)
        if (!as_f32(5 + q, d.cam_pos[q]) || !as_f32(8 + q, d.cam_dir[q])) return CLWH_ERR_BAD_ARGS;
      if (!as_i32(11, d.seed)) return CLWH_ERR_BAD_ARGS;
      d.width = (uint32_t)g[0];
      d.height = (uint32_t)g[1];
      d.accum_mode = CLWH_ACCUM_VOXEL_CACHE;
      d.tile_rank = 0;
      d.tile_world = 1;
      d.write_frame = 1;
      return clwh_render(k, &d);
    }

    case CLWH_K_BUFFER_RESET: {
      // buffer_reset(volume, buffer_volume)  buffer_reset.cl:3
      if (nargs != 2 || !is_mem(0) || !is_mem(1)) return CLWH_ERR_BAD_ARGS;
      return clwh_buffer_reset(ctx, args[1].v.mem);
    }

    case CLWH_K_FETCH_STATS: {
      // fetch_stats(volume, int stats[5])  reference_volume_figures.cl:10
      if (nargs != 2 || !is_mem(0) || !is_mem(1)) return CLWH_ERR_BAD_ARGS;
      clwh_mem *v = args[0].v.mem, *st = args[1].v.mem;
      if (!is_image(v, 3, 1, CLWH_ELEM_S16) || st->bytes < 4 * sizeof(int32_t)) return CLWH_ERR_BAD_ARGS;
      if (v->dims[1] > 65535 || v->dims[2] > 65535) return CLWH_ERR_INVALID_VALUE;
      HIP_TRY(launch_fetch_stats((const int16_t *)v->dptr, (int)v->dims[0], (int)v->dims[1], (int)v->dims[2],
                                 (int32_t *)st->dptr, ctx->stream));
      touch(st);
      return CLWH_OK;
    }

    case CLWH_K_TF_SORT_VALUES: {
      // tf_sort_values(volume, uint* frame, int width, int height, float min_v, max_v, min_g, max_g)  histogram.cl:4
      if (nargs != 8 || !is_mem(0) || !is_mem(1)) return CLWH_ERR_BAD_ARGS;
      clwh_mem *v = args[0].v.mem, *fr = args[1].v.mem;
      int32_t w, h;
      float f[4];
      if (!as_i32(2, w) || !as_i32(3, h)) return CLWH_ERR_BAD_ARGS;
      for (int q = 0; q < 4; ++q)
        if (!as_f32(4 + q, f[q])) return CLWH_ERR_BAD_ARGS;
      if (!is_image(v, 3, 1, CLWH_ELEM_S16) || w <= 0 || h <= 0 || fr->bytes < (size_t)w * (size_t)h * 4u) return CLWH_ERR_BAD_ARGS;
      if (v->dims[1] > 65535 || v->dims[2] > 65535) return CLWH_ERR_INVALID_VALUE;
      HIP_TRY(launch_tf_sort_values((const int16_t *)v->dptr, (int)v->dims[0], (int)v->dims[1], (int)v->dims[2],
                                    (uint32_t *)fr->dptr, w, h, f[0], f[1], f[2], f[3], ctx->stream));
      touch(fr);
      return CLWH_OK;
    }

    case CLWH_K_TF_FLUSH_COLOR_FRAME: {
      // tf_flush_color_frame(image2d color_frame, int* frame, int* lookup, int lookup_len)  histogram.cl:34
      if (nargs != 4 || !is_mem(0) || !is_mem(1) || !is_mem(2)) return CLWH_ERR_BAD_ARGS;
      clwh_mem *cf = args[0].v.mem, *fr = args[1].v.mem, *lk = args[2].v.mem;
      int32_t len;
      if (!as_i32(3, len) || !is_image(cf, 2, 4, CLWH_ELEM_U8)) return CLWH_ERR_BAD_ARGS;
      const size_t fw = cf->dims[0], fh = cf->dims[1];
      if (fr->bytes < fw * fh * 4u || len < 0 || lk->bytes < (size_t)len * 4u) return CLWH_ERR_SIZE_MISMATCH;
      HIP_TRY(launch_tf_flush_color_frame((uint32_t *)cf->dptr, (int)fw, (int)fh, (const int32_t *)fr->dptr,
                                          (const int32_t *)lk->dptr, len, ctx->stream));
      touch(cf);
      return CLWH_OK;
    }

    case CLWH_K_BILATERAL_FILTER: {
      // bilateral_filter(reference_volume, buffer)  volume_filter.cl:5
      if (nargs != 2 || !is_mem(0) || !is_mem(1)) return CLWH_ERR_BAD_ARGS;
      clwh_mem *src = args[0].v.mem, *dst = args[1].v.mem;
      if (!is_image(src, 3, 1, CLWH_ELEM_S16) || !is_image(dst, 3, 1, CLWH_ELEM_S16)) return CLWH_ERR_BAD_ARGS;
      if (src->dptr == dst->dptr) return CLWH_ERR_BAD_ARGS;  // a stencil cannot run in place
      for (int q = 0; q < 3; ++q)
        if (src->dims[q] != dst->dims[q]) return CLWH_ERR_SIZE_MISMATCH;  // the reference writes to src's coordinates
      if (src->dims[0] > 0x7fffffffu || src->dims[1] > 8u * 65535u || src->dims[2] > 8u * 65535u) return CLWH_ERR_INVALID_VALUE;
      if (!ctx->bilateral_weights) {
        // utility_filter.cl:43-44,53-55: w = exp(-r2/(2 sigma_s^2) - d^2/(2 sigma_r^2)), float operands, the
        // exponential evaluated in binary64 and rounded once.  d >= 16 must already round to zero.
        const float sigmas = 0.6f, sigmar = 1.0f;
        float host[13 * 17];
        for (int r2 = 0; r2 < 13; ++r2)
          for (int d = 0; d < 17; ++d) {
            const float posd = ((float)r2) / (2 * sigmas * sigmas);
            const float cold = ((float)(d * d)) / (2 * sigmar * sigmar);
            host[r2 * 17 + d] = (float)std::exp((double)(-posd - cold));
          }
        for (int r2 = 0; r2 < 13; ++r2)
          if (host[r2 * 17 + 16] != 0.0f) return CLWH_ERR_INTERNAL_OVERFLOW;
        HIP_TRY(hipMalloc((void **)&ctx->bilateral_weights, sizeof host));
        HIP_TRY(hipMemcpy(ctx->bilateral_weights, host, sizeof host, hipMemcpyHostToDevice));
      }
      HIP_TRY(launch_bilateral_filter((const int16_t *)src->dptr, (int)src->dims[0], (int)src->dims[1], (int)src->dims[2],
                                      (int16_t *)dst->dptr, ctx->bilateral_weights, ctx->stream));
      touch(dst);
      return CLWH_OK;
    }

    case CLWH_K_APPLY_CLIP: {
      // apply_clip(original, clipped, uint start[3], uint len[4])  reference_volume_clip.cl:4
      if (nargs != 4 || !is_mem(0) || !is_mem(1) || !is_mem(2) || !is_mem(3)) return CLWH_ERR_BAD_ARGS;
      clwh_mem *src = args[0].v.mem, *dst = args[1].v.mem, *start = args[2].v.mem, *len = args[3].v.mem;
      if (!is_image(src, 3, 1, CLWH_ELEM_S16) || !is_image(dst, 3, 1, CLWH_ELEM_S16) || start->bytes < 12 || len->bytes < 12)
        return CLWH_ERR_BAD_ARGS;
      if (dst->dims[1] > 65535 || dst->dims[2] > 65535) return CLWH_ERR_INVALID_VALUE;
      HIP_TRY(launch_apply_clip((const int16_t *)src->dptr, (int)src->dims[0], (int)src->dims[1], (int)src->dims[2],
                                (int16_t *)dst->dptr, (int)dst->dims[0], (int)dst->dims[1], (int)dst->dims[2],
                                (const uint32_t *)start->dptr, (const uint32_t *)len->dptr, ctx->stream));
      touch(dst);
      return CLWH_OK;
    }

    case CLWH_K_SDF_BASE: {
      // create_base_image(volume, ping, pong, uint max_iterations)  signed_distance_field.cl:6
      if (nargs != 4 || !is_mem(0) || !is_mem(1) || !is_mem(2) || !k->has_tf) return CLWH_ERR_BAD_ARGS;
      clwh_mem *v = args[0].v.mem, *ping = args[1].v.mem, *pong = args[2].v.mem;
      int32_t max_it;
      if (!as_i32(3, max_it)) return CLWH_ERR_BAD_ARGS;
      if (!is_image(v, 3, 1, CLWH_ELEM_S16) || !is_image(ping, 3, 1, CLWH_ELEM_S8) || !is_image(pong, 3, 1, CLWH_ELEM_S8))
        return CLWH_ERR_BAD_ARGS;
      for (int q = 0; q < 3; ++q)
        if (v->dims[q] != ping->dims[q] || v->dims[q] != pong->dims[q]) return CLWH_ERR_SIZE_MISMATCH;
      if (v->dims[1] > 65535 || v->dims[2] > 65535) return CLWH_ERR_INVALID_VALUE;
      SdfArgs a;
      std::memset(&a, 0, sizeof a);
      a.volume = (const int16_t *)v->dptr;
      a.X = (int32_t)v->dims[0]; a.Y = (int32_t)v->dims[1]; a.Z = (int32_t)v->dims[2];
      a.ping = (int8_t *)ping->dptr;
      a.pong = (int8_t *)pong->dptr;
      a.max_iterations = max_it;
      if (k->jit) {
        int jrc = ensure_classes(ctx, k->jit, v, a.tf, &a.cls_in);
        if (jrc != CLWH_OK) return jrc;
      } else {
        tf_to_dev(k->tf, a.tf);
      }
      HIP_TRY(launch_sdf_base(a, ctx->stream));
      touch(ping);
      touch(pong);
      return CLWH_OK;
    }

    case CLWH_K_SDF_LAYER: {
      // create_signed_distance_field(in, out, int iteration, int* add_buffer, int max_iterations)
      if (nargs != 5 || !is_mem(0) || !is_mem(1) || !is_mem(3)) return CLWH_ERR_BAD_ARGS;
      clwh_mem *in = args[0].v.mem, *outm = args[1].v.mem, *counter = args[3].v.mem;
      int32_t it, max_it;
      if (!as_i32(2, it) || !as_i32(4, max_it)) return CLWH_ERR_BAD_ARGS;
      if (!is_image(in, 3, 1, CLWH_ELEM_S8) || !is_image(outm, 3, 1, CLWH_ELEM_S8) || counter->bytes < 4)
        return CLWH_ERR_BAD_ARGS;
      for (int q = 0; q < 3; ++q)
        if (in->dims[q] != outm->dims[q]) return CLWH_ERR_SIZE_MISMATCH;
      if (in->dims[1] > 65535 || in->dims[2] > 65535) return CLWH_ERR_INVALID_VALUE;
      SdfArgs a;
      std::memset(&a, 0, sizeof a);
      a.X = (int32_t)in->dims[0]; a.Y = (int32_t)in->dims[1]; a.Z = (int32_t)in->dims[2];
      a.ping = (int8_t *)in->dptr;
      a.pong = (int8_t *)outm->dptr;
      a.iteration = it;
      a.max_iterations = max_it;
      a.counter_out = (int32_t *)counter->dptr;
      HIP_TRY(launch_sdf_layer(a, ctx->stream));
      touch(outm);
      touch(counter);
      return CLWH_OK;
    }
  }
  return CLWH_ERR_UNKNOWN_KERNEL;
}

// ------------------------------------------------------------------------------------------------
// diagnostics

const char *clwh_strerror(int status) {
  switch (status) {
    case CLWH_OK: return "CLWH_OK";
    case CLWH_ERR_INVALID_VALUE: return "CLWH_ERR_INVALID_VALUE";
    case CLWH_ERR_NO_DEVICE: return "CLWH_ERR_NO_DEVICE";
    case CLWH_ERR_OUT_OF_MEMORY: return "CLWH_ERR_OUT_OF_MEMORY";
    case CLWH_ERR_HIP: return "CLWH_ERR_HIP";
    case CLWH_ERR_UNKNOWN_KERNEL: return "CLWH_ERR_UNKNOWN_KERNEL";
    case CLWH_ERR_TF_UNSUPPORTED: return "CLWH_ERR_TF_UNSUPPORTED";
    case CLWH_ERR_BAD_ARGS: return "CLWH_ERR_BAD_ARGS";
    case CLWH_ERR_BAD_NDRANGE: return "CLWH_ERR_BAD_NDRANGE";
    case CLWH_ERR_SIZE_MISMATCH: return "CLWH_ERR_SIZE_MISMATCH";
    case CLWH_ERR_INTERNAL_OVERFLOW: return "CLWH_ERR_INTERNAL_OVERFLOW";
    default: return "CLWH_ERR_UNKNOWN";
  }
}

int clwh_last_hip_error(void) { return g_last_hip_error; }
const char *clwh_version(void) { return "clwhip 0.1 (gfx950)"; }

}  // extern "C"
