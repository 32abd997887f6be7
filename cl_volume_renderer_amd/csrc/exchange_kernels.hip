// exchange_kernels.hip -- the reference's world-space accumulation (utility.cl:20-54) fed by contributions that were computed
// somewhere else: the pixels of other ranks, or of this rank in image-space scratch.
//
// The reference gives a voxel's 256 tokens to whichever work-items reach the atomic first (utility.cl:20-31, ray_marching.cl:28,39).
// Across GPUs "first" has no meaning, so the exchange fixes one legal outcome of that race: the contributions to a voxel are taken
// in the order the caller lists them -- (rank, pixel) -- while the voxel's count is below 256, and dropped afterwards.  Every rank
// applies the same list to its replica of the cache and ends with the same bytes; below the cap the result is the single-GPU cache
// (integer adds commute).
//
//   clwh_cache_exchange_plan          once per camera: group the listed cache entries by voxel, keeping the list order inside a
//                                     group (one stable radix sort of (entry, position) pairs, rocPRIM)
//   clwh_cache_apply_contributions    once per pass: ONE kernel, one lane per voxel group, walks the group's contributions in order
//                                     against the entry's count -- no atomics, no host round trip
#include <cstring>
#include <new>

#include <rocprim/device/device_radix_sort.hpp>

#include "clwh_internal.hpp"
#include "device_math.hpp"

struct clwh_exchange_plan {
  clwh_ctx *ctx = nullptr;
  uint64_t n = 0;
  int64_t *sorted_entries = nullptr;  // [n] ascending, equal entries in list order
  uint32_t *order = nullptr;          // [n] list position of each sorted element
};

namespace clvr {

__global__ __launch_bounds__(256) void k_iota(uint32_t *__restrict__ v, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

// one lane per sorted position; the lane at the first element of a voxel's group applies the whole group
__global__ __launch_bounds__(256) void k_apply_contributions(const int64_t *__restrict__ sorted_entries, const uint32_t *__restrict__ order,
                                                             uint64_t n, const int32_t *__restrict__ rgb, int rgb_stride,
                                                             uint32_t *__restrict__ cache, int64_t cache_entries) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const int64_t e = sorted_entries[i];
  if (i > 0 && sorted_entries[i - 1] == e) return;  // not the head of its group
  if (e < 0 || e >= cache_entries) return;           // a pixel whose hit lies outside the cache contributes nothing (as in k_bounce)
  uint2 w = *reinterpret_cast<const uint2 *>(cache + 2 * e);
  uint32_t r = w.x & 0xFFFFu, g = w.x >> 16, b = w.y & 0xFFFFu, count = w.y >> 16;
  for (uint64_t j = i; j < n && sorted_entries[j] == e && count < 256u; ++j) {
    const int32_t *c = rgb + (size_t)order[j] * (size_t)rgb_stride;
    // a granted token and its add (utility.cl:20-54); <= 256 contributions of <= 255 each: no lane carries into its neighbour
    r += (uint32_t)c[0] & 0xFFFFu;
    g += (uint32_t)c[1] & 0xFFFFu;
    b += (uint32_t)c[2] & 0xFFFFu;
    count += 1u;
  }
  w.x = (r & 0xFFFFu) | (g << 16);
  w.y = (b & 0xFFFFu) | (count << 16);
  *reinterpret_cast<uint2 *>(cache + 2 * e) = w;
}

__global__ __launch_bounds__(256) void k_debug_conversions(const float *__restrict__ in, uint64_t n, int32_t *__restrict__ oi, uint32_t *__restrict__ ou) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float v = in[i];
  oi[i] = f2i(v); oi[n + i] = f2i_reference(v);
  ou[i] = f2u(v); ou[n + i] = f2u_reference(v);
}

// one wave per 64 inputs: [0, n / 64) the DPP minimum (device_math.hpp), [n / 64, 2 n / 64) the shuffle loop
__global__ __launch_bounds__(64) void k_debug_wave_min(const uint32_t *__restrict__ in, uint64_t n_waves, uint32_t *__restrict__ out) {
  const uint32_t v = in[(uint64_t)blockIdx.x * 64u + threadIdx.x];
  const uint32_t a = wave_min_u32(v), b = wave_min_u32_reference(v);
  if (threadIdx.x == (blockIdx.x & 63u)) { out[blockIdx.x] = a; out[n_waves + blockIdx.x] = b; }  // (read from a different lane each time)
}

hipError_t sort_entry_pairs(void *temp, size_t &temp_bytes, const int64_t *keys_in, int64_t *keys_out, const uint32_t *vals_in,
                            uint32_t *vals_out, size_t n, unsigned end_bit, hipStream_t s) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, s);
}

}  // namespace clvr

using namespace clvr;

#define HIP_TRY_X(expr)                                                          \
  do {                                                                           \
    hipError_t _e = (expr);                                                      \
    if (_e != hipSuccess) return _e == hipErrorOutOfMemory ? CLWH_ERR_OUT_OF_MEMORY : CLWH_ERR_HIP; \
  } while (0)

extern "C" {

int clwh_debug_float_conversions(clwh_ctx *ctx, clwh_mem *floats_in, uint64_t n, clwh_mem *i32_out, clwh_mem *u32_out) {
  if (!ctx || !floats_in || !i32_out || !u32_out) return CLWH_ERR_INVALID_VALUE;
  if (floats_in->bytes < n * 4 || i32_out->bytes < n * 8 || u32_out->bytes < n * 8 || n >= (1ull << 31)) return CLWH_ERR_SIZE_MISMATCH;
  if (n == 0) return CLWH_OK;
  HIP_TRY_X(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_debug_conversions, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, ctx->stream, (const float *)floats_in->dptr, n,
                     (int32_t *)i32_out->dptr, (uint32_t *)u32_out->dptr);
  HIP_TRY_X(hipGetLastError());
  return CLWH_OK;
}

int clwh_debug_wave_min(clwh_ctx *ctx, clwh_mem *u32_in, uint64_t n, clwh_mem *u32_out) {
  if (!ctx || !u32_in || !u32_out) return CLWH_ERR_INVALID_VALUE;
  if ((n & 63u) != 0 || u32_in->bytes < n * 4 || u32_out->bytes < (n / 64) * 8 || n >= (1ull << 37)) return CLWH_ERR_SIZE_MISMATCH;
  if (n == 0) return CLWH_OK;
  HIP_TRY_X(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_debug_wave_min, dim3((unsigned)(n / 64)), dim3(64), 0, ctx->stream, (const uint32_t *)u32_in->dptr, n / 64, (uint32_t *)u32_out->dptr);
  HIP_TRY_X(hipGetLastError());
  return CLWH_OK;
}

int clwh_cache_exchange_plan_release(clwh_exchange_plan *plan) {
  if (!plan) return CLWH_ERR_INVALID_VALUE;
  (void)hipSetDevice(plan->ctx->device);
  (void)hipStreamSynchronize(plan->ctx->stream);
  if (plan->sorted_entries) (void)hipFree(plan->sorted_entries);
  if (plan->order) (void)hipFree(plan->order);
  delete plan;
  return CLWH_OK;
}

int clwh_cache_exchange_plan(clwh_ctx *ctx, clwh_mem *entries, uint64_t n, clwh_exchange_plan **out) {
  if (!ctx || !entries || !out) return CLWH_ERR_INVALID_VALUE;
  *out = nullptr;
  if (n >= (1ull << 32) || entries->bytes < n * sizeof(int64_t)) return CLWH_ERR_SIZE_MISMATCH;
  HIP_TRY_X(hipSetDevice(ctx->device));
  clwh_exchange_plan *p = new (std::nothrow) clwh_exchange_plan();
  if (!p) return CLWH_ERR_OUT_OF_MEMORY;
  p->ctx = ctx;
  p->n = n;
  *out = p;
  if (n == 0) return CLWH_OK;
  uint32_t *iota = nullptr;
  void *temp = nullptr;
  size_t temp_bytes = 0;
  int rc = CLWH_OK;
  auto fail = [&](int code) {
    if (iota) (void)hipFree(iota);
    if (temp) (void)hipFree(temp);
    (void)clwh_cache_exchange_plan_release(p);
    *out = nullptr;
    return code;
  };
  if (hipMalloc((void **)&p->sorted_entries, n * sizeof(int64_t)) != hipSuccess || hipMalloc((void **)&p->order, n * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc((void **)&iota, n * sizeof(uint32_t)) != hipSuccess)
    return fail(CLWH_ERR_OUT_OF_MEMORY);
  hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, ctx->stream, iota, n);
  const int64_t *keys_in = (const int64_t *)entries->dptr;
  // radix sorts are stable: equal entries keep the caller's (rank, pixel) order
  if (rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, p->sorted_entries, iota, p->order, (size_t)n, 0u, 64u, ctx->stream) != hipSuccess)
    return fail(CLWH_ERR_HIP);
  if (hipMalloc(&temp, temp_bytes ? temp_bytes : 16) != hipSuccess) return fail(CLWH_ERR_OUT_OF_MEMORY);
  if (rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, p->sorted_entries, iota, p->order, (size_t)n, 0u, 64u, ctx->stream) != hipSuccess)
    return fail(CLWH_ERR_HIP);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(CLWH_ERR_HIP);  // once per camera; the scratch goes here
  (void)hipFree(iota);
  (void)hipFree(temp);
  return rc;
}

int clwh_cache_apply_contributions(clwh_ctx *ctx, clwh_exchange_plan *plan, clwh_mem *buffer_volume, clwh_mem *rgb, int32_t rgb_stride) {
  if (!ctx || !plan || !buffer_volume || !rgb || plan->ctx != ctx) return CLWH_ERR_INVALID_VALUE;
  if (rgb_stride < 3) return CLWH_ERR_INVALID_VALUE;
  if (rgb->bytes < plan->n * (uint64_t)rgb_stride * sizeof(int32_t) || buffer_volume->bytes < 8) return CLWH_ERR_SIZE_MISMATCH;
  if (plan->n == 0) return CLWH_OK;
  HIP_TRY_X(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_apply_contributions, dim3((unsigned)((plan->n + 255u) / 256u)), dim3(256), 0, ctx->stream, plan->sorted_entries,
                     plan->order, plan->n, (const int32_t *)rgb->dptr, (int)rgb_stride, (uint32_t *)buffer_volume->dptr,
                     (int64_t)(buffer_volume->bytes / 8));
  HIP_TRY_X(hipGetLastError());
  clwh_touch(buffer_volume);
  return CLWH_OK;
}

}  // extern "C"
