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
__device__ __forceinline__ bool event_at(const VolumeIntLinear &v, const TfDev &tf, int x, int y, int z) {
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
  const bool e = event_at<USE_GRAD>(v, a.tf, x, y, z);
  bool homogenous = true;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int nx = min(max(x + ((c & 1) ? 1 : -1), 0), a.X - 1);
    const int ny = min(max(y + ((c & 2) ? 1 : -1), 0), a.Y - 1);
    const int nz = min(max(z + ((c & 4) ? 1 : -1), 0), a.Z - 1);
    homogenous &= (event_at<USE_GRAD>(v, a.tf, nx, ny, nz) == e);
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

hipError_t launch_sdf_layer(const SdfArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_sdf_layer, sdf_grid(a), dim3(sdf_block(a)), 0, s, a);
  return hipGetLastError();
}

}  // namespace clvr
