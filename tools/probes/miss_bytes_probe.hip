// miss_bytes_probe.hip -- how many bytes does one L2-missing 1-byte load move on gfx950, and what do the TCC counters
// report for it?  (MI355X_MICROARCH.md calibrates FETCH_SIZE for wide coalesced streams only: "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern".)
//
// An 8 GiB buffer (32x the Infinity Cache) is read four ways; every kernel reads each touched line exactly once:
//   stream   : 16 bytes per lane, fully coalesced                          -> X bytes must move
//   stride S : one byte every S bytes (S = 32, 64, 128, 256), lanes adjacent in the stride sequence
//              -> if the fetch granule is G bytes, max(X, X * G / S) ... i.e. X bytes for S <= G, X * G / S above
//   random   : independent (not chained) random 1-byte loads, L of them     -> L * G bytes
// Timing alone gives G: the stride kernels are bandwidth-bound (64 lines per wave-instruction), so
// time(stride S) / time(stream) = min(1, G / S).  Run under rocprofv3 --pmc (tools/profile_miss_bytes.sh) the same
// launches give FETCH_SIZE / TCC_EA0_RDREQ(_32B) / TCC_MISS per known number of missing loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at line %d\n", (int)e, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_stream(const uint4 *__restrict__ t, size_t n16, unsigned *out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = t[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int STRIDE>
__global__ __launch_bounds__(256) void k_stride(const unsigned char *__restrict__ t, size_t n, unsigned *out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += t[i * STRIDE];
  if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_random(const unsigned char *__restrict__ t, size_t mask, int loads, unsigned *out) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long x = (unsigned long long)gid * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  unsigned acc = 0;
  for (int s = 0; s < loads; ++s) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;  // independent of the loaded data: many loads in flight per lane
    acc += t[x & mask];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// the same independent random loads with other cache policies: does any of them fetch less than a 128-byte line?
template <int POLICY>
__global__ __launch_bounds__(256) void k_random_policy(const unsigned char *__restrict__ t, size_t mask, int loads, unsigned *out) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long x = (unsigned long long)gid * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  unsigned acc = 0;
#define CLVR_PROBE_LOAD(dst)                                                                                       \
  do {                                                                                                             \
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;                                                                       \
    const unsigned char *p = t + (x & mask);                                                                       \
    if (POLICY == 0) asm volatile("global_load_ubyte %0, %1, off nt" : "=&v"(dst) : "v"(p) : "memory");            \
    if (POLICY == 1) asm volatile("global_load_ubyte %0, %1, off sc0 sc1" : "=&v"(dst) : "v"(p) : "memory");       \
    if (POLICY == 2) asm volatile("global_load_ubyte %0, %1, off sc1" : "=&v"(dst) : "v"(p) : "memory");           \
    if (POLICY == 3) asm volatile("global_load_ubyte %0, %1, off sc0 sc1 nt" : "=&v"(dst) : "v"(p) : "memory");    \
    if (POLICY == 4) asm volatile("global_load_ubyte %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");               \
  } while (0)
  for (int s = 0; s < loads; s += 8) {  // eight loads in flight per lane, then one wait
    unsigned v0, v1, v2, v3, v4, v5, v6, v7;
    CLVR_PROBE_LOAD(v0); CLVR_PROBE_LOAD(v1); CLVR_PROBE_LOAD(v2); CLVR_PROBE_LOAD(v3);
    CLVR_PROBE_LOAD(v4); CLVR_PROBE_LOAD(v5); CLVR_PROBE_LOAD(v6); CLVR_PROBE_LOAD(v7);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : : "memory");
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
#undef CLVR_PROBE_LOAD
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t X = 8ull << 30;
  unsigned char *t; unsigned *out;
  CK(hipMalloc(&t, X)); CK(hipMemset(t, 1, X)); CK(hipMalloc(&out, 64));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 256 * 16;
  auto time_it = [&](const char *name, double loads, auto launch) {
    launch();  // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-12s %8.3f ms  %10.1f M loads  %8.2f G loads/s  buffer bytes / time = %7.2f TB/s\n", name, ms, loads / 1e6, loads / ms / 1e6, (double)X / ms / 1e9);
    return ms;
  };
  const float ts = time_it("stream", (double)X / 16, [&] { k_stream<<<blocks, 256>>>((const uint4 *)t, X / 16, out); });
  const float t32 = time_it("stride 32", (double)X / 32, [&] { k_stride<32><<<blocks, 256>>>(t, X / 32, out); });
  const float t64 = time_it("stride 64", (double)X / 64, [&] { k_stride<64><<<blocks, 256>>>(t, X / 64, out); });
  const float t128 = time_it("stride 128", (double)X / 128, [&] { k_stride<128><<<blocks, 256>>>(t, X / 128, out); });
  const float t256 = time_it("stride 256", (double)X / 256, [&] { k_stride<256><<<blocks, 256>>>(t, X / 256, out); });
  const float t512 = time_it("stride 512", (double)X / 512, [&] { k_stride<512><<<blocks, 256>>>(t, X / 512, out); });
  const int loads = 64;
  const double L = (double)blocks * 256 * loads;
  const float tr = time_it("random", L, [&] { k_random<<<blocks, 256>>>(t, X - 1, loads, out); });
  // (inline-asm loads, eight in flight per lane: these five compare with EACH OTHER first)
  const int loads2 = 64;
  const double L2 = (double)blocks * 256 * loads2;
  time_it("rnd plain*", L2, [&] { k_random_policy<4><<<blocks, 256>>>(t, X - 1, loads2, out); });
  time_it("rnd nt", L2, [&] { k_random_policy<0><<<blocks, 256>>>(t, X - 1, loads2, out); });
  time_it("rnd sc0sc1", L2, [&] { k_random_policy<1><<<blocks, 256>>>(t, X - 1, loads2, out); });
  time_it("rnd sc1", L2, [&] { k_random_policy<2><<<blocks, 256>>>(t, X - 1, loads2, out); });
  time_it("rnd sc01nt", L2, [&] { k_random_policy<3><<<blocks, 256>>>(t, X - 1, loads2, out); });
  printf("time relative to the stream: stride32 %.2f  stride64 %.2f  stride128 %.2f  stride256 %.2f  stride512 %.2f\n", t32 / ts, t64 / ts, t128 / ts, t256 / ts, t512 / ts);
  printf("random: %.1f M independent 1-byte loads over 8 GiB in %.3f ms = %.1f G loads/s; at 64 B each %.2f TB/s, at 128 B each %.2f TB/s\n",
         L / 1e6, tr, L / tr / 1e6, L * 64 / tr / 1e9, L * 128 / tr / 1e9);
  return 0;
}
