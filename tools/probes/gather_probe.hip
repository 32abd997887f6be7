// gather_probe.hip -- what does gfx950 sustain for DEPENDENT random 1-byte gathers (the march's access
// pattern)?  Every lane chases its own chain: idx = hash(idx, table[idx]) over a table of T bytes,
// optionally confined to a window of W bytes per wave (page / TLB locality).  Prints G loads/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void chase(const unsigned char *__restrict__ t, size_t mask, size_t window_mask,
                                            int steps, unsigned *out) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned wave = gid >> 6;
  size_t base = ((size_t)wave * 0x9E3779B97F4A7C15ull) & mask & ~window_mask;  // the wave's window
  unsigned x = gid * 2654435761u + 12345u;
  unsigned acc = 0;
  for (int s = 0; s < steps; ++s) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const size_t idx = base + ((size_t)x & window_mask);
    const unsigned v = t[idx & mask];
    acc += v;
    x += v;  // dependent chain
  }
  out[gid] = acc;
}

int main() {
  const size_t sizes[] = {4ull << 20, 32ull << 20, 128ull << 20, 512ull << 20, 2048ull << 20};
  const size_t windows[] = {64ull << 10, 2ull << 20, 32ull << 20, ~0ull};
  const int waves_per_cu[] = {8, 16, 24, 32};
  unsigned char *t; unsigned *out;
  CK(hipMalloc(&t, 2048ull << 20)); CK(hipMemset(t, 1, 2048ull << 20));
  CK(hipMalloc(&out, 256 * 32 * 64 * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int steps = 2000;
  for (size_t T : sizes) for (size_t W : windows) {
    if (W != ~0ull && W > T) continue;
    for (int w : waves_per_cu) {
      const int blocks = 256 * w / 4;
      const size_t wm = (W == ~0ull) ? (T - 1) : (W - 1);
      chase<<<blocks, 256>>>(t, T - 1, wm, 100, out);
      CK(hipEventRecord(a));
      chase<<<blocks, 256>>>(t, T - 1, wm, steps, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double loads = (double)blocks * 256 * steps;
      printf("table %5zu MiB window %8s waves/CU %2d : %7.2f G loads/s  (%.3f ms)\n", T >> 20,
             W == ~0ull ? "all" : (W >= (1ull << 20) ? (std::to_string(W >> 20) + "MiB").c_str() : (std::to_string(W >> 10) + "KiB").c_str()),
             w, loads / ms / 1e6, ms);
    }
  }
  return 0;
}
