// gather_levels_probe.hip -- the rate of divergent 1-byte gathers by the level that serves them: every lane loads independent random
// bytes (eight loads in flight per lane) from a table of T bytes -- 16 KiB (L1), 1 MiB (one XCD's L2), 32 MiB (Infinity Cache), 8 GiB (HBM).
// Prints G loads/s and loads per clock and CU (2.4 GHz, 256 CUs): the address unit / L1 of a CU works a wave's gather off lane by lane, so
// this is the ceiling a kernel of byte gathers meets before it meets the line rate of the memory behind (tools/probes/miss_bytes_probe.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at line %d\n", (int)e, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ACTIVE>
__global__ __launch_bounds__(256) void k_gather(const unsigned char *__restrict__ t, size_t mask, int loads, unsigned *out) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  if ((int)(threadIdx.x & 63u) >= ACTIVE) return;  // ACTIVE of a wave's 64 lanes take part
  unsigned long long x = (unsigned long long)gid * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  unsigned acc = 0;
  for (int s = 0; s < loads; s += 8) {
    unsigned v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      v[k] = t[x & mask];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t X = 8ull << 30;
  unsigned char *t; unsigned *out;
  CK(hipMalloc(&t, X)); CK(hipMemset(t, 1, X)); CK(hipMalloc(&out, 64));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 256 * 16, loads = 512;
  const size_t sizes[] = {16ull << 10, 256ull << 10, 1ull << 20, 2ull << 20, 32ull << 20, 128ull << 20, 8ull << 30};
  for (size_t T : sizes) {
    for (int active : {64, 28}) {
      auto launch = [&] {
        if (active == 64) k_gather<64><<<blocks, 256>>>(t, T - 1, loads, out);
        else k_gather<28><<<blocks, 256>>>(t, T - 1, loads, out);
      };
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double L = (double)blocks * 4 * active * loads;
      printf("table %9zu KiB  %2d lanes per wave: %8.3f ms  %8.1f G loads/s  %.3f loads per clock and CU  %.1f wave instructions per us and CU\n", T >> 10, active, ms,
             L / ms / 1e6, L / ms / 1e6 / (256 * 2.4), (double)blocks * 4 * loads / ms / 1e3 / 256);
    }
  }
  return 0;
}
