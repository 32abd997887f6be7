// sdf_benchmark_main.cpp -- headless SDF-build timing in the manner of the reference's
// app/sdf_benchmark.cpp:15-20 (N builds of the same volume with the test TF, wall time per build).
// Usage: sdf_benchmark <nrrd> [iterations]
#include <chrono>
#include <cstdlib>
#include <iostream>

#include "nrrd_loader.hpp"
#include "signed_distance_field.hpp"

int main(int argc, char const *argv[]) {
  if (argc < 2) {
    std::cerr << "usage: " << argv[0] << " <volume.nrrd> [iterations]\n";
    return 2;
  }
  const int iterations = argc > 2 ? std::atoi(argv[2]) : 100;
  clw_context ctx;
  nrrd_loader loader;
  volume_block b = loader.load_file(argv[1]);
  reference_volume rv(ctx, &b);
  for (int i = 0; i < iterations; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    signed_distance_field sdf(ctx, rv, "inline bool is_event_gen(short value, short gradient, uint4 *color){ return (value > 800); }");
    ctx.finish();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "[CUSTOM TIMER] Loop Iteration" << s << "s\n";
  }
  return 0;
}
