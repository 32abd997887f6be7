// sdf_test_main.cpp -- the reference's only test (tests/sdf/sdf_test.cpp:6-33) as a stand-alone
// program over this project's headers: load the NRRD test block, build the SDF with the TF
// `value > 800`, compare all 50 540 values with the golden list.  Usage: sdf_test <nrrd> <values.x>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "nrrd_loader.hpp"
#include "signed_distance_field.hpp"

int main(int argc, char const *argv[]) {
  if (argc != 3) {
    std::cerr << "usage: " << argv[0] << " <testdata.nrrd> <values.x>\n";
    return 2;
  }
  std::vector<int> expected;
  {
    std::ifstream in(argv[2]);
    std::string tok;
    while (std::getline(in, tok, ',')) {
      const auto b = tok.find_first_of("-0123456789");
      if (b != std::string::npos) expected.push_back(std::atoi(tok.c_str() + b));
    }
  }
  clw_context ctx;
  nrrd_loader loader;
  volume_block b = loader.load_file(argv[1]);
  reference_volume rv(ctx, &b);
  signed_distance_field sdf(ctx, rv, "inline bool is_event_gen(short value, short gradient, uint4 *color){ return (value > 800); }");
  auto &image = sdf.get_sdf_buffer();
  if (image.size() != expected.size()) {
    std::cerr << "size mismatch: " << image.size() << " vs " << expected.size() << "\n";
    return 1;
  }
  image.pull();
  size_t bad = 0;
  for (size_t i = 0; i < expected.size(); ++i) bad += (expected[i] != image[i]);
  if (bad) {
    std::cerr << bad << " of " << expected.size() << " values differ\n";
    return 1;
  }
  std::cout << "EVERYTHING FINE (" << expected.size() << " values, " << sdf.layers() << " layers)\n";
  return 0;
}
