// nrrd_loader.hpp -- NRRD0004 reader for 16-bit volumes: the step before the hot path
// (mirror of the reference's app/nrrd_loader.hpp).  Same public surface
// (`nrrd_loader::load_file(path) -> volume_block`) and the same accepted subset -- type short,
// dimension 3, little endian, encoding raw or gzip -- with 64-bit sizes so that 2048^3 (16 GiB)
// loads, and a streaming inflate instead of one `int`-sized call.
#pragma once

#include <cstdint>
#include <string>

#include "volume_block.hpp"

struct nrrd_header {
  unsigned int x = 0, y = 0, z = 0;
  float x_voxel_size = 1.f, y_voxel_size = 1.f, z_voxel_size = 1.f;  // relative to x
  std::uint64_t data_start = 0, data_end = 0;
  bool raw = true;
};

class nrrd_loader {
 public:
  /// read the header and the payload of `path`; any violation of the accepted subset is fatal (exit 1)
  volume_block load_file(const std::string path);
  nrrd_header load_header(const std::string &path);
};
