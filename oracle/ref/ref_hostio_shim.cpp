// oracle/ref/ref_hostio_shim.cpp -- C entry points over the REFERENCE's own host-side loaders, compiled from the
// reference sources where they lie (never copied): TEST INFRASTRUCTURE ONLY.
//
// oracle/ref/Makefile builds  oracle/_ref/libref_hostio.so  from
//     /root/reference/app/hdre_loader.cpp   (+ subprojects/stb/stb_image.h, which that file instantiates)
//     /root/reference/app/image.cpp
//     /root/reference/app/nrrd_loader.cpp   (+ zlib)
//     /root/reference/app/volume_block.cpp
//     /root/reference/app/common.hpp        (Position3D, header only)
// and this shim, which only calls the reference's public classes.  The library pins this project's mirrors
// (cl_volume_renderer_amd/app/{hdre_loader,png_reader,nrrd_loader}.*, scene.camera_direction / orc_camera_direction)
// against the real reference code in tests/test_ref_hostio.py.  The rest of the reference (renderer, SDF, kernels)
// needs an OpenCL runtime and ImGui/SDL that the image lacks and is not buildable here (Oracle).
#include <cstring>
#include <string>

#include "common.hpp"
#include "hdre_loader.hpp"
#include "nrrd_loader.hpp"

extern "C" {

// app/hdre_loader.cpp:7-24 -> RGBA8; returns the number of bytes, or -needed when `cap` is too small
long long ref_env_load(const char *path, unsigned dims[2], unsigned char *out, long long cap) {
  hdre_loader loader;
  image im = loader.load_file(path);
  dims[0] = im.m_width;
  dims[1] = im.m_height;
  const long long n = (long long)im.m_pixels.size();
  if (n > cap) return -n;
  std::memcpy(out, im.m_pixels.data(), (size_t)n);
  return n;
}

// app/nrrd_loader.cpp load_file -> voxels (x fastest), counts and voxel sizes
long long ref_nrrd_load(const char *path, unsigned counts[3], float sizes[3], short *out, long long cap_voxels) {
  nrrd_loader loader;
  volume_block v = loader.load_file(path);
  counts[0] = v.m_voxel_count_x; counts[1] = v.m_voxel_count_y; counts[2] = v.m_voxel_count_z;
  sizes[0] = v.m_voxel_size_x; sizes[1] = v.m_voxel_size_y; sizes[2] = v.m_voxel_size_z;
  const long long n = (long long)v.m_voxels.size();
  if (n > cap_voxels) return -n;
  std::memcpy(out, v.m_voxels.data(), (size_t)n * sizeof(short));
  return n;
}

// app/renderer.cpp:140: Position3D(direction_look[0], direction_look[1], 0.0, {1, 0, 0})
void ref_camera_direction(double alpha, double beta, float out[3]) {
  Position3D v(alpha, beta, 0.0, {1.0, 0.0, 0.0});
  out[0] = v.val[0]; out[1] = v.val[1]; out[2] = v.val[2];
}

}  // extern "C"
