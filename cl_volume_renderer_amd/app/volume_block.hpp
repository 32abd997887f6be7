// volume_block.hpp -- flat host container of a 16-bit volume, x fastest (reference
// app/volume_block.hpp:6-34).  NOT bricked: bricking happens inside the shim at render time.
#pragma once
#include <utility>
#include <vector>

struct volume_block {
  std::vector<short> m_voxels;
  const unsigned int m_voxel_count_x, m_voxel_count_y, m_voxel_count_z;
  const float m_voxel_size_x, m_voxel_size_y, m_voxel_size_z;  // parsed but unused by rendering

  volume_block(unsigned int nx, unsigned int ny, unsigned int nz, float sx, float sy, float sz)
      : m_voxel_count_x(nx), m_voxel_count_y(ny), m_voxel_count_z(nz), m_voxel_size_x(sx), m_voxel_size_y(sy),
        m_voxel_size_z(sz) {}
  volume_block(std::vector<short> &&voxels, unsigned int nx, unsigned int ny, unsigned int nz, float sx, float sy,
               float sz)
      : m_voxels(std::move(voxels)), m_voxel_count_x(nx), m_voxel_count_y(ny), m_voxel_count_z(nz),
        m_voxel_size_x(sx), m_voxel_size_y(sy), m_voxel_size_z(sz) {}
};
