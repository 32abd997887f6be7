#include "reference_volume.hpp"

#include <algorithm>

#include "common.hpp"

reference_volume::reference_volume(clw_context &c, volume_block *b)
    : ctx(c),
      volume_size({b->m_voxel_count_x, b->m_voxel_count_y, b->m_voxel_count_z}),
      original_volume(ctx, std::vector<short>(b->m_voxels), volume_size, true) {
  // value range on the host (one pass over data that is in cache anyway); the reference gets it from
  // its fetch_stats kernel (app/reference_volume.cpp:22-40), which is not on the render path
  if (!b->m_voxels.empty()) {
    auto mm = std::minmax_element(b->m_voxels.begin(), b->m_voxels.end());
    value_range = {*mm.first, *mm.second};
  }
}

std::array<int, 2> reference_volume::get_value_range() const {
  return {std::max(value_clip[0], value_range[0]), std::min(value_clip[1], value_range[1])};
}

std::array<int, 2> reference_volume::get_gradient_range() const {
  return {std::max(gradient_clip[0], gradient_range[0]), std::min(gradient_clip[1], gradient_range[1])};
}

std::array<size_t, 3> reference_volume::get_volume_size_evenness(unsigned int l) const {
  return {evenness((unsigned int)volume_size[0], l), evenness((unsigned int)volume_size[1], l),
          evenness((unsigned int)volume_size[2], l)};
}
