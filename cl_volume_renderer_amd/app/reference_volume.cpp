#include "reference_volume.hpp"

#include <algorithm>
#include <cassert>

#include <clw_function.hpp>
#include <clw_vector.hpp>

#include "common.hpp"

// reference app/reference_volume.cpp:11-44: upload, then one fetch_stats launch for the value and
// gradient ranges the transfer-function editor works with
reference_volume::reference_volume(clw_context &c, volume_block *b)
    : ctx(c),
      volume_size({b->m_voxel_count_x, b->m_voxel_count_y, b->m_voxel_count_z}),
      cropped_volume_size(volume_size),
      original_volume(ctx, std::move(b->m_voxels), volume_size, true),
      cropped_volume(ctx, std::vector<short>(8), {2, 2, 2}, false) {
  const int lowest = std::numeric_limits<int>::min(), highest = std::numeric_limits<int>::max();
  clw_vector<int> stats(ctx, std::vector<int>{highest, lowest, highest, lowest, lowest}, true);
  clw_function fetch_stats(ctx, "reference_volume_figures.cl", "fetch_stats");
  fetch_stats.execute(get_volume_size_evenness(8), {4, 4, 4}, original_volume, stats);
  stats.pull();
  value_range = {stats[0], stats[1]};
  gradient_range = {stats[2], stats[3]};
}

// reference :54-68
void reference_volume::set_clipping(std::array<size_t, 3> min, std::array<size_t, 3> max) {
  assert(min[0] < max[0] && min[1] < max[1] && min[2] < max[2]);
  cropped_volume_size = {max[0] - min[0], max[1] - min[1], max[2] - min[2]};
  clw_vector<unsigned int> start(ctx, std::vector<unsigned int>{(unsigned)min[0], (unsigned)min[1], (unsigned)min[2]}, true);
  clw_vector<unsigned int> length(ctx, std::vector<unsigned int>{(unsigned)cropped_volume_size[0], (unsigned)cropped_volume_size[1],
                                                                (unsigned)cropped_volume_size[2], 4u}, true);
  cropped_volume = clw_image<short>(ctx, std::vector<short>(get_volume_length()), get_volume_size());
  clw_function copy(ctx, "reference_volume_clip.cl", "apply_clip");
  copy.execute(get_volume_size_evenness(4), {4, 4, 4}, original_volume, cropped_volume, start, length);
}

// reference :70-80.  The move assignment at :77 does take effect: the device image of the volume becomes the
// filtered one, while its host copy is the zero-filled staging vector of `buffer` (never pulled) -- mirrored as is.
void reference_volume::filter() {
  clw_image<short> &ref = is_cropped() ? cropped_volume : original_volume;
  clw_image<short> buffer(ctx, std::vector<short>(get_volume_length(), 0), get_volume_size(), false);
  clw_function bilateral_filter(ctx, "volume_filter.cl", "bilateral_filter");
  bilateral_filter.execute(get_volume_size_evenness(8), {4, 4, 4}, ref, buffer);
  ref = std::move(buffer);
}

std::array<int, 2> reference_volume::get_value_range() const {
  return {std::max(value_clip[0], value_range[0]), std::min(value_clip[1], value_range[1])};
}

std::array<int, 2> reference_volume::get_gradient_range() const {
  return {std::max(gradient_clip[0], gradient_range[0]), std::min(gradient_clip[1], gradient_range[1])};
}

std::array<size_t, 3> reference_volume::get_volume_size_evenness(unsigned int l) const {
  return {evenness((unsigned int)cropped_volume_size[0], l), evenness((unsigned int)cropped_volume_size[1], l),
          evenness((unsigned int)cropped_volume_size[2], l)};
}
