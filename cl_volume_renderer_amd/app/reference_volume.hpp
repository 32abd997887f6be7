// reference_volume.hpp -- owner of the volume image, its statistics and the size helpers every launch
// uses (mirror of the reference's app/reference_volume.hpp / .cpp).
// On the hot path: constructor (upload), get_volume_size*, get_volume_length, get_reference_volume.
// Next to it: fetch_stats at construction, set_clipping (apply_clip), filter (bilateral_filter).
#pragma once

#include <array>
#include <limits>

#include <clw_context.hpp>
#include <clw_image.hpp>

#include "volume_block.hpp"

struct Volume_Stats {
  float min_v = 0, max_v = 0, min_g = 0, max_g = 0;
  Volume_Stats() = default;
  template <typename T>
  Volume_Stats(const T &value_stats, const T &gradient_stats)
      : min_v(value_stats[0]), max_v(value_stats[1]), min_g(gradient_stats[0]), max_g(gradient_stats[1]) {}
};

class reference_volume {
 public:
  reference_volume(clw_context &c, volume_block *b);
  void set_value_clip(std::array<int, 2> clip) { value_clip = clip; }
  void set_gradient_clip(std::array<int, 2> clip) { gradient_clip = clip; }
  /// crop the volume to [min, max) -- rendering, SDF and cache then use the cropped copy
  void set_clipping(std::array<size_t, 3> min, std::array<size_t, 3> max);
  /// 5x5x5 bilateral filter of the current (cropped or original) volume; the filtered image replaces it on the device
  void filter();
  std::array<int, 2> get_value_range() const;
  std::array<int, 2> get_gradient_range() const;
  const std::array<size_t, 3> &get_original_volume_size() const { return volume_size; }
  const std::array<size_t, 3> &get_volume_size() const { return cropped_volume_size; }
  std::array<size_t, 3> get_volume_size_evenness(unsigned int l) const;
  size_t get_volume_length() const { return cropped_volume_size[0] * cropped_volume_size[1] * cropped_volume_size[2]; }
  const clw_image<short> &get_reference_volume() const { return is_cropped() ? cropped_volume : original_volume; }
  Volume_Stats get_volume_stats() const { return Volume_Stats(get_value_range(), get_gradient_range()); }

 private:
  bool is_cropped() const { return cropped_volume.size() > 8; }  // the placeholder image has 8 voxels
  clw_context &ctx;
  std::array<size_t, 3> volume_size;
  std::array<size_t, 3> cropped_volume_size;
  clw_image<short> original_volume;
  clw_image<short> cropped_volume;
  std::array<int, 2> value_range{0, 0};     // over the original volume
  std::array<int, 2> gradient_range{0, 0};  // over the original volume
  std::array<int, 2> value_clip = {std::numeric_limits<int>::min(), std::numeric_limits<int>::max()};
  std::array<int, 2> gradient_clip = {std::numeric_limits<int>::min(), std::numeric_limits<int>::max()};
};
