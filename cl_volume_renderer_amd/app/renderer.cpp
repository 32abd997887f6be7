#include "renderer.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <set>

#include <clwh.h>

#include "common_defines.hpp"

// Mirrors the sequencing of the reference's app/renderer.cpp; what differs is below the clw_* API:
// no JIT on flush, the SDF build is one call, and the frame the caller receives is resolved after the
// pass instead of being read from the cache while other work-items still add to it.

renderer::renderer(clw_context &c)
    : ctx(c),
      render_func(ctx, "empty.cl", "empty"),
      frame(ctx, std::vector<unsigned char>((size_t)SCREEN_WIDTH * SCREEN_HEIGHT * 4), {SCREEN_WIDTH, SCREEN_HEIGHT, 1}),
      buffer_volume(ctx, std::vector<unsigned short>(8 * 4)),
      tfframe(ctx, std::vector<unsigned char>(2 * 2 * 4), {2, 2, 1}),
      sdf(ctx) {}

void renderer::image_set(const reference_volume *rv, const env_map *map) {
  volume = rv;
  emap = map;
}

void renderer::next_event_code_set(const std::string cl_code) { local_cl_code = cl_code; }

// reference :25-43 -- (re)allocate + zero the voxel cache, rebuild the render function and the SDF
void renderer::flush_changes() {
  const auto dims = volume->get_volume_size();
  // the reference allocates X*Y*Z*4 ushorts and can index one row past it (opencl_kernels/utility.cl:21
  // with utility_ray.cl:112-117); the padded length keeps that access inside the buffer
  const size_t cache_len = (size_t)clwh_cache_len((uint32_t)dims[0], (uint32_t)dims[1], (uint32_t)dims[2]);
  if (buffer_volume.size() != cache_len)
    buffer_volume = clw_vector<unsigned short>(ctx, std::vector<unsigned short>(cache_len), false);

  auto buffer_reset = clw_function(ctx, "buffer_reset.cl", "buffer_reset");
  buffer_reset.execute(volume->get_volume_size_evenness(4), {4, 4, 4}, volume->get_reference_volume(), buffer_volume);

  render_func = clw_function(ctx, "ray_marching.cl", "render", local_cl_code);
  sdf = signed_distance_field(ctx, *volume, local_cl_code);
}

// reference :131-158 -- one sample per pixel with a fresh std::rand() seed, then a blocking readback
void *renderer::render_frame(struct ui_state &state, bool &frame_changed) {
  frame_changed = false;
  if (!state.cam_changed && !state.path_changed) return &frame[0];

  Position3D vec(state.direction_look[0], state.direction_look[1], 0.0, {1.0, 0.0, 0.0});
  const int random_seed = std::rand();

  render_func.execute({(size_t)state.width, (size_t)state.height, 1}, {8, 8, 1}, frame, volume->get_reference_volume(),
                      sdf.get_sdf_buffer(), emap->get_buffer(), buffer_volume, state.position.val[0],
                      state.position.val[1], state.position.val[2], vec.val[0], vec.val[1], vec.val[2], random_seed);
  frame.pull();

  state.cam_changed = false;
  state.path_changed = false;
  frame_changed = true;
  return &frame[0];
}

void renderer::render_frame_device(struct ui_state &state, const clw_foreign_memory &target, int passes) {
  if (!state.cam_changed && !state.path_changed) return;
  Position3D vec(state.direction_look[0], state.direction_look[1], 0.0, {1.0, 0.0, 0.0});
  clwh_render_desc d{};
  d.frame = target.get_device_reference();
  d.volume = volume->get_reference_volume().get_device_reference();
  d.sdf = sdf.get_sdf_buffer().get_device_reference();
  d.env = emap->get_buffer().get_device_reference();
  d.buffer_volume = buffer_volume.get_device_reference();
  for (int q = 0; q < 3; ++q) {
    d.cam_pos[q] = (float)state.position.val[q];
    d.cam_dir[q] = (float)vec.val[q];
  }
  d.width = (uint32_t)state.width;
  d.height = (uint32_t)state.height;
  d.accum_mode = CLWH_ACCUM_VOXEL_CACHE;
  d.tile_world = 1;
  d.write_frame = 1;
  passes = std::max(1, std::min(passes, CLWH_MAX_SEEDS));
  d.n_seeds = passes;
  for (int k = 0; k < passes; ++k) d.seeds[k] = std::rand();  // the seeds `passes` render_frame calls would have drawn
  target.acquire();
  clw_fail_hard_on_error(clwh_render(render_func.get_kernel(), &d));
  target.release();
  state.cam_changed = false;
  state.path_changed = false;
}

// reference :45-124 -- the 2-D (value, |gradient|) histogram texture of the transfer-function editor:
// bin the volume, quantise the counts on the host so that small counts stay distinguishable, rank the
// distinct counts, colour each bin by its rank.  (The reference declares render_tf(width, height) and
// defines render_tf(height, width); both call sites pass 500 x 500.)
void *renderer::render_tf(const unsigned int height, const unsigned int width) {
  tfframe = clw_image<unsigned char, 4>(ctx, std::vector<unsigned char>((size_t)height * width * 4), {width, height, 1});

  clw_vector<unsigned int> bins(ctx, std::vector<unsigned int>((size_t)width * height, 0));
  bins.push();
  const Volume_Stats stats = volume->get_volume_stats();
  clw_function sort_values(ctx, "histogram.cl", "tf_sort_values");
  sort_values.execute(volume->get_volume_size_evenness(8), {4, 4, 4}, volume->get_reference_volume(), bins, width, height,
                      stats.min_v, stats.max_v, stats.min_g, stats.max_g);
  bins.pull();

  // round every count down to its two leading decimal digits and collect the distinct results
  std::set<int> distinct;
  for (size_t i = 0; i < bins.size(); ++i) {
    const int value = (int)bins[i];
    if (value == 0) continue;
    const int unit = std::max((int)std::pow(10, std::floor(std::log10(value)) - 1), 1);
    const int corrected = (int)(std::floor(value / unit) * unit);
    bins[i] = (unsigned int)corrected;
    distinct.insert(corrected);
  }
  bins.push();

  if (distinct.empty()) {
    std::cout << "Warning, histogram does not contain non-zero entries.\n";
  } else {
    clw_vector<int> ranks(ctx, std::vector<int>(distinct.begin(), distinct.end()));
    ranks.push();
    clw_function flush_colors(ctx, "histogram.cl", "tf_flush_color_frame");
    flush_colors.execute({evenness((unsigned)tfframe.get_dimensions()[0], 16), evenness((unsigned)tfframe.get_dimensions()[1], 16), 1},
                         {16, 16, 1}, tfframe, bins, ranks, (int)ranks.size());
  }
  tfframe.pull();
  return &tfframe[0];
}
