// renderer.hpp -- the renderer the SDL/ImGui application talks to: same class name, base class and
// public methods as the reference's app/renderer.hpp:10-29, so `ui::run(&renderer)` is a drop-in.
#pragma once
#include <string>

#include <clw_context.hpp>
#include <clw_foreign_memory.hpp>
#include <clw_function.hpp>
#include <clw_image.hpp>
#include <clw_vector.hpp>

#include "signed_distance_field.hpp"
#include "ui.hpp"

class renderer : public frame_emitter {
 public:
  explicit renderer(clw_context &ctx);
  void image_set(const reference_volume *rv, const env_map *map) override;
  void flush_changes() override;
  void *render_frame(struct ui_state &state, bool &frame_changed) override;
  void *render_tf(const unsigned int width, const unsigned int height) override;
  void next_event_code_set(const std::string cl_code) override;

  // not in the reference: the display hand-off without the host readback.  Same pass as
  // render_frame, but the frame is written into the display's own device memory (`target`: a mapped GL buffer,
  // see clw_foreign_memory.hpp) and ordered before the display stream's next work by an event -- no pull(), no
  // 8 MiB over PCIe per frame, no host synchronisation.  `passes` > 1 batches that many std::rand() seeds into
  // ONE launch while the camera stands still (same cache as `passes` consecutive render_frame calls below the
  // token cap).
  void render_frame_device(struct ui_state &state, const clw_foreign_memory &target, int passes = 1);

  // not in the reference: read-only access for tests and headless tools
  clw_vector<unsigned short> &voxel_cache() { return buffer_volume; }
  signed_distance_field &distance_field() { return sdf; }

 private:
  clw_context &ctx;
  clw_function render_func;
  clw_image<unsigned char, 4> frame;         // RGBA8 frame the caller blits
  clw_vector<unsigned short> buffer_volume;  // world-space radiance cache, 4 x u16 per voxel
  clw_image<unsigned char, 4> tfframe;
  const reference_volume *volume = nullptr;  // borrowed
  const env_map *emap = nullptr;             // borrowed
  signed_distance_field sdf;
  std::string local_cl_code;
};
