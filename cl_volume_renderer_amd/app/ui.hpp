// ui.hpp -- the part of the reference's app/ui.hpp the hot path is called through: `ui_state`
// (:14-25) and the `frame_emitter` interface (:29-37).  The SDL/ImGui window class itself is the
// CALLER of this interface and is out of scope; these declarations keep its call sites compiling.
#pragma once
#include <string>

#include "common.hpp"
#include "env_map.hpp"
#include "reference_volume.hpp"

struct ui_state {
  std::string path;
  bool path_changed;
  int height;
  int width;
  Position3D position;      // camera position in voxel units
  float direction_look[2];  // yaw, pitch
  bool cam_changed;
};

class frame_emitter {
 public:
  virtual ~frame_emitter() {}
  virtual void image_set(const reference_volume *volume, const env_map *map) = 0;
  virtual void next_event_code_set(const std::string cl_code) = 0;
  virtual void flush_changes() = 0;
  virtual void *render_frame(struct ui_state &state, bool &frame_changed) = 0;
  virtual void *render_tf(const unsigned int width, const unsigned int height) = 0;
};
