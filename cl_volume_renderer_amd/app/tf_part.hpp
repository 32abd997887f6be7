// tf_part.hpp -- transfer-function selections -> the `is_event_gen` source the renderer is flushed with.
// The reference's app/tf_part.{hpp,cpp} mixes ImGui widgets with the code generator; only the generator is
// an input of the hot path, so only it is mirrored: tf_rect_selection::create_cl_condition
// (app/tf_part.cpp:55-79) and the assembly done by ui::flush_tf (app/ui.cpp:160-168).
#pragma once

#include <string>
#include <vector>

#include "reference_volume.hpp"

class tf_selection {
 public:
  virtual ~tf_selection() {}
  virtual std::string create_cl_condition(Volume_Stats stats) = 0;
};

class tf_rect_selection : public tf_selection {
 public:
  tf_rect_selection(unsigned id, float min_v, float max_v, float min_g, float max_g);
  std::string create_cl_condition(Volume_Stats stats) override;
  float color[4] = {1.0f, 1.0f, 1.0f, 1.0f};  // r, g, b, roughness
  float min_v, max_v, min_g, max_g;

 private:
  unsigned id;
};

/// "inline bool is_event_gen(short value, short gradient, int4 *color){ ... return false; }"
std::string tf_generate_source(Volume_Stats stats, const std::vector<tf_selection *> &selections);
