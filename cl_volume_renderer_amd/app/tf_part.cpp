#include "tf_part.hpp"

#include <sstream>

tf_rect_selection::tf_rect_selection(unsigned i, float lo_v, float hi_v, float lo_g, float hi_g)
    : min_v(lo_v), max_v(hi_v), min_g(lo_g), max_g(hi_g), id(i * 100) {}

// one `if` per rectangle; floats go through operator<< with default flags (6 significant digits); the
// gradient clause is emitted only when the rectangle is narrower than the volume's gradient range
std::string tf_rect_selection::create_cl_condition(Volume_Stats stats) {
  std::ostringstream code;
  code << "  if(value >= " << min_v << " && value <= " << max_v;
  if (min_g > stats.min_g || max_g < stats.max_g) code << " && gradient > " << min_g << " && gradient < " << max_g;
  code << ")\n {\n";
  code << "    int4 tmp_color = {" << (int)(color[0] * 255) << "," << (int)(color[1] * 255) << "," << (int)(color[2] * 255)
       << "," << (int)(color[3] * 255) << "};\n";
  code << "    *color = tmp_color;\n    return true;\n }\n";
  return code.str();
}

std::string tf_generate_source(Volume_Stats stats, const std::vector<tf_selection *> &selections) {
  std::string code = "inline bool is_event_gen(short value, short gradient, int4 *color){\n";
  for (tf_selection *s : selections) code += s->create_cl_condition(stats);
  code += "  \n  return false;\n}\n";
  return code;
}
