// signed_distance_field.hpp -- the SDF image used for empty-space skipping
// (reference app/signed_distance_field.hpp:5-12).
#pragma once
#include <string>

#include "reference_volume.hpp"

class signed_distance_field {
 public:
  explicit signed_distance_field(clw_context &c);
  signed_distance_field(clw_context &c, const reference_volume &rf, std::string local_cl_code);
  clw_image<char> &get_sdf_buffer() { return sdf; }
  int layers() const { return n_layers; }  // create_signed_distance_field launches the reference's loop would run

 private:
  clw_image<char> sdf;
  int n_layers = 0;
};
