#include "signed_distance_field.hpp"

#include <clwh.h>

// The reference runs create_base_image and then up to 129 create_signed_distance_field launches,
// each followed by a blocking 4-byte push and pull of the progress counter
// (app/signed_distance_field.cpp:7-35).  The shim's clwh_sdf_build keeps the same layer semantics
// (same values, same final buffer) without a host round trip per layer.
signed_distance_field::signed_distance_field(clw_context &c, const reference_volume &rv, std::string local_cl_code)
    : sdf(c, std::vector<char>(rv.get_volume_length()), rv.get_volume_size(), false) {
  int32_t launches = 0;
  clw_fail_hard_on_error(clwh_sdf_build(c.get_handle(), rv.get_reference_volume().get_device_reference(),
                                        local_cl_code.c_str(), sdf.get_device_reference(), &launches));
  n_layers = launches;
}

signed_distance_field::signed_distance_field(clw_context &c) : sdf(c, std::vector<char>(8, 0), {2, 2, 2}, true) {}
