// env_map.hpp -- RGBA8 environment map held as a 2-D device image (reference app/env_map.hpp:6-15)
#pragma once

#include <clw_image.hpp>

#include "image.hpp"

class env_map {
 public:
  env_map(clw_context &ctx, image &source)
      : buffer(ctx, std::move(source.m_pixels), {source.m_width, source.m_height, 1}, true) {}
  const clw_image<unsigned char, 4> &get_buffer() const { return buffer; }

 private:
  clw_image<unsigned char, 4> buffer;
};
