// image.hpp -- RGBA8 host image, the output type of the reference's env-map loader
// (app/image.hpp, app/hdre_loader.hpp); decoding .hdr/.png is outside the hot path.
#pragma once
#include <vector>

struct image {
  std::vector<unsigned char> m_pixels;  // RGBA8, row-major
  unsigned int m_width = 0;
  unsigned int m_height = 0;
};
