// hdre_loader.hpp -- Radiance RGBE (.hdr) environment map -> RGBA8 image: the step before the hot path
// on the env-map side (mirror of the reference's app/hdre_loader.hpp).
// The reference decodes through stb_image with `stbi_hdr_to_ldr_gamma(2.2f)`, scale 1, 4 channels
// (app/hdre_loader.cpp:7-24), i.e. per colour channel  byte = clamp(pow(x, 1/2.2) * 255 + 0.5), alpha 255.
// This reader implements the Radiance format itself (flat and run-length scanlines, `-Y h +X w`) and that
// conversion.  PNG and JPEG files (recognised by signature, as stb does) go through png_reader.hpp and
// jpeg_reader.hpp; the other containers stb accepts (BMP, TGA, GIF, PSD, PIC, PNM) are not handled.
#pragma once

#include <string>

#include "image.hpp"

class hdre_loader {
 public:
  /// fatal (exit 1) on unreadable or unsupported files, like the reference
  image load_file(const std::string path);
};
