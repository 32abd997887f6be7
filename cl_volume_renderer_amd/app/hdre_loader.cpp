#include "hdre_loader.hpp"

#include "jpeg_reader.hpp"
#include "png_reader.hpp"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <vector>

namespace {

[[noreturn]] void fail(const std::string &path, const char *why) {
  std::cerr << "Error, failed to load file: " << path << " (" << why << ")\n";
  std::exit(1);
}

// RGBE -> linear float: mantissa * 2^(e - 136); e == 0 means black
inline void rgbe_to_float(const unsigned char p[4], float out[3]) {
  if (p[3] == 0) {
    out[0] = out[1] = out[2] = 0.0f;
    return;
  }
  const float f = (float)std::ldexp(1.0f, (int)p[3] - (128 + 8));
  out[0] = p[0] * f;
  out[1] = p[1] * f;
  out[2] = p[2] * f;
}

// HDR -> LDR with gamma 2.2, scale 1: pow in double, the rest in float, truncation after + 0.5
inline unsigned char to_ldr(float v) {
  const float inv_gamma = 1 / 2.2f, inv_scale = 1 / 1.0f;
  float z = (float)std::pow((double)(v * inv_scale), (double)inv_gamma) * 255 + 0.5f;
  if (z < 0) z = 0;
  if (z > 255) z = 255;
  return (unsigned char)(int)z;
}

}  // namespace

image hdre_loader::load_file(const std::string path) {
  std::ifstream in(path, std::ios::in | std::ios::binary);
  if (in.fail()) fail(path, "cannot open");
  {
    // stb_image picks the decoder from the file's signature, not its name
    unsigned char sig[8] = {0};
    in.read(reinterpret_cast<char *>(sig), 8);
    const size_t got = (size_t)in.gcount();
    in.clear();
    in.seekg(0);
    const bool png = png_has_signature(sig, got), jpeg = jpeg_has_signature(sig, got);
    if (png || jpeg) {
      std::vector<unsigned char> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
      image out;
      std::string why;
      const bool ok = png ? png_decode_rgba8(file, out.m_width, out.m_height, out.m_pixels, why)
                          : jpeg_decode_rgba8(file, out.m_width, out.m_height, out.m_pixels, why);
      if (!ok) fail(path, why.c_str());
      return out;
    }
  }
  std::string line;
  std::getline(in, line);
  if (line != "#?RADIANCE" && line != "#?RGBE") fail(path, "not a Radiance HDR file");
  bool format_ok = false;
  while (std::getline(in, line) && !line.empty())
    if (line == "FORMAT=32-bit_rle_rgbe") format_ok = true;
  if (!format_ok) fail(path, "unsupported HDR format");
  std::getline(in, line);
  int height = 0, width = 0;
  if (std::sscanf(line.c_str(), "-Y %d +X %d", &height, &width) != 2 || width <= 0 || height <= 0)
    fail(path, "unsupported data layout");

  {
    // a corrupt resolution line must not turn into a huge allocation: a run-length scanline expands at most 64:1
    const std::streamoff here = in.tellg();
    in.seekg(0, std::ios::end);
    const unsigned long long remaining = (unsigned long long)(in.tellg() - here);
    in.seekg(here);
    if ((unsigned long long)width * (unsigned long long)height * 4ull > remaining * 64ull + 64ull) fail(path, "resolution exceeds the data");
  }
  std::vector<unsigned char> rgbe((size_t)width * height * 4);
  auto get = [&]() -> int { return in.get(); };
  bool flat = width < 8 || width >= 32768;
  size_t first_row = 0;
  if (!flat) {
    std::vector<unsigned char> scan((size_t)width * 4);
    for (int j = 0; j < height && !flat; ++j) {
      const int c1 = get(), c2 = get(), hi = get();
      if (c1 != 2 || c2 != 2 || (hi & 0x80)) {
        // an old-style file: these three bytes are the start of the first flat pixel
        if (j != 0) fail(path, "corrupt HDR");
        rgbe[0] = (unsigned char)c1; rgbe[1] = (unsigned char)c2; rgbe[2] = (unsigned char)hi; rgbe[3] = (unsigned char)get();
        flat = true;
        first_row = 4;
        break;
      }
      const int len = (hi << 8) | get();
      if (len != width) fail(path, "invalid decoded scanline length");
      for (int k = 0; k < 4; ++k) {
        int i = 0;
        while (i < width) {
          int count = get();
          if (count < 0) fail(path, "truncated");
          if (count > 128) {  // run
            const int value = get();
            count -= 128;
            if (count > width - i) fail(path, "bad RLE data in HDR");
            for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = (unsigned char)value;
          } else {  // literal bytes
            if (count > width - i) fail(path, "bad RLE data in HDR");
            for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = (unsigned char)get();
          }
        }
      }
      std::copy(scan.begin(), scan.end(), rgbe.begin() + (size_t)j * width * 4);
    }
  }
  if (flat) in.read(reinterpret_cast<char *>(rgbe.data() + first_row), (std::streamsize)(rgbe.size() - first_row));

  image out;
  out.m_width = (unsigned)width;
  out.m_height = (unsigned)height;
  out.m_pixels.resize((size_t)width * height * 4);
  for (size_t p = 0; p < (size_t)width * height; ++p) {
    float rgb[3];
    rgbe_to_float(&rgbe[p * 4], rgb);
    out.m_pixels[p * 4 + 0] = to_ldr(rgb[0]);
    out.m_pixels[p * 4 + 1] = to_ldr(rgb[1]);
    out.m_pixels[p * 4 + 2] = to_ldr(rgb[2]);
    out.m_pixels[p * 4 + 3] = 255;  // alpha 1.0 * 255 + 0.5
  }
  return out;
}
