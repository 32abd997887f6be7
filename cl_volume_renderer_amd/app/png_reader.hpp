// png_reader.hpp -- PNG -> RGBA8 for environment maps.  The reference loads every env-map format through
// stb_image with 4 requested channels (app/hdre_loader.cpp:13); this reader covers PNG the way that call
// behaves: all colour types and bit depths, Adam7 interlace, PLTE / tRNS, 16-bit samples reduced to their
// high byte, grey expanded to RGB, alpha 255 where the file has none.  Checksums are not verified (stb
// does not either).  Lossless format, so the result is defined by the PNG specification itself.
#pragma once

#include <string>
#include <vector>

/// true on success; on failure `error` says why
bool png_decode_rgba8(const std::vector<unsigned char> &file, unsigned &width, unsigned &height,
                      std::vector<unsigned char> &rgba, std::string &error);
/// the 8-byte PNG signature
bool png_has_signature(const unsigned char *bytes, size_t n);
