// jpeg_reader.hpp -- JPEG -> RGBA8 for environment maps ("You can also use regular JPG or PNG files for the
// environment map", the reference's README.md:29; it decodes them through the stb_image.h it vendors, with 4
// requested channels, app/hdre_loader.cpp:13).
//
// A JPEG decoder is not bit-exactly specified by the standard (ITU T.81): the inverse DCT, the chroma upsampling
// and the colour conversion are implementation choices.  This reader follows T.81 for the bitstream (baseline /
// extended sequential and progressive Huffman, 8 bit, 1 / 3 / 4 components, restart intervals) and makes the same
// three choices as the reference's decoder, so that the same file gives the same bytes
// (tests/test_ref_hostio.py compares against oracle/_ref, the reference's own code):
//   * IDCT: the libjpeg "islow" integer algorithm with 12-bit constants, 2 extra bits kept after the column pass;
//   * upsampling: 2x horizontal / vertical / both by the (3, 1) / 4 and (9, 3, 3, 1) / 16 triangle filters with the
//     edge samples repeated, anything else by sample repetition;
//   * YCbCr -> RGB in 20-bit fixed point with 12-bit coefficients; CMYK / YCCK (Adobe marker) through the
//     (x * k + 128) * 257 >> 16 product.
#pragma once

#include <string>
#include <vector>

/// true on success; on failure `error` says why
bool jpeg_decode_rgba8(const std::vector<unsigned char> &file, unsigned &width, unsigned &height,
                       std::vector<unsigned char> &rgba, std::string &error);
/// SOI marker at the start
bool jpeg_has_signature(const unsigned char *bytes, size_t n);
