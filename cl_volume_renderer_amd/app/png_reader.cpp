#include "png_reader.hpp"

#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace {

inline uint32_t be32(const unsigned char *p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

inline int paeth(int a, int b, int c) {
  const int p = a + b - c;
  const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  if (pb <= pc) return b;
  return c;
}

// undo the per-scanline filters of one (sub-)image in place; `raw` holds rows of 1 + stride bytes
bool unfilter(unsigned char *raw, size_t rows, size_t stride, size_t bpp) {
  const unsigned char *prev = nullptr;
  for (size_t y = 0; y < rows; ++y) {
    unsigned char *line = raw + y * (stride + 1);
    const int type = line[0];
    unsigned char *cur = line + 1;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0;
      const int b = prev ? prev[i] : 0;
      const int c = (prev && i >= bpp) ? prev[i - bpp] : 0;
      int v = cur[i];
      switch (type) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: return false;
      }
      cur[i] = (unsigned char)v;
    }
    prev = cur;
  }
  return true;
}

struct Header {
  unsigned w = 0, h = 0;
  int depth = 0, color = 0, interlace = 0;
  int channels() const { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4; }
};

// sample `index` of a scanline (bit depths 1, 2, 4, 8, 16; 16-bit returns the full value)
inline unsigned sample_at(const unsigned char *row, size_t index, int depth) {
  if (depth == 8) return row[index];
  if (depth == 16) return ((unsigned)row[index * 2] << 8) | row[index * 2 + 1];
  const size_t bit = index * (size_t)depth;
  const unsigned byte = row[bit >> 3];
  const unsigned shift = 8u - (unsigned)depth - (unsigned)(bit & 7u);
  return (byte >> shift) & ((1u << depth) - 1u);
}

}  // namespace

bool png_has_signature(const unsigned char *b, size_t n) {
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  return n >= 8 && std::memcmp(b, sig, 8) == 0;
}

bool png_decode_rgba8(const std::vector<unsigned char> &file, unsigned &width, unsigned &height,
                      std::vector<unsigned char> &rgba, std::string &error) {
  if (!png_has_signature(file.data(), file.size())) { error = "not a PNG"; return false; }
  Header hd;
  bool have_header = false;
  std::vector<unsigned char> idat, palette, trns;
  size_t pos = 8;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const unsigned char *type = &file[pos + 4];
    const unsigned char *data = &file[pos + 8];
    if ((size_t)len > file.size() - pos - 12) { error = "truncated chunk"; return false; }
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) { error = "bad IHDR"; return false; }
      hd.w = be32(data); hd.h = be32(data + 4);
      hd.depth = data[8]; hd.color = data[9]; hd.interlace = data[12];
      if (data[10] != 0 || data[11] != 0) { error = "unknown compression or filter method"; return false; }
      have_header = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      palette.assign(data, data + len);
    } else if (!std::memcmp(type, "tRNS", 4)) {
      trns.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_header || hd.w == 0 || hd.h == 0) { error = "missing IHDR"; return false; }
  const bool depth_ok = (hd.color == 0 && (hd.depth == 1 || hd.depth == 2 || hd.depth == 4 || hd.depth == 8 || hd.depth == 16)) ||
                        (hd.color == 3 && (hd.depth == 1 || hd.depth == 2 || hd.depth == 4 || hd.depth == 8)) ||
                        ((hd.color == 2 || hd.color == 4 || hd.color == 6) && (hd.depth == 8 || hd.depth == 16));
  if (!depth_ok || hd.interlace > 1) { error = "unsupported colour type / bit depth / interlace"; return false; }
  if (hd.color == 3 && palette.size() < 3) { error = "palette image without PLTE"; return false; }
  if ((uint64_t)hd.w * hd.h > (1ull << 31)) { error = "image too large"; return false; }

  const int ch = hd.channels();
  const size_t bits_pp = (size_t)ch * (size_t)hd.depth;
  const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
  // the (sub-)images of the stream: one, or the seven Adam7 passes
  struct Pass { unsigned x0, y0, dx, dy, w, h; };
  std::vector<Pass> passes;
  if (!hd.interlace) {
    passes.push_back({0, 0, 1, 1, hd.w, hd.h});
  } else {
    static const unsigned X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1};
    static const unsigned DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
    for (int p = 0; p < 7; ++p) {
      const unsigned pw = hd.w > X0[p] ? (hd.w - X0[p] + DX[p] - 1) / DX[p] : 0;
      const unsigned ph = hd.h > Y0[p] ? (hd.h - Y0[p] + DY[p] - 1) / DY[p] : 0;
      if (pw && ph) passes.push_back({X0[p], Y0[p], DX[p], DY[p], pw, ph});
    }
  }
  size_t raw_size = 0;
  for (const Pass &p : passes) raw_size += (size_t)p.h * (1 + ((size_t)p.w * bits_pp + 7) / 8);
  std::vector<unsigned char> raw(raw_size);
  {
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) { error = "zlib init"; return false; }
    zs.next_in = idat.data();
    zs.avail_in = (uInt)idat.size();
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw.size();
    const int rc = inflate(&zs, Z_FINISH);
    const size_t got = raw.size() - zs.avail_out;
    inflateEnd(&zs);
    if ((rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) || got != raw.size()) { error = "corrupt or short image data"; return false; }
  }

  width = hd.w; height = hd.h;
  rgba.assign((size_t)hd.w * hd.h * 4, 0);
  // low bit depths of greyscale are scaled to 0..255; palette indices are not
  static const unsigned grey_scale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
  size_t off = 0;
  for (const Pass &p : passes) {
    const size_t stride = ((size_t)p.w * bits_pp + 7) / 8;
    if (!unfilter(raw.data() + off, p.h, stride, bpp)) { error = "unknown filter type"; return false; }
    for (unsigned y = 0; y < p.h; ++y) {
      const unsigned char *row = raw.data() + off + (size_t)y * (stride + 1) + 1;
      for (unsigned x = 0; x < p.w; ++x) {
        unsigned s[4] = {0, 0, 0, 0};
        for (int c = 0; c < ch; ++c) s[c] = sample_at(row, (size_t)x * ch + c, hd.depth);
        unsigned r, g, b, a = 255;
        if (hd.color == 3) {
          const unsigned idx = s[0];
          if ((size_t)idx * 3 + 2 >= palette.size()) { error = "palette index out of range"; return false; }
          r = palette[idx * 3]; g = palette[idx * 3 + 1]; b = palette[idx * 3 + 2];
          if (idx < trns.size()) a = trns[idx];
        } else {
          // tRNS names one fully transparent colour (compared at the file's bit depth)
          bool transparent = false;
          if (hd.color == 0 && trns.size() >= 2) transparent = s[0] == ((((unsigned)trns[0] << 8) | trns[1]) & ((1u << hd.depth) - 1u));
          if (hd.color == 2 && trns.size() >= 6) {
            const unsigned m = (1u << hd.depth) - 1u;
            transparent = s[0] == ((((unsigned)trns[0] << 8) | trns[1]) & m) && s[1] == ((((unsigned)trns[2] << 8) | trns[3]) & m) &&
                          s[2] == ((((unsigned)trns[4] << 8) | trns[5]) & m);
          }
          auto to8 = [&](unsigned v) -> unsigned {
            if (hd.depth == 16) return v >> 8;
            if (hd.depth == 8) return v;
            return v * grey_scale[hd.depth];
          };
          if (hd.color == 0) { r = g = b = to8(s[0]); }
          else if (hd.color == 4) { r = g = b = to8(s[0]); a = to8(s[1]); }
          else if (hd.color == 2) { r = to8(s[0]); g = to8(s[1]); b = to8(s[2]); }
          else { r = to8(s[0]); g = to8(s[1]); b = to8(s[2]); a = to8(s[3]); }
          if (transparent) a = 0;
        }
        unsigned char *o = &rgba[(((size_t)p.y0 + (size_t)y * p.dy) * hd.w + (p.x0 + (size_t)x * p.dx)) * 4];
        o[0] = (unsigned char)r; o[1] = (unsigned char)g; o[2] = (unsigned char)b; o[3] = (unsigned char)a;
      }
    }
    off += (size_t)p.h * (stride + 1);
  }
  return true;
}
