#include "jpeg_reader.hpp"

#include <cstdint>
#include <cstring>

namespace {

const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Fail {
  const char *why;
};

// canonical Huffman table (T.81 annex C / F.2.2.3): decode by code length
struct Huffman {
  bool defined = false;
  unsigned char values[256];
  int mincode[17], maxcode[18], valptr[17];
  void build(const int counts[16]) {
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valptr[len] = k;
      mincode[len] = code;
      k += counts[len - 1];
      code += counts[len - 1];
      if (counts[len - 1] && code - 1 >= (1 << len)) throw Fail{"bad Huffman code lengths"};
      maxcode[len] = counts[len - 1] ? code - 1 : -1;
      code <<= 1;
    }
    if (k > 256) throw Fail{"bad Huffman table size"};
    defined = true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
  int x = 0, y = 0;        // samples the component really has
  int w2 = 0, h2 = 0;      // padded to whole MCUs
  int blocks_w = 0;        // coefficient blocks per row of the padded grid
  int dc_pred = 0;
  std::vector<short> coeff;          // 64 per block, natural order, not yet dequantised
  std::vector<unsigned char> plane;  // w2 x h2 samples after the inverse DCT
};

class Decoder {
 public:
  Decoder(const unsigned char *data, size_t size) : d_(data), n_(size) {}

  void run(unsigned &width, unsigned &height, std::vector<unsigned char> &rgba) {
    if (marker() != 0xD8) throw Fail{"no SOI"};
    int m = marker();
    while (!(m == 0xC0 || m == 0xC1 || m == 0xC2)) {
      segment(m);
      m = marker();
      while (m == 0xFF) {
        if (pos_ >= n_) throw Fail{"no SOF"};
        m = marker();
      }
    }
    progressive_ = m == 0xC2;
    frame_header();
    m = marker();
    while (m != 0xD9) {
      if (m == 0xDA) {
        scan_header();
        scan();
        if (pending_ == 0xFF) {  // look for the marker that ends the entropy-coded segment
          while (pos_ < n_) {
            if (d_[pos_++] == 0xFF) {
              pending_ = get8();
              break;
            }
          }
        }
      } else if (m == 0xDC) {
        const int ld = get16(), nl = get16();
        if (ld != 4 || nl != (int)img_y_) throw Fail{"bad DNL"};
      } else {
        segment(m);
      }
      m = marker();  // no marker where one must be: `segment` rejects the file, as the reference's decoder does
    }
    reconstruct();
    output(rgba);
    width = img_x_;
    height = img_y_;
  }

 private:
  // ---- bytes and markers
  int get8() { return pos_ < n_ ? d_[pos_++] : 0; }
  int get16() { const int a = get8(); return (a << 8) | get8(); }
  void skip(int k) { pos_ = (k < 0 || (size_t)k > n_ - pos_) ? n_ : pos_ + (size_t)k; }
  // the pending marker of the entropy decoder, or the next one in the stream; 0xFF when there is none
  int marker() {
    if (pending_ != 0xFF) { const int x = pending_; pending_ = 0xFF; return x; }
    int x = get8();
    if (x != 0xFF) return 0xFF;
    while (x == 0xFF) x = get8();
    return x;
  }

  void segment(int m) {
    if (m == 0xFF) throw Fail{"expected marker"};
    if (m == 0xDD) {
      if (get16() != 4) throw Fail{"bad DRI length"};
      restart_interval_ = get16();
      return;
    }
    if (m == 0xDB) {
      int len = get16() - 2;
      while (len > 0) {
        const int q = get8(), precision = q >> 4, t = q & 15;
        if (precision > 1 || t > 3) throw Fail{"bad DQT"};
        for (int i = 0; i < 64; ++i) quant_[t][kZigzag[i]] = (uint16_t)(precision ? get16() : get8());
        len -= precision ? 129 : 65;
      }
      if (len != 0) throw Fail{"bad DQT length"};
      return;
    }
    if (m == 0xC4) {
      int len = get16() - 2;
      while (len > 0) {
        const int q = get8(), cls = q >> 4, id = q & 15;
        if (cls > 1 || id > 3) throw Fail{"bad DHT"};
        int counts[16], total = 0;
        for (int i = 0; i < 16; ++i) { counts[i] = get8(); total += counts[i]; }
        Huffman &h = cls ? ac_[id] : dc_[id];
        h.build(counts);
        for (int i = 0; i < total; ++i) h.values[i] = (unsigned char)get8();
        len -= 17 + total;
      }
      if (len != 0) throw Fail{"bad DHT length"};
      return;
    }
    if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
      int len = get16();
      if (len < 2) throw Fail{"bad APP / COM length"};
      len -= 2;
      if (m == 0xE0 && len >= 5) {
        static const char tag[5] = {'J', 'F', 'I', 'F', 0};
        bool ok = true;
        for (int i = 0; i < 5; ++i) ok &= get8() == (unsigned char)tag[i];
        len -= 5;
        if (ok) jfif_ = true;
      } else if (m == 0xEE && len >= 12) {
        static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
        bool ok = true;
        for (int i = 0; i < 6; ++i) ok &= get8() == (unsigned char)tag[i];
        len -= 6;
        if (ok) {
          get8(); get16(); get16();
          adobe_transform_ = get8();
          len -= 6;
        }
      }
      skip(len);
      return;
    }
    throw Fail{"unknown marker"};
  }

  void frame_header() {
    const int lf = get16();
    if (lf < 11) throw Fail{"bad SOF length"};
    if (get8() != 8) throw Fail{"only 8-bit JPEG is supported"};
    img_y_ = (unsigned)get16();
    img_x_ = (unsigned)get16();
    if (img_y_ == 0 || img_x_ == 0) throw Fail{"zero image size"};
    ncomp_ = get8();
    if (ncomp_ != 1 && ncomp_ != 3 && ncomp_ != 4) throw Fail{"bad component count"};
    if (lf != 8 + 3 * ncomp_) throw Fail{"bad SOF length"};
    if ((uint64_t)img_x_ * img_y_ > (1ull << 28)) throw Fail{"image too large"};
    rgb_ids_ = 0;
    for (int i = 0; i < ncomp_; ++i) {
      Component &c = comp_[i];
      c.id = get8();
      if (ncomp_ == 3 && c.id == "RGB"[i]) ++rgb_ids_;
      const int q = get8();
      c.h = q >> 4; c.v = q & 15; c.tq = get8();
      if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) throw Fail{"bad sampling factors or table"};
      if (c.h > h_max_) h_max_ = c.h;
      if (c.v > v_max_) v_max_ = c.v;
    }
    mcu_x_ = ((int)img_x_ + h_max_ * 8 - 1) / (h_max_ * 8);
    mcu_y_ = ((int)img_y_ + v_max_ * 8 - 1) / (v_max_ * 8);
    for (int i = 0; i < ncomp_; ++i) {
      Component &c = comp_[i];
      c.x = ((int)img_x_ * c.h + h_max_ - 1) / h_max_;
      c.y = ((int)img_y_ * c.v + v_max_ - 1) / v_max_;
      c.w2 = mcu_x_ * c.h * 8;
      c.h2 = mcu_y_ * c.v * 8;
      c.blocks_w = c.w2 / 8;
      c.coeff.assign((size_t)c.w2 * c.h2, 0);
      c.plane.assign((size_t)c.w2 * c.h2, 0);
    }
  }

  void scan_header() {
    const int ls = get16();
    scan_n_ = get8();
    if (scan_n_ < 1 || scan_n_ > 4 || scan_n_ > ncomp_ || ls != 6 + 2 * scan_n_) throw Fail{"bad SOS"};
    for (int i = 0; i < scan_n_; ++i) {
      const int id = get8(), q = get8();
      int which = 0;
      while (which < ncomp_ && comp_[which].id != id) ++which;
      if (which == ncomp_) throw Fail{"SOS names an unknown component"};
      comp_[which].hd = q >> 4; comp_[which].ha = q & 15;
      if (comp_[which].hd > 3 || comp_[which].ha > 3) throw Fail{"bad Huffman table index"};
      order_[i] = which;
    }
    ss_ = get8(); se_ = get8();
    const int a = get8();
    ah_ = a >> 4; al_ = a & 15;
    if (progressive_) {
      if (ss_ > 63 || se_ > 63 || ss_ > se_ || ah_ > 13 || al_ > 13) throw Fail{"bad SOS"};
    } else {
      if (ss_ != 0 || ah_ != 0 || al_ != 0) throw Fail{"bad SOS"};
      se_ = 63;
    }
  }

  // ---- entropy-coded data: MSB-first bit reader; after a marker (or the end of the file) only zero bits follow
  void reset_entropy() {
    bits_ = 0; nbits_ = 0; exhausted_ = false;
    for (int i = 0; i < 4; ++i) comp_[i].dc_pred = 0;
    pending_ = 0xFF;
    todo_ = restart_interval_ ? restart_interval_ : 0x7fffffff;
    eob_run_ = 0;
  }
  void refill() {
    if (nbits_ < 0) nbits_ = 0;
    do {
      unsigned b = exhausted_ ? 0u : (unsigned)get8();
      if (b == 0xFF) {
        int c = get8();
        while (c == 0xFF) c = get8();
        if (c != 0) { pending_ = c; exhausted_ = true; return; }
      }
      bits_ |= b << (24 - nbits_);
      nbits_ += 8;
    } while (nbits_ <= 24);
  }
  int bit() { return get_bits(1); }
  int get_bits(int n) {
    if (n == 0) return 0;
    if (nbits_ < n) refill();
    const int v = (int)(bits_ >> (32 - n));
    if (nbits_ < n) {  // a marker cut the data short: the missing bits read as zero
      bits_ = 0; nbits_ = 0;
      return v;
    }
    bits_ <<= n; nbits_ -= n;
    return v;
  }
  int extend(int n) {  // T.81 F.2.2.1 EXTEND(RECEIVE(n), n)
    if (n == 0) return 0;
    const int v = get_bits(n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
  }
  int decode(const Huffman &h) {
    if (!h.defined) throw Fail{"scan uses an undefined Huffman table"};
    if (nbits_ < 16) refill();
    int code = 0;
    for (int len = 1; len <= 16; ++len) {
      code = (int)(bits_ >> (32 - len));
      if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) {
        if (len > nbits_) throw Fail{"bad Huffman code"};
        bits_ <<= len; nbits_ -= len;
        return h.values[h.valptr[len] + code - h.mincode[len]];
      }
    }
    throw Fail{"bad Huffman code"};
  }

  short *block(Component &c, int bx, int by) { return &c.coeff[((size_t)by * c.blocks_w + bx) * 64]; }

  void block_sequential(Component &c, short *b) {
    const int t = decode(dc_[c.hd]);
    if (t > 15) throw Fail{"bad DC difference"};
    c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)extend(t));  // (wraps instead of overflowing on corrupt streams)
    b[0] = (short)c.dc_pred;
    const Huffman &h = ac_[c.ha];
    int k = 1;
    while (k < 64) {
      const int rs = decode(h), s = rs & 15, r = rs >> 4;
      if (s == 0) {
        if (rs != 0xF0) break;
        k += 16;
      } else {
        k += r;
        if (k > 63) throw Fail{"bad AC run"};
        b[kZigzag[k++]] = (short)extend(s);
      }
    }
  }
  void block_dc_progressive(Component &c, short *b) {
    if (se_ != 0) throw Fail{"DC and AC in one progressive scan"};
    if (ah_ == 0) {
      const int t = decode(dc_[c.hd]);
      if (t > 15) throw Fail{"bad DC difference"};
      c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)extend(t));
      b[0] = (short)((unsigned)c.dc_pred << al_);
    } else if (bit()) {
      b[0] = (short)(b[0] + (1 << al_));
    }
  }
  void block_ac_progressive(Component &c, short *b) {
    if (ss_ == 0) throw Fail{"DC and AC in one progressive scan"};
    const Huffman &h = ac_[c.ha];
    if (ah_ == 0) {  // first pass over this band (T.81 G.1.2.2)
      if (eob_run_) { --eob_run_; return; }
      int k = ss_;
      do {
        const int rs = decode(h), s = rs & 15, r = rs >> 4;
        if (s == 0) {
          if (r < 15) {
            eob_run_ = 1 << r;
            if (r) eob_run_ += get_bits(r);
            --eob_run_;
            break;
          }
          k += 16;
        } else {
          k += r;
          if (k > 63) throw Fail{"bad AC run"};
          b[kZigzag[k++]] = (short)(extend(s) * (1 << al_));
        }
      } while (k <= se_);
      return;
    }
    // refinement (T.81 G.1.2.3): one more bit for the coefficients that are already non-zero, new +-1 ones in between
    const short one = (short)(1 << al_);
    auto refine = [&](short &v) {
      if (bit() && (v & one) == 0) v = (short)(v > 0 ? v + one : v - one);
    };
    if (eob_run_) {
      --eob_run_;
      for (int k = ss_; k <= se_; ++k) {
        short &v = b[kZigzag[k]];
        if (v != 0) refine(v);
      }
      return;
    }
    int k = ss_;
    do {
      const int rs = decode(h);
      int s = rs & 15, r = rs >> 4;
      if (s == 0) {
        if (r < 15) {
          eob_run_ = (1 << r) - 1;
          if (r) eob_run_ += get_bits(r);
          r = 64;  // run to the end of the band, refining on the way
        }
      } else {
        if (s != 1) throw Fail{"bad refinement code"};
        s = bit() ? one : -one;
      }
      while (k <= se_) {
        short &v = b[kZigzag[k++]];
        if (v != 0) {
          refine(v);
        } else {
          if (r == 0) { v = (short)s; break; }
          --r;
        }
      }
    } while (k <= se_);
  }

  // end of a restart interval: the next thing in the stream must be RSTn, otherwise the scan ends here
  bool interval_done() {
    if (--todo_ > 0) return true;
    if (nbits_ < 24) refill();
    if (!(pending_ >= 0xD0 && pending_ <= 0xD7)) return false;
    reset_entropy();
    return true;
  }

  void scan() {
    reset_entropy();
    auto one_block = [&](Component &c, int bx, int by) {
      short *b = block(c, bx, by);
      if (!progressive_) block_sequential(c, b);
      else if (ss_ == 0) block_dc_progressive(c, b);
      else block_ac_progressive(c, b);
    };
    if (scan_n_ == 1) {
      Component &c = comp_[order_[0]];
      const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;  // the blocks the component really has
      for (int by = 0; by < h; ++by)
        for (int bx = 0; bx < w; ++bx) {
          one_block(c, bx, by);
          if (!interval_done()) return;
        }
      return;
    }
    if (progressive_ && ss_ != 0) throw Fail{"interleaved AC scan"};
    for (int my = 0; my < mcu_y_; ++my)
      for (int mx = 0; mx < mcu_x_; ++mx) {
        for (int k = 0; k < scan_n_; ++k) {
          Component &c = comp_[order_[k]];
          for (int y = 0; y < c.v; ++y)
            for (int x = 0; x < c.h; ++x) one_block(c, mx * c.h + x, my * c.v + y);
        }
        if (!interval_done()) return;
      }
  }

  // ---- inverse DCT: libjpeg's jidctint ("islow") arithmetic, constants = trunc(c * 4096 + 0.5) of the float
  // literals (so the negative ones are one above minus the positive ones), rounding 512 >> 10 after the columns
  // (two guard bits), (65536 + (128 << 17)) >> 17 after the rows, clamp to 0..255
  // (64-bit intermediates: identical to 32-bit arithmetic for every valid stream, no signed overflow on corrupt ones)
  typedef long long wide;
  static void idct_1d(const wide s[8], wide e[4], wide o[4]) {
    const wide z = (s[2] + s[6]) * 2217;
    const wide e2 = z + s[6] * -7567, e3 = z + s[2] * 3135;
    const wide e0 = (s[0] + s[4]) * 4096, e1 = (s[0] - s[4]) * 4096;
    e[0] = e0 + e3; e[3] = e0 - e3; e[1] = e1 + e2; e[2] = e1 - e2;
    const wide p3 = s[7] + s[3], p4 = s[5] + s[1], p1 = s[7] + s[1], p2 = s[5] + s[3];
    const wide p5 = (p3 + p4) * 4816;
    const wide q1 = p5 + p1 * -3685, q2 = p5 + p2 * -10497, q3 = p3 * -8034, q4 = p4 * -1597;
    o[3] = s[1] * 6149 + q1 + q4;   // pairs with e[0]
    o[2] = s[3] * 12586 + q2 + q3;  // pairs with e[1]
    o[1] = s[5] * 8410 + q2 + q4;   // pairs with e[2]
    o[0] = s[7] * 1223 + q1 + q3;   // pairs with e[3]
  }
  static unsigned char clamp8(long long v) { return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }
  static void idct(const short *in, unsigned char *out, int stride) {
    wide tmp[64];
    for (int c = 0; c < 8; ++c) {
      const short *d = in + c;
      if (!(d[8] | d[16] | d[24] | d[32] | d[40] | d[48] | d[56])) {
        for (int r = 0; r < 8; ++r) tmp[r * 8 + c] = d[0] * 4;
        continue;
      }
      const wide s[8] = {d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56]};
      wide e[4], o[4];
      idct_1d(s, e, o);
      for (int k = 0; k < 4; ++k) {
        tmp[k * 8 + c] = (e[k] + 512 + o[3 - k]) >> 10;
        tmp[(7 - k) * 8 + c] = (e[k] + 512 - o[3 - k]) >> 10;
      }
    }
    for (int r = 0; r < 8; ++r) {
      wide e[4], o[4];
      idct_1d(tmp + r * 8, e, o);
      unsigned char *row = out + (size_t)r * stride;
      const wide bias = 65536 + (128 << 17);
      for (int k = 0; k < 4; ++k) {
        row[k] = clamp8((e[k] + bias + o[3 - k]) >> 17);
        row[7 - k] = clamp8((e[k] + bias - o[3 - k]) >> 17);
      }
    }
  }

  void reconstruct() {
    for (int i = 0; i < ncomp_; ++i) {
      Component &c = comp_[i];
      const uint16_t *q = quant_[c.tq];
      // only the blocks that carry samples; the rest of the MCU padding never reaches the output
      const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
      for (int by = 0; by < h; ++by)
        for (int bx = 0; bx < w; ++bx) {
          short *b = block(c, bx, by);
          for (int k = 0; k < 64; ++k) b[k] = (short)(b[k] * q[k]);
          idct(b, &c.plane[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2);
        }
    }
  }

  // ---- upsampling of one output row of a component into `line`; `near` is the closer source row, `far` the other
  static const unsigned char *upsample(unsigned char *line, const unsigned char *near, const unsigned char *far, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) {
      for (int i = 0; i < w; ++i) line[i] = (unsigned char)((3 * near[i] + far[i] + 2) >> 2);
      return line;
    }
    if (hs == 2 && vs == 1) {
      if (w == 1) { line[0] = line[1] = near[0]; return line; }
      line[0] = near[0];
      line[1] = (unsigned char)((near[0] * 3 + near[1] + 2) >> 2);
      for (int i = 1; i < w - 1; ++i) {
        const int n = 3 * near[i] + 2;
        line[i * 2] = (unsigned char)((n + near[i - 1]) >> 2);
        line[i * 2 + 1] = (unsigned char)((n + near[i + 1]) >> 2);
      }
      line[(w - 1) * 2] = (unsigned char)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
      line[(w - 1) * 2 + 1] = near[w - 1];
      return line;
    }
    if (hs == 2 && vs == 2) {
      int t1 = 3 * near[0] + far[0];
      if (w == 1) { line[0] = line[1] = (unsigned char)((t1 + 2) >> 2); return line; }
      line[0] = (unsigned char)((t1 + 2) >> 2);
      for (int i = 1; i < w; ++i) {
        const int t0 = t1;
        t1 = 3 * near[i] + far[i];
        line[i * 2 - 1] = (unsigned char)((3 * t0 + t1 + 8) >> 4);
        line[i * 2] = (unsigned char)((3 * t1 + t0 + 8) >> 4);
      }
      line[w * 2 - 1] = (unsigned char)((t1 + 2) >> 2);
      return line;
    }
    for (int i = 0; i < w; ++i)
      for (int j = 0; j < hs; ++j) line[i * hs + j] = near[i];
    return line;
  }

  static unsigned char mul8(unsigned x, unsigned y) {  // x * y / 255, rounded
    const unsigned t = x * y + 128;
    return (unsigned char)((t + (t >> 8)) >> 8);
  }

  void output(std::vector<unsigned char> &rgba) {
    rgba.assign((size_t)img_x_ * img_y_ * 4, 255);
    struct Row { int hs, vs, ystep, w_lores, ypos; const unsigned char *line0, *line1; std::vector<unsigned char> buf; };
    Row rows[4];
    for (int k = 0; k < ncomp_; ++k) {
      Row &r = rows[k];
      r.hs = h_max_ / comp_[k].h;
      r.vs = v_max_ / comp_[k].v;
      r.ystep = r.vs >> 1;
      r.w_lores = ((int)img_x_ + r.hs - 1) / r.hs;
      r.ypos = 0;
      r.line0 = r.line1 = comp_[k].plane.data();
      r.buf.assign((size_t)img_x_ + 8, 0);
    }
    const bool is_rgb = ncomp_ == 3 && (rgb_ids_ == 3 || (adobe_transform_ == 0 && !jfif_));
    for (unsigned j = 0; j < img_y_; ++j) {
      const unsigned char *c[4] = {nullptr, nullptr, nullptr, nullptr};
      for (int k = 0; k < ncomp_; ++k) {
        Row &r = rows[k];
        const bool bottom = r.ystep >= (r.vs >> 1);
        c[k] = upsample(r.buf.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
        if (++r.ystep >= r.vs) {
          r.ystep = 0;
          r.line0 = r.line1;
          if (++r.ypos < comp_[k].y) r.line1 += comp_[k].w2;
        }
      }
      unsigned char *out = &rgba[(size_t)j * img_x_ * 4];
      auto ycc = [&](unsigned i, unsigned char *o) {
        const int yf = (c[0][i] << 20) + (1 << 19);
        const int cr = c[2][i] - 128, cb = c[1][i] - 128;
        int r = yf + cr * 1470208;
        int g = yf + cr * -748800 + (int)((unsigned)(cb * -360960) & 0xffff0000u);
        int b = yf + cb * 1858048;
        r >>= 20; g >>= 20; b >>= 20;
        o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(b);
      };
      for (unsigned i = 0; i < img_x_; ++i, out += 4) {
        if (ncomp_ == 1) {
          out[0] = out[1] = out[2] = c[0][i];
        } else if (ncomp_ == 3) {
          if (is_rgb) { out[0] = c[0][i]; out[1] = c[1][i]; out[2] = c[2][i]; }
          else ycc(i, out);
        } else if (adobe_transform_ == 0) {  // CMYK
          const unsigned k = c[3][i];
          out[0] = mul8(c[0][i], k); out[1] = mul8(c[1][i], k); out[2] = mul8(c[2][i], k);
        } else if (adobe_transform_ == 2) {  // YCCK
          ycc(i, out);
          const unsigned k = c[3][i];
          out[0] = mul8(255u - out[0], k); out[1] = mul8(255u - out[1], k); out[2] = mul8(255u - out[2], k);
        } else {
          ycc(i, out);
        }
      }
    }
  }

  const unsigned char *d_;
  size_t n_, pos_ = 0;
  int pending_ = 0xFF;
  bool progressive_ = false, jfif_ = false, exhausted_ = false;
  int adobe_transform_ = -1, rgb_ids_ = 0;
  unsigned img_x_ = 0, img_y_ = 0;
  int ncomp_ = 0, h_max_ = 1, v_max_ = 1, mcu_x_ = 0, mcu_y_ = 0;
  int restart_interval_ = 0, todo_ = 0, eob_run_ = 0;
  int scan_n_ = 0, order_[4] = {0, 0, 0, 0}, ss_ = 0, se_ = 63, ah_ = 0, al_ = 0;
  uint32_t bits_ = 0;
  int nbits_ = 0;
  uint16_t quant_[4][64] = {};
  Huffman dc_[4], ac_[4];
  Component comp_[4];
};

}  // namespace

bool jpeg_has_signature(const unsigned char *b, size_t n) { return n >= 3 && b[0] == 0xFF && b[1] == 0xD8 && b[2] == 0xFF; }

bool jpeg_decode_rgba8(const std::vector<unsigned char> &file, unsigned &width, unsigned &height,
                       std::vector<unsigned char> &rgba, std::string &error) {
  try {
    Decoder d(file.data(), file.size());
    d.run(width, height, rgba);
    return true;
  } catch (const Fail &f) {
    error = f.why;
    return false;
  } catch (const std::bad_alloc &) {
    error = "out of memory";
    return false;
  }
}
