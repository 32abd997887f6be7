// fuzz_parsers.cpp -- mutation fuzzing of the two parsers that take caller-controlled bytes and report errors by
// return value: the PNG reader (app/png_reader.cpp) and the transfer-function source parser (csrc/tf_parse.cpp).
// Built by tests/test_fuzz_parsers.py with -fsanitize=address,undefined (CPU build only) and run for a bounded
// number of iterations; any sanitizer report aborts the process and fails the test.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

#include "../../cl_volume_renderer_amd/app/hdre_loader.hpp"
#include "../../cl_volume_renderer_amd/app/jpeg_reader.hpp"
#include "../../cl_volume_renderer_amd/app/nrrd_loader.hpp"
#include "../../cl_volume_renderer_amd/app/png_reader.hpp"
#include "../../include/clwh.h"

namespace {
uint64_t rng_state = 0x9E3779B97F4A7C15ull;
uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 11);
}
std::vector<unsigned char> read_file(const char *p) {
  std::ifstream in(p, std::ios::binary);
  return std::vector<unsigned char>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
void mutate(std::vector<unsigned char> &b) {
  if (b.empty()) return;
  const int edits = 1 + (int)(rnd() % 6);
  for (int e = 0; e < edits; ++e) {
    const uint32_t kind = rnd() % 6, at = rnd() % (uint32_t)b.size();
    switch (kind) {
      case 0: b[at] = (unsigned char)rnd(); break;
      case 1: b[at] ^= (unsigned char)(1u << (rnd() % 8)); break;
      case 2: b.resize(at + 1); break;                                   // truncate
      case 3: b.insert(b.begin() + at, (unsigned char)rnd()); break;     // insert
      case 4: b.erase(b.begin() + at); if (b.empty()) b.push_back(0); break;
      default: {                                                          // overwrite 4 bytes with an extreme value
        static const unsigned char ext[4][4] = {{0xff, 0xff, 0xff, 0xff}, {0x7f, 0xff, 0xff, 0xff}, {0, 0, 0, 0}, {0x80, 0, 0, 0}};
        const unsigned char *v = ext[rnd() % 4];
        for (uint32_t k = 0; k < 4 && at + k < b.size(); ++k) b[at + k] = v[k];
      }
    }
  }
}
}  // namespace

// The file loaders follow the reference's fail-hard convention (message + exit(1)), so each mutated file is loaded
// in a forked child: exit status 0 or 1 is fine, anything else (a sanitizer report exits with ASAN_OPTIONS'
// exitcode, UBSan aborts, a crash is a signal) fails the run.
int fuzz_loader(const std::string &kind, long iters, const std::vector<std::vector<unsigned char>> &seeds, const char *scratch) {
  long ok = 0, rejected = 0;
  for (long it = 0; it < iters; ++it) {
    std::vector<unsigned char> b = seeds[rnd() % seeds.size()];
    if (it >= (long)seeds.size()) mutate(b);
    {
      std::ofstream out(scratch, std::ios::binary | std::ios::trunc);
      out.write(reinterpret_cast<const char *>(b.data()), (std::streamsize)b.size());
    }
    const pid_t pid = fork();
    if (pid < 0) return 2;
    if (pid == 0) {
      std::fclose(stderr);  // the loaders' own error messages
      if (kind == "nrrd") {
        nrrd_loader l;
        volume_block v = l.load_file(scratch);
        (void)v;
      } else {
        hdre_loader l;
        image im = l.load_file(scratch);
        if (im.m_pixels.size() != (size_t)im.m_width * im.m_height * 4) _exit(3);
      }
      _exit(0);
    }
    int status = 0;
    waitpid(pid, &status, 0);
    if (WIFEXITED(status) && WEXITSTATUS(status) == 0) ++ok;
    else if (WIFEXITED(status) && WEXITSTATUS(status) == 1) ++rejected;
    else {
      std::fprintf(stderr, "iteration %ld: child status 0x%x (input kept in %s)\n", it, status, scratch);
      return 1;
    }
  }
  std::printf("%ld iterations, %ld accepted, %ld rejected\n", iters, ok, rejected);
  return 0;
}

// usage: fuzz_parsers <iterations> png|jpg|tf <file>...   |   fuzz_parsers <iterations> nrrd|hdr <scratch file> <file>...
int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const long iters = std::atol(argv[1]);
  const std::string kind = argv[2];
  if (kind == "nrrd" || kind == "hdr") {
    if (argc < 5) return 2;
    std::vector<std::vector<unsigned char>> seeds;
    for (int i = 4; i < argc; ++i) seeds.push_back(read_file(argv[i]));
    return fuzz_loader(kind, iters, seeds, argv[3]);
  }
  const bool png = kind == "png", jpg = kind == "jpg";
  std::vector<std::vector<unsigned char>> seeds;
  for (int i = 3; i < argc; ++i) seeds.push_back(read_file(argv[i]));
  long accepted = 0;
  for (long it = 0; it < iters; ++it) {
    std::vector<unsigned char> b = seeds[rnd() % seeds.size()];
    if (it >= (long)seeds.size()) mutate(b);  // the first rounds run the unmodified seeds
    if (jpg) {
      unsigned w = 0, h = 0;
      std::vector<unsigned char> rgba;
      std::string err;
      // keep decoded sizes bounded: find SOF and look at its dimensions
      bool huge = false;
      for (size_t i = 0; i + 9 < b.size(); ++i)
        if (b[i] == 0xFF && (b[i + 1] == 0xC0 || b[i + 1] == 0xC1 || b[i + 1] == 0xC2)) {
          const uint64_t hh = ((uint64_t)b[i + 5] << 8) | b[i + 6], ww = ((uint64_t)b[i + 7] << 8) | b[i + 8];
          if (ww * hh > (1u << 20)) huge = true;
        }
      if (huge) continue;
      if (jpeg_decode_rgba8(b, w, h, rgba, err)) {
        if (rgba.size() != (size_t)w * h * 4) { std::fprintf(stderr, "size mismatch\n"); return 1; }
        ++accepted;
      }
    } else if (png) {
      // keep decoded sizes bounded: a mutated IHDR may ask for gigabytes, which is legal but not what is tested here
      if (b.size() >= 24) {
        const uint64_t w = ((uint64_t)b[16] << 24) | (b[17] << 16) | (b[18] << 8) | b[19];
        const uint64_t h = ((uint64_t)b[20] << 24) | (b[21] << 16) | (b[22] << 8) | b[23];
        if (w * h > (1u << 22)) continue;
      }
      unsigned w = 0, h = 0;
      std::vector<unsigned char> rgba;
      std::string err;
      if (png_decode_rgba8(b, w, h, rgba, err)) {
        if (rgba.size() != (size_t)w * h * 4) { std::fprintf(stderr, "size mismatch\n"); return 1; }
        ++accepted;
      }
    } else {
      b.push_back(0);
      clwh_tf tf;
      if (clwh_tf_parse(reinterpret_cast<const char *>(b.data()), &tf) == CLWH_OK) {
        if (tf.n < 0 || tf.n > CLWH_TF_MAX_RULES) { std::fprintf(stderr, "rule count out of range\n"); return 1; }
        ++accepted;
      }
    }
  }
  std::printf("%ld iterations, %ld accepted\n", iters, accepted);
  return 0;
}
