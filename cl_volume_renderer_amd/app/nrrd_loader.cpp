#include "nrrd_loader.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>

namespace {

[[noreturn]] void fail(const std::string &what) {
  std::cerr << "Error: NRRD file " << what << "\n";
  std::exit(EXIT_FAILURE);
}

std::vector<std::string> split(const std::string &s, char delimiter) {
  std::vector<std::string> out;
  std::stringstream ss(s);
  std::string item;
  while (std::getline(ss, item, delimiter))
    if (!item.empty()) out.push_back(item);
  return out;
}

std::string trim(const std::string &s) {
  const auto b = s.find_first_not_of(" \t\r"), e = s.find_last_not_of(" \t\r");
  return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
}

// numbers of the header: a malformed one is a format violation (fail hard), not an uncaught exception
unsigned parse_size(const std::string &word) {
  if (word.empty() || word.find_first_not_of("0123456789") != std::string::npos || word.size() > 9) fail("does not declare sizes correctly.");
  return (unsigned)std::stoul(word);
}
float parse_float(const std::string &word) {
  char *end = nullptr;
  const float v = std::strtof(word.c_str(), &end);
  if (end == word.c_str() || *end != '\0') fail("does not declare space direction correctly.");
  return v;
}

// "(a,b,c)" -> component `which`
float vector_component(const std::string &token, int which) {
  std::string inner = token;
  if (!inner.empty() && inner.front() == '(') inner.erase(0, 1);
  if (!inner.empty() && inner.back() == ')') inner.pop_back();
  const auto parts = split(inner, ',');
  if ((int)parts.size() != 3) fail("does not declare space direction correctly.");
  return parse_float(parts[which]);
}

}  // namespace

nrrd_header nrrd_loader::load_header(const std::string &path) {
  std::ifstream in(path, std::ios::in | std::ios::binary);
  if (in.fail()) {
    std::cerr << "Error, failed to open .nrrd file: " << path << '\n';
    std::exit(1);
  }
  nrrd_header h;
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty()) break;                                      // blank line: payload follows
    if (line[0] == '#' || line.compare(0, 4, "NRRD") == 0) continue;  // comments, magic
    const auto colon = line.find(':');
    if (colon == std::string::npos) fail("does not declare tags correctly.");
    const std::string tag = line.substr(0, colon);
    const std::string value = trim(line.substr(colon + 1));
    const auto words = split(value, ' ');
    if (words.empty()) fail("does not declare tags correctly.");
    if (tag == "type") {
      if (words[0] != "short") fail("not using short as type.");
    } else if (tag == "encoding") {
      if (words[0] == "gzip") h.raw = false;
      else if (words[0] == "raw") h.raw = true;
      else fail("not using gzip compression or raw.");
    } else if (tag == "endian") {
      if (words[0] != "little") fail("not using little endian format.");
    } else if (tag == "dimension") {
      if (words[0] != "3") fail("not using dimension of 3.");
    } else if (tag == "sizes") {
      if (words.size() != 3) fail("does not declare sizes correctly.");
      h.x = parse_size(words[0]);
      h.y = parse_size(words[1]);
      h.z = parse_size(words[2]);
    } else if (tag == "space directions") {
      if (words.size() != 3) fail("does not declare space direction correctly.");
      const float sx = vector_component(words[0], 0), sy = vector_component(words[1], 1), sz = vector_component(words[2], 2);
      h.x_voxel_size = 1.f;  // sizes are kept relative to x, as the reference does
      h.y_voxel_size = sy / sx;
      h.z_voxel_size = sz / sx;
    }
  }
  h.data_start = (std::uint64_t)in.tellg();
  in.seekg(0, std::ios::end);
  h.data_end = (std::uint64_t)in.tellg();
  return h;
}

volume_block nrrd_loader::load_file(const std::string path) {
  const nrrd_header h = load_header(path);
  const std::uint64_t voxels = (std::uint64_t)h.x * h.y * h.z;
  if (voxels == 0) fail("does not declare sizes correctly.");
  const std::uint64_t want_bytes = voxels * sizeof(short);
  // a corrupt header must not turn into a terabyte allocation: deflate expands at most 1032:1
  const std::uint64_t payload = h.data_end - h.data_start;
  if (h.raw ? want_bytes > payload : want_bytes / 1032u > payload + 64u) fail("declares sizes its payload cannot hold.");
  std::vector<short> data(voxels);

  std::ifstream in(path, std::ios::in | std::ios::binary);
  in.seekg((std::streamoff)h.data_start);
  if (h.raw) {
    in.read(reinterpret_cast<char *>(data.data()), (std::streamsize)std::min<std::uint64_t>(want_bytes, h.data_end - h.data_start));
  } else {
    // streaming inflate: bounded input chunks, output straight into the voxel vector (window bits
    // 15 + 32: zlib or gzip wrapper detected automatically, as in the reference)
    z_stream zs{};
    if (inflateInit2(&zs, 15 + 32) != Z_OK) fail("could not be inflated.");
    std::vector<unsigned char> chunk(1u << 22);
    unsigned char *out = reinterpret_cast<unsigned char *>(data.data());
    std::uint64_t produced = 0;
    int status = Z_OK;
    while (status != Z_STREAM_END && produced < want_bytes) {
      in.read(reinterpret_cast<char *>(chunk.data()), (std::streamsize)chunk.size());
      const std::streamsize got = in.gcount();
      if (got <= 0) break;
      zs.next_in = chunk.data();
      zs.avail_in = (uInt)got;
      while (zs.avail_in > 0 && status != Z_STREAM_END && produced < want_bytes) {
        const std::uint64_t room = std::min<std::uint64_t>(want_bytes - produced, 1u << 30);
        zs.next_out = out + produced;
        zs.avail_out = (uInt)room;
        status = inflate(&zs, Z_NO_FLUSH);
        if (status != Z_OK && status != Z_STREAM_END) {
          inflateEnd(&zs);
          fail("could not be inflated.");
        }
        produced += room - zs.avail_out;
      }
    }
    inflateEnd(&zs);
  }
  return volume_block(std::move(data), h.x, h.y, h.z, h.x_voxel_size, h.y_voxel_size, h.z_voxel_size);
}
