// host_c_api.cpp -- a small extern "C" facade over the C++ host mirror (renderer, reference_volume,
// env_map, signed_distance_field) so that pytest can drive the very call sequence the SDL/ImGui
// application performs (reference app/ui.cpp:170-199, 296) without SDL.  Test / tooling entry points;
// the product ABI is include/clwh.h.
#include <cstring>
#include <memory>
#include <string>

#include "hdre_loader.hpp"
#include "nrrd_loader.hpp"
#include "renderer.hpp"
#include "tf_part.hpp"

struct clvr_host {
  clw_context ctx;
  renderer rend;
  std::unique_ptr<volume_block> block;
  std::unique_ptr<reference_volume> rv;
  std::unique_ptr<env_map> emap;
  ui_state state;
  clvr_host() : rend(ctx), state{"", true, 0, 0, Position3D(0, 0, 0), {0.f, 0.f}, true} {}
};

extern "C" {

clvr_host *clvr_host_create(void) { return new clvr_host(); }
void clvr_host_destroy(clvr_host *h) { delete h; }

// nrrd_loader::load_file + reference_volume + hdre_loader + env_map + image_set  (ui.cpp:182-194)
void clvr_host_load(clvr_host *h, const short *voxels, unsigned X, unsigned Y, unsigned Z, const unsigned char *env_rgba,
                    unsigned env_w, unsigned env_h) {
  h->rv.reset();  // release the old device images before allocating the new ones
  h->block.reset(new volume_block(std::vector<short>(voxels, voxels + (size_t)X * Y * Z), X, Y, Z, 1.f, 1.f, 1.f));
  h->rv.reset(new reference_volume(h->ctx, h->block.get()));
  h->rv->set_value_clip({-2000, 3000});
  h->rv->set_gradient_clip({0, 4000});
  image em;
  em.m_pixels.assign(env_rgba, env_rgba + (size_t)env_w * env_h * 4);
  em.m_width = env_w;
  em.m_height = env_h;
  h->emap.reset(new env_map(h->ctx, em));
  h->rend.image_set(h->rv.get(), h->emap.get());
}

// flush_tf + flush_changes  (ui.cpp:195-197)
void clvr_host_flush(clvr_host *h, const char *cl_code) {
  h->rend.next_event_code_set(cl_code);
  h->rend.flush_changes();
}

// one iteration of the ui::run loop body that matters here (ui.cpp:296)
const void *clvr_host_render_frame(clvr_host *h, const float pos[3], const float look[2], int width, int height,
                                   int cam_changed, int *frame_changed) {
  h->state.position = Position3D(pos[0], pos[1], pos[2]);
  h->state.direction_look[0] = look[0];
  h->state.direction_look[1] = look[1];
  h->state.width = width;
  h->state.height = height;
  h->state.cam_changed = cam_changed != 0;
  bool changed = false;
  void *p = h->rend.render_frame(h->state, changed);
  if (frame_changed) *frame_changed = changed ? 1 : 0;
  return p;
}

// the same iteration with the display hand-off instead of the readback: the frame goes into `device_frame`
// (frame_w x frame_h RGBA8 device memory of the caller, read on `stream`), `passes` seeds in one launch
void clvr_host_render_frame_device(clvr_host *h, const float pos[3], const float look[2], int width, int height,
                                   int cam_changed, void *device_frame, unsigned frame_w, unsigned frame_h, void *stream,
                                   int passes) {
  h->state.position = Position3D(pos[0], pos[1], pos[2]);
  h->state.direction_look[0] = look[0];
  h->state.direction_look[1] = look[1];
  h->state.width = width;
  h->state.height = height;
  h->state.cam_changed = cam_changed != 0;
  h->state.path_changed = true;
  clw_foreign_memory target(h->ctx, device_frame, frame_w, frame_h, stream);
  h->rend.render_frame_device(h->state, target, passes);
}

size_t clvr_host_cache_len(clvr_host *h) { return h->rend.voxel_cache().size(); }
void clvr_host_pull_cache(clvr_host *h, unsigned short *out) {
  auto &c = h->rend.voxel_cache();
  c.pull();
  std::memcpy(out, &c[0], c.size() * sizeof(unsigned short));
}
size_t clvr_host_sdf_len(clvr_host *h) { return h->rend.distance_field().get_sdf_buffer().size(); }
void clvr_host_pull_sdf(clvr_host *h, signed char *out) {
  auto &s = h->rend.distance_field().get_sdf_buffer();
  s.pull();
  std::memcpy(out, &s[0], s.size());
}
// reference_volume statistics (fetch_stats at construction) and clipping (apply_clip)
void clvr_host_volume_stats(clvr_host *h, float out[4]) {
  const Volume_Stats s = h->rv->get_volume_stats();
  out[0] = s.min_v; out[1] = s.max_v; out[2] = s.min_g; out[3] = s.max_g;
}
void clvr_host_set_clipping(clvr_host *h, const unsigned lo[3], const unsigned hi[3]) {
  h->rv->set_clipping({lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]});
}
// the "Apply Filter" checkbox (ui.cpp:274-275)
void clvr_host_filter(clvr_host *h) { h->rv->filter(); }
// _create_tf (ui.cpp:151-158): the transfer-function editor's histogram texture, RGBA8, width x height
const void *clvr_host_render_tf(clvr_host *h, unsigned width, unsigned height) { return h->rend.render_tf(width, height); }
// nrrd_loader::load_file probe (no device involved): dims, voxel count, sum of all voxels
long long clvr_host_nrrd_probe(const char *path, unsigned dims[3], long long *checksum) {
  nrrd_loader loader;
  volume_block b = loader.load_file(path);
  dims[0] = b.m_voxel_count_x; dims[1] = b.m_voxel_count_y; dims[2] = b.m_voxel_count_z;
  long long sum = 0;
  for (short v : b.m_voxels) sum += v;
  *checksum = sum;
  return (long long)b.m_voxels.size();
}
// nrrd_loader::load_file, whole result (no device involved): counts, voxel sizes, voxels
long long clvr_host_nrrd_load(const char *path, unsigned counts[3], float sizes[3], short *out, long long cap_voxels) {
  nrrd_loader loader;
  volume_block b = loader.load_file(path);
  counts[0] = b.m_voxel_count_x; counts[1] = b.m_voxel_count_y; counts[2] = b.m_voxel_count_z;
  sizes[0] = b.m_voxel_size_x; sizes[1] = b.m_voxel_size_y; sizes[2] = b.m_voxel_size_z;
  const long long n = (long long)b.m_voxels.size();
  if (n > cap_voxels) return -n;
  std::memcpy(out, b.m_voxels.data(), (size_t)n * sizeof(short));
  return n;
}
// hdre_loader::load_file probe (no device involved): decodes into `out` (w*h*4 bytes) when it is large enough
long long clvr_host_hdr_probe(const char *path, unsigned dims[2], unsigned char *out, long long out_bytes) {
  hdre_loader loader;
  image im = loader.load_file(path);
  dims[0] = im.m_width; dims[1] = im.m_height;
  if ((long long)im.m_pixels.size() <= out_bytes) std::memcpy(out, im.m_pixels.data(), im.m_pixels.size());
  return (long long)im.m_pixels.size();
}
// ui::flush_tf string for a list of rectangles {min_v, max_v, min_g, max_g, r, g, b, a} (no device involved)
long long clvr_host_tf_source(const float *rects, int n, const float stats[4], char *out, long long out_bytes) {
  std::vector<tf_selection *> sel;
  for (int i = 0; i < n; ++i) {
    auto *r = new tf_rect_selection((unsigned)i, rects[i * 8 + 0], rects[i * 8 + 1], rects[i * 8 + 2], rects[i * 8 + 3]);
    for (int c = 0; c < 4; ++c) r->color[c] = rects[i * 8 + 4 + c];
    sel.push_back(r);
  }
  Volume_Stats st;
  st.min_v = stats[0]; st.max_v = stats[1]; st.min_g = stats[2]; st.max_g = stats[3];
  const std::string code = tf_generate_source(st, sel);
  for (auto *s : sel) delete s;
  if ((long long)code.size() + 1 <= out_bytes) std::memcpy(out, code.c_str(), code.size() + 1);
  return (long long)code.size();
}
int clvr_host_sdf_layers(clvr_host *h) { return h->rend.distance_field().layers(); }
void clvr_host_camera_direction(float alpha, float beta, float out[3]) {
  Position3D v(alpha, beta, 0.0, {1.0, 0.0, 0.0});
  out[0] = v.val[0]; out[1] = v.val[1]; out[2] = v.val[2];
}

}  // extern "C"
