// clwh_internal.hpp -- structures shared by the host runtime and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/clwh.h"

namespace clvr {

// ---- transfer function as a kernel argument (uniform -> SGPRs / scalar cache)
struct TfRuleDev {
  int32_t v_lo, v_hi, g_lo, g_hi;
  uint32_t flags;   // bit0 use_gradient, bit1 writes_color, bit2 terminal
  uint32_t color;   // r | g<<8 | b<<16 | a<<24
};
struct TfDev {
  int32_t n;
  int32_t uses_gradient;
  int32_t literal_gradient_taps;  // 1: always take the reference's 7-fetch step (test knob CLWH_TUNE_LITERAL_GRADIENT)
  int32_t opaque;                 // 1: classes come from a hiprtc-compiled is_event_gen (tf_jit.cpp); rules hold only colours
  int32_t border_class;           // class of the border texel (value 0) for gradient-free rule tables: 1 + index of the first rule that
                                  // contains 0, or 0.  read_imagei returns 0 outside the volume, and a position with a NaN coordinate
                                  // or a coordinate == dimension is not 'exited' (utility_ray.cl:112-117): it classifies as value 0
  TfRuleDev rules[CLWH_TF_MAX_RULES];
};

enum : uint32_t { TF_USE_GRADIENT = 1u, TF_WRITES_COLOR = 2u, TF_TERMINAL = 4u };

// ---- render pass arguments
// One hit record = one 64-byte line: what compute_light needs to start the sample's bounce paths.
struct HitRec {
  float origin[3];     // hit_information.origin (position of the primary Hit event)
  float direction[3];  // hit_information.direction (the camera ray's direction)
  float normal[3];     // -normalize(gradient) at the hit (ray_marching.cl:42)
  uint32_t color;      // transfer-function colour of the hit: r | g<<8 | b<<16 | roughness<<24
  int32_t entry_lo;    // voxel-cache entry (y-major), or -2 when the entry lies outside the cache
  int32_t entry_hi;
  uint32_t xy;         // global pixel: x | y<<16  (feeds the per-pixel RNG, utility_sampling.cl:41)
  uint32_t pslot;      // tile-major pixel slot of this rank (accumulation / scratch index)
  uint32_t pad[2];
};
static_assert(sizeof(HitRec) == 64, "HitRec must be one 64-byte line");

enum : uint32_t { PIX_HIT = 0x80000000u };  // pix_slot: PIX_HIT | hit index, else the miss colour (rgb)

struct RenderArgs {
  int32_t X, Y, Z;
  const uint2 *grec;       // bricked hit records {gx, gy, gz, class} (packed_volume.hpp)
  const uint8_t *stepb;    // bricked per-step bytes (packed_volume.hpp)
  const int16_t *volume_lin;  // the caller's images (x fastest): literal taps of the rare paths
  const int8_t *sdf_lin;
  // exit certificates (render_kernels.hip, certify_exit): eight bytes per macro cell of 16^3 voxels; byte o: the smallest SDF value in
  // the box a march from this cell in direction octant o crosses until it leaves the volume, 0 = the box is not free
  const uint8_t *macro;
  int32_t MNX, MNY, MNZ, macro_shift;  // cells per axis, log2 of the cell's edge in voxels
  int32_t cert_min_step;   // a march asks for a certificate once its next step is at least this long; 0 = certificates off
  int32_t cert_min_lanes;  // ... and the wave looks them up once this many lanes wait for one (launch_bounce sets it)
  int32_t NBX, NBY;
  const uint32_t *env;     // RGBA8 packed, row-major
  int32_t env_w, env_h;
  uint32_t *cache;         // voxel cache as 2 x u32 per entry
  int64_t cache_entries;   // entries that may be indexed
  uint32_t *frame;         // RGBA8 packed, row-major, frame_w x frame_h
  int32_t frame_w, frame_h;
  int32_t launch_w, launch_h;
  int32_t tiles_x, tiles_y, tiles_per_row;  // 8x8 pixel tiles; tiles_per_row = ceil(tiles_x / world)
  int32_t tile_rank, tile_world;
  float cam_pos[3];
  float cam_dir[3];
  int32_t mode;            // clwh_accum_mode
  float4 *accum;           // tile-major float4 per pixel slot (mode 1)
  unsigned long long *delta;  // mode 1 (and planned voxel-cache launches): this launch's packed sums per hit (see finish_item / k_commit)
  // planned voxel-cache launches (several seeds fused into one launch, clwh_render): the tokens of the launch are dealt out BEFORE it
  // runs -- grants[h] = how many of the launch's seeds hit h may trace (the first grants[h] of them) -- instead of three atomics per
  // sample on a cache entry that all the seeds and pixels of a voxel hammer at the same time; nullptr: the reference's own
  // token-per-sample protocol (utility.cl:20-36)
  const uint32_t *grants;
  uint32_t *pix_slot;      // tile-major, per pixel: PIX_HIT | hit index, or the miss colour
  HitRec *hits;            // compacted primary hits of this camera
  uint32_t *counters;      // [0] hits (k_primary), [2] fix-up records, [32*(q+1)] unit-queue heads, one per 128-B line
  uint32_t *fixups;        // 128-byte records of samples whose env lookup needs the exact route
  uint32_t fixup_capacity;
  uint32_t *sticky_flags;  // [0] fix-up overflow (outside the per-launch reset range: survives until the host reads it)
  uint32_t n_hits;         // host copy of counters[0], or an upper bound of it when n_hits_on_device
  uint32_t n_hits_estimate;  // the likely count (the count itself once known): picks the scheduling class of the launch
  int32_t n_hits_on_device;  // 1: kernels read the hit count from counters[0] (no host round trip after k_primary)
  int32_t shading;         // clwh_shading
  int64_t *hit_index_out;  // optional, row-major over launch_w x launch_h
  uint32_t *contrib_out;   // optional, row-major uint32[4] (single seed)
  uint32_t num_tile_slots; // tile slots of this rank
  int32_t n_seeds;
  int32_t seeds[CLWH_MAX_SEEDS];
  // scheduling knobs of k_bounce (defaults in clwh_runtime.hip; CLWH_TUNE_* override for experiments)
  int32_t step_min_lanes;    // keep stepping while at least this many lanes march
  int32_t refill_min_lanes;  // idle lanes fetch new items once this many are idle (64: only an empty wave refills)
  int32_t force_long_launch; // tests: schedule every launch like a long one (thresholds 16 / 16, exit certificates)
  int32_t bounce_rays;       // 2: long launches run k_bounce2, two rays per lane (CLWH_TUNE_BOUNCE_RAYS; the round-3 experiment)
  uint32_t bounce_max_blocks;  // persistent grid size (256-thread blocks)
  int32_t unit_group;          // chunks per queue group (see k_bounce refill)
  int32_t unit_block_log2;     // 2^n consecutive chunks go to the same queue
  int32_t unit_queues;         // number of unit queues (1..8)
  int32_t unit_affinity;       // 0: the wave's XCD picks its home queue (default); 1: wave number; 2: queue 0 (experiments)
  TfDev tf;
};

// ---- repack arguments (volume + sdf + TF -> step bytes + hit records)
struct RepackArgs {
  const int16_t *volume;
  const int8_t *sdf;
  int32_t X, Y, Z;
  int32_t NBX, NBY, NBZ;
  uint2 *grec;
  uint8_t *stepb;
  uint32_t *brick_min;     // per brick: 0 if a voxel may be an event or has a non-positive SDF value, else the smallest SDF value
  const uint8_t *cls_in;   // opaque TF: class byte per voxel (linear), computed by the JIT classifier
  TfDev tf;
};

// ---- SDF build arguments
struct SdfArgs {
  const int16_t *volume;
  int32_t X, Y, Z;
  int8_t *ping;
  int8_t *pong;
  int32_t max_iterations;
  int32_t iteration;
  int32_t *counters;   // [layer] write counts
  int32_t *done;       // [layer] early-out chain (fused build only; nullptr for the generic launch)
  int32_t *counter_out;  // generic launch: the caller's `add_buffer`
  const uint8_t *cls_in; // opaque TF: class byte per voxel (linear)
  TfDev tf;
};

// hiprtc fallback for TF source outside the rule grammar (tf_jit.cpp)
int tf_jit_compile(const char *user_source, std::vector<char> &code, std::string &log);
struct JitTf {
  std::string source;
  std::vector<char> code;
  hipModule_t module = nullptr;
  hipFunction_t classify = nullptr;
};

// host-side launchers implemented in the .hip files
hipError_t launch_repack(const RepackArgs &a, hipStream_t s);
// log2 of the macro cell's edge: 16 voxels up to 512^3, then growing with the volume so that the table (8 B per cell) stays at a
// few hundred KB, resident in every L2 -- at 2048^3 cells of 16^3 would make it 16 MiB and every look-up a miss of its own
inline int macro_cell_shift(int X, int Y, int Z, int forced) {
  if (forced >= 3 && forced <= 8) return forced;  // CLWH_TUNE_MACRO_SHIFT (tests, experiments; 3: a cell is one brick)
  // 16-voxel cells while the table stays L2-sized (512^3: 33^3 cells x 8 octants = 0.3 MB); beyond that 32-voxel cells up to 2048^3 (65^3
  // cells, 2.2 MB: measured 28.3-28.5 ms per launch against 29.2-29.3 with 64-voxel cells and 29.2-30.2 with 16-voxel cells), then larger
  auto cells = [&](int sh) { return (int64_t)((X >> sh) + 1) * ((Y >> sh) + 1) * ((Z >> sh) + 1); };
  int shift = 4;
  if (cells(4) > 40000) {
    shift = 5;
    while (shift < 8 && cells(shift) > 300000) ++shift;
  }
  return shift;
}
hipError_t launch_macro_table(const uint32_t *brick_min, int NBX, int NBY, int NBZ, uint8_t *macro, int X, int Y, int Z, int shift, hipStream_t s);
hipError_t launch_primary(const RenderArgs &a, hipStream_t s);
hipError_t launch_bounce(const RenderArgs &a, hipStream_t s);
hipError_t launch_env_fixup(const RenderArgs &a, hipStream_t s);
hipError_t launch_commit(const RenderArgs &a, hipStream_t s);
// planned voxel-cache launches: sort keys of the camera's hits (their cache entries), the token deal of one launch, the launch's sums into the cache
constexpr int64_t kVoxKeyInvalid = (int64_t)1 << 38;  // a hit whose entry lies outside the cache; kVoxKeyNone: no such hit (padding)
constexpr int64_t kVoxKeyNone = ((int64_t)1 << 38) + 1;
hipError_t launch_vox_keys(const RenderArgs &a, int64_t *keys, uint32_t *iota, uint32_t n, hipStream_t s);
hipError_t launch_vox_grant(const RenderArgs &a, const int64_t *sorted_keys, const uint32_t *order, uint32_t n, uint32_t *grants, hipStream_t s);
hipError_t launch_commit_voxel(const RenderArgs &a, hipStream_t s);
// stable radix sort of (entry, position) pairs on bits [0, end_bit) (exchange_kernels.hip: the one translation unit with rocPRIM)
hipError_t sort_entry_pairs(void *temp, size_t &temp_bytes, const int64_t *keys_in, int64_t *keys_out, const uint32_t *vals_in,
                            uint32_t *vals_out, size_t n, unsigned end_bit, hipStream_t s);
hipError_t launch_resolve(const RenderArgs &a, hipStream_t s);
hipError_t launch_ao(const RenderArgs &a, hipStream_t s);
hipError_t launch_accum_resolve(const RenderArgs &a, const float4 *accum_all, hipStream_t s);
hipError_t launch_accum_resolve_tiles(const RenderArgs &a, const float4 *accum, uint32_t *tiles_out, hipStream_t s);
hipError_t launch_frame_from_tiles(const RenderArgs &a, const uint32_t *tiles_all, hipStream_t s);
// ---- fused SDF build: one breadth-first layer over the active 8x8x8 tiles (sdf_kernels.hip)
struct SdfFrontArgs {
  int8_t *sdf;
  int32_t X, Y, Z;
  int32_t TX, TY, TZ;         // 8x8x8 tiles per axis
  const uint8_t *flags_cur;   // tiles that settled voxels in the previous layer (or hold |v| == 1 for layer 1)
  uint8_t *flags_next;        // tiles that settle voxels in this layer
  uint8_t *flags_clear;       // third buffer, zeroed here for the layer after next
  uint8_t *tile_done;         // 1: every voxel of the tile is settled, the tile is never visited again
  int32_t iteration;
  int32_t max_iterations;
  int32_t *counters;          // [i] != 0: layer i settled a voxel to a value < max_iterations; [0] != 0: some |v| == 1
};

// ---- fused SDF build, bit-parallel (sdf_kernels.hip): one bit per voxel, eight layers per launch
struct SdfBitArgs {
  int8_t *sdf;
  const uint32_t *ev;      // event bit per voxel, rows of WP 32-bit words
  const uint32_t *r_in;    // reached set after r0 layers
  uint32_t *r_out;         // reached set after r0 + steps layers (core rows of the active blocks)
  uint8_t *state;          // per block of sdfbit_block_grid: 0 empty, 1 some, 2 complete (just now), 3 complete in both buffers
  uint8_t *wake;           // per block: launch index + 1 for which a neighbour's reached voxels came within 8 voxels of its core
  int32_t launch;          // index of this launch (8 layers each)
  int32_t *presence;       // [D] != 0: some voxel lies D corner moves from the nearest seed; [0]: a seed exists
  int32_t X, Y, Z, WP;
  int32_t BX, BY, BZ, core_z;  // blocks of 64 x 48 x core_z voxels (sdfbit_block_grid)
  int32_t r0, steps;       // steps <= 8
  uint32_t *planes;        // seven bit planes of the layer index (8 launch + layer-in-launch + 1, 1..127) of every voxel a layer reached,
                           // each tiled like the reached sets; OR-ed in by the launches, expanded to bytes ONCE by k_sdfbit_expand
  size_t plane_words;      // words per plane
  uint32_t *list;          // blocks that can change in this launch (k_sdfbit_list)
  uint32_t *list_count;    // their number; list_head: the persistent grid's queue position
  uint32_t *list_head;
#ifdef CLVR_SDFBIT_TIMING
  unsigned long long *timing;  // tools/ builds only: per-phase sums of wall_clock64 ticks over all regions
#endif
};
void sdfbit_block_grid(int X, int Y, int Z, int waves, int32_t *BX, int32_t *BY, int32_t *BZ, int32_t *core_z);  // blocks of 64 x 48 x (4 waves - 16) voxels
hipError_t launch_sdfbit_events(const SdfArgs &a, uint32_t *ev, int32_t WP, hipStream_t s);
hipError_t launch_sdfbit_seed(const SdfBitArgs &a, hipStream_t s);                                // seeds into a.r_out
hipError_t launch_sdfbit_expand(const SdfBitArgs &a, const uint32_t *reached, int32_t max_iterations, hipStream_t s);  // bit planes + final reached set -> a.sdf
hipError_t launch_sdfbit_state(const SdfBitArgs &a, hipStream_t s);                              // block states of a.r_in
hipError_t launch_sdfbit_layers(const SdfBitArgs &a, int waves, unsigned grid_blocks, bool rec_in_lds, hipStream_t s);

hipError_t launch_sdf_base(const SdfArgs &a, hipStream_t s);
hipError_t launch_sdf_base_front(const SdfArgs &a, uint8_t *flags, int32_t TX, int32_t TY, hipStream_t s);
hipError_t launch_sdf_front(const SdfFrontArgs &a, hipStream_t s);
hipError_t launch_sdf_layer(const SdfArgs &a, hipStream_t s);
hipError_t launch_fetch_stats(const int16_t *vol, int X, int Y, int Z, int32_t *stats, hipStream_t s);
hipError_t launch_tf_sort_values(const int16_t *vol, int X, int Y, int Z, uint32_t *frame, int width, int height,
                                 float min_v, float max_v, float min_g, float max_g, hipStream_t s);
hipError_t launch_tf_flush_color_frame(uint32_t *color_frame, int fw, int fh, const int32_t *frame, const int32_t *lookup,
                                       int lookup_len, hipStream_t s);
hipError_t launch_bilateral_filter(const int16_t *src, int X, int Y, int Z, int16_t *dst, const float *weights, hipStream_t s);
hipError_t launch_apply_clip(const int16_t *src, int SX, int SY, int SZ, int16_t *dst, int DX, int DY, int DZ,
                             const uint32_t *start, const uint32_t *len, hipStream_t s);

// ---- derived scene data (step bytes + hit records + per-brick minima + exit-certificate table), ONE copy per device however many
// contexts (frame lanes, callers) render the same (volume content, SDF content, transfer function): contexts hold it by
// shared_ptr and find it in a process-wide registry (clwh_runtime.hip); the memory goes when the last context lets go of it.
struct PackedScene {
  int device = 0;
  uint8_t *data = nullptr;
  size_t bytes = 0;
  const void *vol = nullptr, *sdf = nullptr;
  uint64_t vol_ver = 0, sdf_ver = 0;
  TfDev tf{};
  std::string tf_identity;   // opaque (hiprtc) transfer functions: the source text -- two sources may share a palette
  int32_t macro_shift = 0;
  uint64_t generation = 0;   // process-wide unique id of this content (part of the primary-hit key)
  hipEvent_t ready = nullptr;  // recorded on the building stream after the last build kernel; adopters make their stream wait for it
  bool stale = false;        // clwh_ctx_invalidate_derived: nobody adopts it any more
  ~PackedScene();
};

// content version of a device allocation, shared by every clwh_mem that names the same device pointer (the owner and all
// wraps): a push, a rebuild or clwh_mem_mark_dirty through ANY of them is seen by all
struct VersionCell {
  std::atomic<uint64_t> v{0};
};

}  // namespace clvr

// ---- opaque handle layouts (host only)
struct clwh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t handoff_event = nullptr;  // clwh_ctx_acquire_from / clwh_ctx_release_to
  // per-camera primary hits (derived data, rebuilt when the key below changes)
  uint32_t *pix_slot = nullptr;
  size_t pix_slot_bytes = 0;
  clvr::HitRec *hits = nullptr;
  size_t hits_bytes = 0;
  uint32_t *render_counters = nullptr;  // kRenderCounters x u32 on the device
  static constexpr size_t kRenderCounters = 32 * 9;
  uint32_t *fixups = nullptr;
  size_t fixups_bytes = 0;
  unsigned long long *delta = nullptr;
  size_t delta_bytes = 0;
  // planned voxel-cache launches: the camera's hits grouped by voxel (sorted once per camera), this launch's grants
  uint8_t *vox_plan = nullptr;     // keys_in | keys_sorted (int64 each) | iota | order | grants (u32 each), n_capacity elements each
  size_t vox_plan_bytes = 0;
  void *vox_temp = nullptr;
  size_t vox_temp_bytes = 0;
  bool vox_plan_valid = false;
  uint32_t vox_plan_n = 0;         // elements sorted (the hit count, or its bound when the plan was made)
  // hiprtc-compiled transfer functions, by source text; and the class bytes of the current (volume, source)
  std::map<std::string, std::shared_ptr<clvr::JitTf>> jit_cache;
  uint8_t *jit_cls = nullptr;
  size_t jit_cls_bytes = 0;
  unsigned long long *jit_palette = nullptr;  // CLWH_TF_MAX_RULES keys + 1 error word
  const void *jit_vol = nullptr;
  uint64_t jit_vol_ver = 0;
  std::string jit_source;
  clvr::TfDev jit_tf{};
  bool fixup_overflow_pending = false;
  uint32_t *sticky_flags = nullptr;  // [0] fix-up buffer overflow: set by kernels, cleared only when the host has read it
  // measured on MI355X (round 1, git history: profiles/r01_tune_*.txt): a wave that runs its 64 samples to completion with
  // steps and events in separate wave-wide phases beats mid-flight refills (12.5 vs 7.6 Gsamples/s)
  int32_t tune_step_min_lanes = 0;    // 0: chosen per launch (launch_bounce)
  int32_t tune_refill_min_lanes = 0;  // 0: chosen per launch (launch_bounce)
  int32_t tune_macro_shift = 0;       // CLWH_TUNE_MACRO_SHIFT: 4..8 forces the macro cell's edge to 2^n voxels (0: by volume size)
  int32_t tune_bounce_rays = 1;       // CLWH_TUNE_BOUNCE_RAYS=2: k_bounce2 for long launches
  int32_t tune_force_long_launch = 0; // CLWH_TUNE_LONG_LAUNCH=1: every launch is scheduled like a long one (the parity tests use it)
  int32_t tune_literal_gradient = 0;
  int32_t tune_unit_block_log2 = 4;
  int32_t tune_unit_group = 1, tune_unit_affinity = 0, tune_unit_queues = 8;
  int32_t tune_cert_min_step = -1;  // CLWH_TUNE_CERT: 0 = exit certificates off; -1 = by volume size (12 at 512^3, 24 at 1024^3, 48 at 2048^3:
                                    // the best of the sweeps in profiles/r02_sweep_k_bounce_lds_state.txt)
  uint32_t tune_bounce_max_blocks = 2048;  // CLWH_TUNE_BLOCKS
  bool primary_valid = false;
  uint32_t primary_n_hits = 0;
  bool primary_n_hits_known = false;  // false: the count of this camera's hits is only on the device so far
  // the count travels to the host behind the camera's k_primary without anybody waiting for it: a 4-byte copy into page-locked
  // memory + an event; later launches of the same camera pick it up once the event has completed (hipEventQuery)
  uint32_t *host_n_hits = nullptr;
  hipEvent_t n_hits_event = nullptr;
  bool n_hits_in_flight = false;
  uint32_t last_known_n_hits = 0;     // of any earlier camera of this context (0: none yet): sizes work buffers while the count is unknown
  struct PrimaryKey {
    float cam_pos[3], cam_dir[3];
    int32_t frame_w, frame_h, launch_w, launch_h, tile_rank, tile_world;
    int64_t cache_entries;
    int32_t mode, shading;
    uint64_t packed_generation;
    // miss pixels keep the environment colour of their camera ray: the env map's identity and content are part of the key
    const void *env;
    uint64_t env_version;
    int32_t env_w, env_h;
  } primary_key{};
  float *bilateral_weights = nullptr;  // 13 x 17 tap weights of the bilateral volume filter (built on first use)
  int32_t *sdf_counters = nullptr;  // 160 ints: settled voxels per layer
  uint8_t *sdf_flags = nullptr;     // 4 x tiles bytes (current / next / being cleared / done)
  size_t sdf_flags_bytes = 0;
  uint32_t *sdf_bits = nullptr;     // bit-parallel build: event bits, two reached-set buffers, block states
  size_t sdf_bits_bytes = 0;
  int32_t tune_sdfbit_waves = 8;    // CLWH_TUNE_SDFBIT_WAVES: 8 or 16 waves per block of the bit-parallel build
  int32_t tune_sdfbit_grid = 512;   // CLWH_TUNE_SDFBIT_GRID: its persistent grid
  int32_t tune_sdfbit_rec_lds = 0;  // CLWH_TUNE_SDFBIT_REC=lds: the layer records in LDS, three blocks of eight waves per CU (grid x 3 / 2)
  int32_t tune_sdf_front = 0;       // CLWH_TUNE_SDF=front: the byte-front build (one launch per layer) instead of the bit-parallel one
  // derived packed volume: hit records (8 B per voxel of the brick grid), the step bytes (1 B), the per-brick minima (4 B per
  // brick), the macro-cell table -- shared with every other context of the device that renders the same scene
  std::shared_ptr<clvr::PackedScene> scene;

  // timing: one HIP event pair per clwh_render, recorded on the context's stream around the
  // dominant kernel and read back (without a sync per pass) by clwh_ctx_timing_read
  bool timing = false;
  std::vector<hipEvent_t> ev_begin, ev_end;
  std::vector<int> ev_which;  // enum clwh_timer of each pair
  size_t ev_used = 0;
};

struct clwh_mem {
  clwh_ctx *ctx = nullptr;
  void *dptr = nullptr;
  size_t bytes = 0;
  bool owned = false;
  bool is_image = false;
  size_t dims[3] = {1, 1, 1};
  int channels = 1;
  int elem_kind = CLWH_ELEM_U8;
  int flags = 0;
  std::shared_ptr<clvr::VersionCell> cell;  // shared with every other clwh_mem of the same device pointer
  uint64_t version() const { return cell ? cell->v.load(std::memory_order_relaxed) : 0; }
};

// a new content version for the object's device memory (seen through every clwh_mem that names the same pointer)
void clwh_touch(clwh_mem *m);

enum clwh_kernel_id {
  CLWH_K_EMPTY = 0,
  CLWH_K_RENDER,
  CLWH_K_SDF_BASE,
  CLWH_K_SDF_LAYER,
  CLWH_K_BUFFER_RESET,
  CLWH_K_FETCH_STATS,
  CLWH_K_APPLY_CLIP,
  CLWH_K_TF_SORT_VALUES,
  CLWH_K_TF_FLUSH_COLOR_FRAME,
  CLWH_K_BILATERAL_FILTER
};

struct clwh_kernel {
  clwh_ctx *ctx = nullptr;
  int id = CLWH_K_EMPTY;
  clwh_tf tf{};
  bool has_tf = false;
  std::shared_ptr<clvr::JitTf> jit;  // set when the source is outside the rule grammar
};
