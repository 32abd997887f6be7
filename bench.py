#!/usr/bin/env python3
"""bench.py -- headline benchmark of the volumetric path-tracing hot path on MI355X.

Metric (BASELINE.json): Msamples/s = width x height x spp / seconds on configs[1]: 512^3 volume +
4096x2048 env map, 1920x1080 frame, 64 spp (one `render` pass = one sample per pixel, reference
app/renderer.cpp:131-158), synthetic seeded inputs.

A STEP is one whole frame job of the config: the first frame after a camera move -- primary hits of the
camera (k_primary), `spp` render passes with fresh glibc-rand() seeds in launches of 64 passes each
(k_bounce + k_env_fixup + k_commit), resolve [N > 1: of the rank's own tiles, then ONE RCCL all-gather of RGBA8 tiles]
to the RGBA8 frame.  `--steps K` times exactly K such jobs between barrier + synchronize brackets; because K
jobs of the headline config take only K x 5.5 ms, the K-step region is repeated until at least
MIN_TIMED_SECONDS have been measured and the MEDIAN region is reported (every region's time is the max over
ranks), so the number does not depend on --steps and does not rest on a 2 ms measurement.  Inputs (volume,
SDF, env map, step bytes and hit records) are resident in HBM before the clock starts; the frame stays in HBM.
Consecutive frame jobs alternate between two HIP streams (--frames-in-flight, each stream with its own
accumulation and frame buffers): every persistent launch ends with ~0.3 ms in which its last waves wait for
their longest samples, and the next frame's primary hits and first waves fill the GPU meanwhile (+4 % on one
GPU).  The per-kernel durations of the roofline are measured in a separate region with ONE frame at a time.

`--gpus N` starts its own N ranks (torch.distributed.run as a child process, before this process touches a
GPU) unless it already runs under a launcher; the ranks share ONE frame as interleaved 8x8 image tiles
(strong scaling).  On a box with fewer than N GPUs the same code path is rehearsed on one GPU over gloo.

The JSON line also carries
  roofline       the dominant kernel k_bounce against the 8 TB/s HBM peak, counted with the bytes k_bounce
                 itself performs (oracle counters of the bounce phase only); per-kernel figures and the
                 SURVEY 8d contract figure (whole path, whole job) beside it;
  cpu_baseline   the CPU oracle (a port, not POCL) timed on this host on a bounded sample of the same
                 workload (rank 0, N=1 only);
  end_to_end     the same jobs including the final blocking frame readback (SURVEY 8d metric ii);
  drop_in_path   renderer::render_frame x 64 through the C++ host mirror (one pass per launch + frame.pull(),
                 the reference's own call pattern), and the readback-free / batched variants beside it;
  hit_samples_per_sec, frame_filling_view
                 what the headline hides: 84 % of the default view's pixels miss the volume and cost nothing per pass.  The
                 traced (hit pixel, pass) samples per second of the headline job, and the same job from a pose whose frame the
                 volume fills (scene.close_camera: 78 % hit pixels) with its own value and k_bounce roofline;
  reference_exact_mode
                 the reference's own accumulation (world-space voxel cache, 256-token cap): one pass per launch and 64 passes
                 fused into one launch;
  tolerance_vs_reference_exact
                 how far the headline's image-space frame is from the reference-exact frame of the same job (RGBA8, hit pixels).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MIN_TIMED_SECONDS = 0.5    # the K-step region is repeated until this much has been timed
MAX_REGIONS = 60

PRESETS = {
    # BASELINE.json configs[1..4]; the launch shape (64 passes per launch) is the same for every one
    2: dict(volume=512, width=1920, height=1080, spp=64, tf="default", steps=40, warmup=3),
    3: dict(volume=512, width=1920, height=1080, spp=256, tf="gradient", steps=10, warmup=1),
    4: dict(volume=2048, width=3840, height=2160, spp=256, tf="default", steps=3, warmup=1),
    5: dict(volume=512, width=3840, height=2160, spp=1024, tf="default", steps=3, warmup=1),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_threads():
    """Threads for the CPU baseline: the cores this process may really use (affinity, cgroup quota),
    capped at the 16-core share a one-GPU box grants."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("BENCH_CPU_THREADS", "16"))))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """Start n ranks of this script under torch.distributed.run and exit with the child's code.  Called before
    anything in this process has touched a GPU (a process that has initialised the GPU must not exec / fork
    GPU children on this pool); device_count() does not initialise it."""
    import torch

    env = dict(os.environ)
    have = torch.cuda.device_count()
    if have < n:
        # rehearsal: every rank on GPU 0, collectives over gloo staged through the host (functional check only)
        env["BENCH_DIST_BACKEND"] = "gloo"
        env["BENCH_ALL_ON_DEVICE"] = "0"
        log("[bench] %d GPU(s) visible, %d ranks requested: REHEARSAL on one GPU over gloo" % (have, n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    # rank 0 prints the one JSON line; whatever else the ranks' libraries write to stdout (gloo's connection banner in a
    # rehearsal) goes to stderr so that stdout stays that single line
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:
        if line.lstrip().startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
        elif line.strip():
            sys.stderr.write(line)
    raise SystemExit(child.wait())


# ---------------------------------------------------------------------------------------------------------------
# algorithmic bytes: texel sizes from SURVEY 8d, counts from the instrumented oracle


def bytes_contract_image_space(c):
    """SURVEY 8d's formula for the whole path, image-space accumulation: per sample
    N_sdf + 2 N_vol + 4 N_env + 8 N_add (ONE packed 64-bit atomic per granted sample) + 4 (frame write)."""
    return c["n_sdf"] + 2 * c["n_vol"] + 4 * c["n_env"] + 8 * c["n_add"] + 4 * c["samples"]


def bytes_primary(c):
    """what k_primary performs, once per camera: the primary march, the hit's 6-tap normal, the miss pixels'
    environment texel, and its per-pixel result (4 B)"""
    return c["n_sdf_primary"] + 2 * c["n_vol_primary"] + 4 * c["n_env_primary"] + 4 * c["samples"]


def bytes_bounce(c, tf_reads_gradient=False):
    """what k_bounce performs per pass: the distribution rays' marches, their normals, the environment texels of
    the rays that leave, one 8-byte atomic per granted sample.  tf_reads_gradient: SURVEY 8d's formula credits a transfer
    function that reads `gradient` with seven volume texels per classified step (the value and six taps); k_bounce reads
    the class baked at repack time instead, so it is credited like any other table -- one volume texel per classified
    step plus the six taps of every Hit's normal ("every kernel is credited only with what it executes")"""
    n_vol = c["n_vol"] - c["n_vol_primary"]
    if tf_reads_gradient:
        n_vol = (n_vol - 6 * c["n_hit_bounce"]) / 7.0 + 6 * c["n_hit_bounce"]
    return ((c["n_sdf"] - c["n_sdf_primary"]) + 2 * n_vol + 4 * (c["n_env"] - c["n_env_primary"]) + 8 * c["n_add"])


def bytes_bounce_executed(c):
    """what k_bounce really fetches per pass, at texel granularity: ONE step byte per march step (the class bit and the next
    step's SDF value share it), one 8-byte hit record per secondary Hit (gradient + class; the reference's colour + six taps),
    one env texel per exit, one 8-byte atomic per granted sample.  For a transfer function that reads `gradient` SURVEY 8d's
    formula credits seven volume texels per step; the kernel reads the class byte baked at repack time instead."""
    return ((c["n_sdf"] - c["n_sdf_primary"]) + 8 * c["n_hit_bounce"] + 4 * (c["n_env"] - c["n_env_primary"]) + 8 * c["n_add"])


def measured_traffic(key):
    """HBM-side bytes per launch of a kernel from the rocprofv3 PMC passes committed under profiles/ (bench.py
    cannot run the profiler on itself): {workload key: {kernel: {...}}} in profiles/r03_traffic.json (round 2's file for shapes not profiled again)."""
    for name in ("r03_traffic.json", "r02_traffic.json"):  # the newest profile that holds this launch shape
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            entry = json.load(open(path)).get(key)
            if entry:
                return entry
    return None


# ---------------------------------------------------------------------------------------------------------------
# secondary measurements (rank 0, N = 1)


def run_dropin(vol, env, tf_source, pos, torch):
    """The reference application's own call pattern through the C++ host mirror (libclvr_host.so, the classes of
    app/renderer.hpp over include/clw_*.hpp): ui::run's loop body is `render_frame(state, changed)` = ONE pass per
    launch in the voxel-cache mode + a blocking frame.pull() (app/renderer.cpp:131-158), at the application's own
    2048x1024 frame (common_defines.hpp:3-4).  Beside it: the same passes with the display hand-off instead of the
    readback (renderer::render_frame_device), one pass and eight passes per call."""
    import ctypes as C

    L = C.CDLL(os.path.join(ROOT, "cl_volume_renderer_amd", "libclvr_host.so"))
    L.clvr_host_create.restype = C.c_void_p
    L.clvr_host_destroy.argtypes = [C.c_void_p]
    L.clvr_host_load.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_uint, C.c_uint]
    L.clvr_host_flush.argtypes = [C.c_void_p, C.c_char_p]
    fpp = C.POINTER(C.c_float)
    L.clvr_host_render_frame.restype = C.c_void_p
    L.clvr_host_render_frame.argtypes = [C.c_void_p, fpp, fpp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.clvr_host_render_frame_device.argtypes = [C.c_void_p, fpp, fpp, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_uint,
                                                C.c_void_p, C.c_int]
    Z, Y, X = vol.shape
    W, H = 2048, 1024
    h = L.clvr_host_create()
    L.clvr_host_load(h, vol.ctypes.data, X, Y, Z, env.ctypes.data, env.shape[1], env.shape[0])
    L.clvr_host_flush(h, tf_source.encode())
    p = np.asarray(pos, np.float32).copy()
    look = np.array([0.9, 6.183], np.float32)  # app/ui.cpp:178
    changed = C.c_int(0)
    fp = lambda a: a.ctypes.data_as(fpp)
    out = {"frame": "%dx%d (the application's SCREEN_WIDTH x SCREEN_HEIGHT)" % (W, H), "passes_timed": 64}

    def frames(n, moving):
        for i in range(n):
            if moving:
                p[0] = np.float32(pos[0] + 0.25 * (i % 7))  # a slightly different pose every frame: k_primary runs every frame
            L.clvr_host_render_frame(h, fp(p), fp(look), W, H, 1, C.byref(changed))

    for rep in range(2):  # both variants twice, alternating; the faster run of each is reported (the first run of all also warms the copy path)
        for label, moving in (("render_frame_still_camera", False), ("render_frame_moving_camera", True)):
            L.clvr_host_flush(h, tf_source.encode())  # fresh cache (no voxel near the token cap), SDF rebuilt as the app does
            frames(4, moving)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frames(64, moving)
            t = time.perf_counter() - t0  # render_frame ends in a blocking pull: nothing is left in flight
            if label not in out or t * 1e3 / 64 < out[label]["ms_per_frame"]:
                out[label] = {"ms_per_frame": round(t * 1e3 / 64, 4), "msamples_per_sec": round(W * H * 64 / t / 1e6, 1),
                              "includes": "launch + k_bounce (1 pass) + resolve + blocking pull of the 8 MiB frame"}

    display = torch.cuda.Stream()
    shown = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for label, per_call in (("render_frame_device_1_pass_per_call", 1), ("render_frame_device_8_passes_per_call", 8)):
        L.clvr_host_flush(h, tf_source.encode())
        p[0] = np.float32(pos[0])
        calls = 64 // per_call
        for _ in range(2):
            L.clvr_host_render_frame_device(h, fp(p), fp(look), W, H, 1, C.c_void_p(shown.data_ptr()), W, H,
                                            C.c_void_p(display.cuda_stream), per_call)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            L.clvr_host_render_frame_device(h, fp(p), fp(look), W, H, 1, C.c_void_p(shown.data_ptr()), W, H,
                                            C.c_void_p(display.cuda_stream), per_call)
        display.synchronize()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        out[label] = {"ms_per_pass": round(t * 1e3 / 64, 4), "msamples_per_sec": round(W * H * 64 / t / 1e6, 1),
                      "includes": "launch + k_bounce + resolve into the display's device buffer, event hand-off; no readback"}
    L.clvr_host_destroy(h)
    return out


def run_voxel_single(ctx, kernel, d_vol, d_sdf, d_env, d_frame, pos, cdir, N, W, H, seeds, torch, ffi):
    """the reference-exact world-space voxel cache (token cap on) through the C ABI, one pass per launch, no readback"""
    n_cache = ffi.cache_len(N, N, N)
    cache = torch.zeros(n_cache // 2, dtype=torch.int32, device="cuda")
    m_cache = ctx.wrap(cache.data_ptr(), n_cache * 2)

    def voxel_pass(seed, wf):
        kernel.render(frame=d_frame, volume=d_vol, sdf=d_sdf, env=d_env, buffer_volume=m_cache, cam_pos=pos,
                      cam_dir=cdir, seed=seed, width=W, height=H, mode=ffi.ACCUM_VOXEL_CACHE, write_frame=wf)

    for s in seeds[:4]:
        voxel_pass(s, False)
    cache.zero_()
    ctx.invalidate_derived(scene=False, camera=True)
    ctx.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, s in enumerate(seeds):
        voxel_pass(s, i == len(seeds) - 1)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    timers = ctx.timing_read_all()
    ctx.set_timing(False)
    counts = (cache.view(-1, 2)[:, 1] >> 16) & 0xFFFF
    out = {
        "what": "world-space voxel cache with the 256-token cap (utility.cl:20-54), fresh cache, %d passes, one pass per "
                "launch, no host round trip per pass, frame resolved after the last pass" % len(seeds),
        "value": round(W * H * len(seeds) / el / 1e6, 3),
        "unit": "Msamples/s",
        "ms_per_pass": round(el * 1e3 / len(seeds), 4),
        "k_bounce_ms_per_pass": round(timers["bounce"][0] / max(timers["bounce"][1], 1), 4),
        "voxels_touched": int((counts > 0).sum().item()),
        "voxels_at_cap": int((counts >= 256).sum().item()),
    }
    exact_frame = d_frame.pull().copy()
    # the same job with the passes FUSED into launches of 64 (clwh_render_desc.n_seeds; legal: every request is granted while the
    # voxel's count is below 256 in any order, so counts are exact and entries below the cap identical -- tested): frame jobs as
    # the headline times them (fresh cache, camera changed, resolve at the end)
    S = ffi.MAX_SEEDS

    def fused_job(wf=True):
        cache.zero_()
        ctx.invalidate_derived(scene=False, camera=True)
        for i in range(0, len(seeds), S):
            kernel.render(frame=d_frame, volume=d_vol, sdf=d_sdf, env=d_env, buffer_volume=m_cache, cam_pos=pos, cam_dir=cdir,
                          seed=0, seeds=seeds[i:i + S], width=W, height=H, mode=ffi.ACCUM_VOXEL_CACHE,
                          write_frame=wf and i + S >= len(seeds))

    fused_job()
    torch.cuda.synchronize()
    reps = 20
    ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fused_job()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    timers = ctx.timing_read_all()
    ctx.set_timing(False)
    fcounts = (cache.view(-1, 2)[:, 1] >> 16) & 0xFFFF
    out["fused"] = {
        "what": "the same %d passes in launches of %d (one frame job = cache reset + k_primary + fused k_bounce + resolve), "
                "%d jobs one after the other" % (len(seeds), S, reps),
        "value": round(W * H * len(seeds) * reps / el / 1e6, 3), "unit": "Msamples/s",
        "ms_per_job": round(el * 1e3 / reps, 4),
        "k_bounce_ms_per_launch": round(timers["bounce"][0] / max(timers["bounce"][1], 1), 4),
        "counts_equal_one_pass_per_launch": bool(torch.equal(fcounts, counts)),
    }
    m_cache.release()
    del cache
    return out, exact_frame


def run_voxel_multi(args, ctx, kernel, d_vol, d_sdf, d_env, d_frame, pos, cdir, N, W, H, SPP, K, WU, rank, world, dev,
                    backend, barrier, max_over_ranks, sdf_build_s, n_layers, torch, dist, ffi, scene, tiles):
    """--accumulation voxel: the reference's world-space cache across ranks with its 256-token rule applied to the GLOBAL
    count (tiles.VoxelExchange): every rank renders its image tiles one pass per launch, the ranks all-gather the pass's
    per-pixel contributions and every rank applies all of them to its replica of the cache in (rank, pixel) order.
    A step is one frame job of SPP passes ending in the resolve of the whole frame from the replica."""
    n_cache = ffi.cache_len(N, N, N)
    n_entries = n_cache // 4
    cache = torch.zeros(n_cache // 2, dtype=torch.int32, device=dev)
    m_cache = ctx.wrap(cache.data_ptr(), n_cache * 2)
    npx = W * H
    scratch = torch.zeros(ffi.accum_len(W, H, world) * 4, dtype=torch.float32, device=dev)
    m_scratch = ctx.wrap(scratch.data_ptr(), scratch.numel() * 4)
    contrib = torch.zeros((npx, 4), dtype=torch.int32, device=dev)
    m_contrib = ctx.wrap(contrib.data_ptr(), npx * 16)
    hit = torch.full((npx,), -1, dtype=torch.int64, device=dev)
    m_hit = ctx.wrap(hit.data_ptr(), npx * 8)
    seeds = scene.glibc_rand((WU + K + 1) * SPP)
    xch = tiles.VoxelExchange(cache, world, ctx=ctx)

    def frame_job(j):
        cache.zero_()
        sd = seeds[j * SPP:(j + 1) * SPP]
        idx = None
        for i, s in enumerate(sd):
            kernel.render(frame=None, volume=d_vol, sdf=d_sdf, env=d_env, accum=m_scratch, cam_pos=pos, cam_dir=cdir, seed=s,
                          width=W, height=H, mode=ffi.ACCUM_IMAGE_SPACE, tile_rank=rank, tile_world=world, write_frame=False,
                          hit_index=m_hit if i == 0 else None, contrib=m_contrib)
            if i == 0:  # once per camera: which pixels of this rank hit which voxel
                idx = torch.nonzero((hit >= 0) & (hit < n_entries)).flatten()
                xch.set_camera(hit[idx])
            xch.add_pass(contrib[idx, :3])
        # every rank holds the whole cache: resolve the whole frame locally (no frame gather)
        kernel.render(frame=d_frame, volume=d_vol, sdf=d_sdf, env=d_env, buffer_volume=m_cache, cam_pos=pos, cam_dir=cdir, seed=0,
                      width=W, height=H, mode=ffi.ACCUM_VOXEL_CACHE, resolve_only=True)

    def region(first, k):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(first, first + k):
            frame_job(j)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for j in range(WU):
        region(j, 1)
    elapsed = max_over_ranks([region(WU, K)])[0]
    counts = (cache.view(-1, 2)[:, 1] >> 16) & 0xFFFF
    if rank == 0:
        print(json.dumps({
            "metric": "msamples_per_sec", "value": round(W * H * SPP * K / elapsed / 1e6, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": K, "warmup": WU, "ms_per_step": round(elapsed * 1e3 / K, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[%d]: %d^3 phantom, %dx%d frame, %d spp per step, reference-exact voxel cache with "
                                   "the 256-token rule applied to the global count" % (args.config - 1, N, W, H, SPP),
                       "accumulation": "replicated world-space cache; per pass one all-gather of the hit pixels' contributions "
                                       "(12 B each) + deterministic capped scatter-add on every rank (tiles.VoxelExchange)",
                       "sdf_build_s": round(sdf_build_s, 5), "sdf_layers": n_layers,
                       "voxels_touched": int((counts > 0).sum().item()), "voxels_at_cap": int((counts >= 256).sum().item()),
                       "max_count": int(counts.max().item())},
        }), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frame jobs per region (a job = spp passes of the config)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frame jobs before the first region")
    ap.add_argument("--config", type=int, default=2, choices=sorted(PRESETS),
                    help="BASELINE.json config: 2 = headline (512^3, 1080p, 64 spp, default TF); 3 = 512^3, 1080p, 256 spp, "
                         "gradient-reading TF (the 7-texel step); 4 = 2048^3, 3840x2160, 256 spp; 5 = 3840x2160, 1024 spp "
                         "(meant for --gpus 8)")
    ap.add_argument("--volume", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="render passes per frame job")
    ap.add_argument("--env", type=int, nargs=2, default=[4096, 2048])
    ap.add_argument("--tf", choices=["default", "gradient"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--no-secondary", action="store_true", help="skip end_to_end / drop_in_path / reference_exact_mode")
    ap.add_argument("--single-region", action="store_true", help="time ONE K-step region (profiling runs)")
    ap.add_argument("--accumulation", choices=["image", "voxel"], default="image",
                    help="image: float4 per pixel + one all-gather (the headline); voxel: the reference's world-space "
                         "cache, one pass per launch, contributions exchanged pass by pass under the global 256-token rule")
    ap.add_argument("--seeds-per-launch", type=int, default=64, help="render passes fused into one launch (1..64)")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="default 3 on one GPU and from six ranks on, else 2: consecutive frame jobs alternate between this many HIP streams (each with its own accumulation and "
                         "frame buffers), so that the next frame's primary hits and first waves fill the GPU while the last "
                         "waves of the previous frame's persistent launch drain; 1 = strictly one frame after the other")
    args = ap.parse_args()
    preset = PRESETS[args.config]
    for key, val in preset.items():
        if getattr(args, key, None) is None:
            setattr(args, key, val)

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args.gpus)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = world_env
    if args.gpus != world:
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    from cl_volume_renderer_amd import ffi, scene, tiles

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    rehearsal = os.environ.get("BENCH_ALL_ON_DEVICE") is not None
    if rehearsal:
        local_rank = int(os.environ["BENCH_ALL_ON_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        log("[bench] rank %d: backend %s, world size seen %d, device cuda:%d" % (rank, dist.get_backend(), dist.get_world_size(), local_rank))

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    def max_over_ranks(values):
        t = torch.tensor(values, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    N, W, H, SPP = args.volume, args.width, args.height, args.spp
    K, WU = args.steps, args.warmup
    S = max(1, min(ffi.MAX_SEEDS, args.seeds_per_launch))
    threads = host_threads()
    t_setup = time.time()
    vol = scene.phantom(N) if N < 1024 else scene.phantom_mt(N, threads=threads)
    env = scene.env_map(args.env[0], args.env[1])
    tf_source = scene.tf_default_source() if args.tf == "default" else scene.tf_gradient_source()
    pos, cdir = scene.default_camera(N)
    if rank == 0:
        log("[bench] scene: phantom(%d) %.1fs" % (N, time.time() - t_setup))

    # Frame jobs in flight share the GPU.  A rank's share of a multi-GPU job is small (a few million items per launch); each of its
    # launches then does best on a SHARE of the persistent grid, so that they co-reside instead of queueing behind each other:
    # 2.12 -> 2.06 / 1.20 -> 1.11 / 0.76 -> 0.65 ms per job for 2 / 4 / 8 ranks with two jobs in flight on 1024 blocks each, and 0.60 ms
    # for 8 ranks with three jobs on 768 blocks each (tools/emulate_rank.py, profiles/r02_emulate_rank_grid_sweep.txt); the single-GPU
    # job is indifferent and keeps the library default.  Placement knobs, read when a context is created; results do not depend on them.
    calibrate_lanes = args.frames_in_flight is None and world == 1
    if args.frames_in_flight is None:
        # one GPU: 1 / 2 / 3 jobs in flight = 31.5 / 32.5 / 33.0 Gsamples/s on the headline job early in round 3 (profiles/r03_frame_lanes.txt)
        # and 33.1 / 35.6 / 34.4 with the final k_bounce -- which of two and three is ahead differs by build and by box (the derived scene data
        # is shared, a lane costs its own accumulation / frame / hit buffers only), so a single GPU creates three lanes and the untimed
        # warm-up picks two or three of them (below); ranks of a multi-GPU job: see below
        args.frames_in_flight = 3 if (world >= 6 or world == 1) else 2
    if world > 1 and args.frames_in_flight >= 2:
        os.environ.setdefault("CLWH_TUNE_BLOCKS", "1024" if args.frames_in_flight == 2 else "768")
    # everything below runs on torch's current stream so torch.cuda.synchronize() covers it
    ctx = ffi.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    d_vol = ctx.image_from(vol)
    d_env = ctx.image_from(env, channels=4)
    d_sdf = ctx.image([N, N, N], 1, np.int8, (N, N, N))
    ctx.sdf_build(d_vol, tf_source, d_sdf)  # warm (allocations)
    torch.cuda.synchronize()
    sdf_times = []
    for _ in range(3 if N <= 512 else 1):
        t0 = time.perf_counter()
        n_layers = ctx.sdf_build(d_vol, tf_source, d_sdf)
        torch.cuda.synchronize()
        sdf_times.append(time.perf_counter() - t0)
    sdf_build_s = min(sdf_times)
    d_frame = ctx.image([W, H], 4, np.uint8, (H, W, 4))
    kernel = ctx.kernel("ray_marching.cl", "render", tf_source)

    if args.accumulation == "voxel":
        run_voxel_multi(args, ctx, kernel, d_vol, d_sdf, d_env, d_frame, pos, cdir, N, W, H, SPP, K, WU, rank, world, dev,
                        backend, barrier, max_over_ranks, sdf_build_s, n_layers, torch, dist, ffi, scene, tiles)
        return

    # ---- frame lanes: lane 0 runs on torch's current stream with the objects above; every further lane is another HIP
    # stream with its own context (derived arrays, primary hits), accumulation and frame buffers over the SAME volume /
    # SDF / env-map device memory (adopted with clwh_image_wrap, not copied)
    n_acc = ffi.accum_len(W, H, world)
    n_lanes = max(1, args.frames_in_flight)
    # One GPU, no --frames-in-flight given: besides the three lanes on the library's default grid, two lanes whose launches take HALF the
    # persistent grid each (CLWH_TUNE_BLOCKS, read when a context is created; placement only): two such launches co-reside instead of
    # queueing behind each other.  The untimed warm-up runs a region with each set and the timed regions use the fastest (below).
    half_grid = [n_lanes, n_lanes + 1] if (calibrate_lanes and n_lanes == 3 and "CLWH_TUNE_BLOCKS" not in os.environ) else []
    lanes = []
    for li in range(n_lanes + len(half_grid)):
        if li == 0:
            stream, lctx, lvol, lsdf, lenv, lframe, lkernel = torch.cuda.current_stream(), ctx, d_vol, d_sdf, d_env, d_frame, kernel
        else:
            # (the first half-grid lane shares torch's current stream with lane 0 -- they never run in the same region -- so that the pair is
            # one job on the current stream and one on a side stream, like the first two default lanes)
            stream = torch.cuda.current_stream() if (half_grid and li == half_grid[0]) else torch.cuda.Stream()
            if li in half_grid:
                os.environ["CLWH_TUNE_BLOCKS"] = "1024"
            try:
                lctx = ffi.Context(local_rank, stream=stream.cuda_stream)
            finally:
                if li in half_grid:
                    del os.environ["CLWH_TUNE_BLOCKS"]
            lvol = lctx.image_wrap(d_vol.device_ptr, [N, N, N], 1, np.int16, (N, N, N))
            lsdf = lctx.image_wrap(d_sdf.device_ptr, [N, N, N], 1, np.int8, (N, N, N))
            lenv = lctx.image_wrap(d_env.device_ptr, [args.env[0], args.env[1]], 4, np.uint8, (args.env[1], args.env[0], 4))
            lframe = lctx.image([W, H], 4, np.uint8, (H, W, 4))
            lkernel = lctx.kernel("ray_marching.cl", "render", tf_source)
        with torch.cuda.stream(stream):
            acc = torch.zeros(n_acc * 4, dtype=torch.float32, device=dev)
            # N > 1: the rank's tiles resolved to RGBA8 (4 B per pixel) are what the ranks exchange
            tl = torch.zeros(n_acc, dtype=torch.int32, device=dev)
            tl_all = torch.zeros(n_acc * world, dtype=torch.int32, device=dev)
        lanes.append(dict(stream=stream, ctx=lctx, vol=lvol, sdf=lsdf, env=lenv, frame=lframe, kernel=lkernel, accum=acc,
                          m_accum=lctx.wrap(acc.data_ptr(), acc.numel() * 4), tiles=tl, tiles_all=tl_all,
                          m_tiles=lctx.wrap(tl.data_ptr(), tl.numel() * 4), m_tiles_all=lctx.wrap(tl_all.data_ptr(), tl_all.numel() * 4)))
    torch.cuda.synchronize()
    lane_sets = {"%d x the default grid" % n_lanes: list(range(n_lanes))}
    if half_grid:
        lane_sets["2 x the default grid"] = [0, 1]
        lane_sets["2 x half the grid"] = half_grid
    active = list(range(n_lanes))  # the lanes the frame jobs alternate over

    # seeds: the glibc rand() stream the never-seeded reference draws from, one per pass, continuing over the jobs
    n_jobs_max = WU + 1 + K * (MAX_REGIONS + 6) + 96
    seed_stream = scene.glibc_rand(min(n_jobs_max * SPP, 4_000_000))

    def job_seeds(j):
        a = (j * SPP) % max(1, len(seed_stream) - SPP)
        return seed_stream[a:a + SPP]

    view = {"pos": pos, "dir": cdir}  # the camera of the frame jobs (the secondary measurement below switches to the close pose)

    def frame_job(j, pull=False, serial=False):
        """one step: camera -> primary hits -> SPP passes -> (gather) -> RGBA8 frame in HBM"""
        pos, cdir = view["pos"], view["dir"]
        ln = lanes[0 if serial else active[j % len(active)]]
        with torch.cuda.stream(ln["stream"]):
            ln["ctx"].invalidate_derived(scene=False, camera=True)  # the first frame after a camera move: k_primary runs
            ln["accum"].zero_()
            sd = job_seeds(j)
            for i in range(0, SPP, S):
                ln["kernel"].render(frame=None, volume=ln["vol"], sdf=ln["sdf"], env=ln["env"], accum=ln["m_accum"], cam_pos=pos,
                                    cam_dir=cdir, seed=0, seeds=sd[i:i + S], width=W, height=H, mode=ffi.ACCUM_IMAGE_SPACE,
                                    tile_rank=rank, tile_world=world, write_frame=False)
            if world == 1:
                ln["ctx"].accum_resolve(ln["m_accum"], 1, W, H, ln["frame"], ln["env"], pos, cdir)
            else:
                # resolve this rank's tiles, then ONE RCCL all-gather of RGBA8 tiles over xGMI (4 B per pixel, not 16), then place them
                ln["ctx"].accum_resolve_tiles(ln["m_accum"], rank, world, W, H, ln["m_tiles"], ln["env"], pos, cdir)
                tiles.gather_accum(ln["tiles"], ln["tiles_all"], world)
                ln["ctx"].frame_from_tiles(ln["m_tiles_all"], world, W, H, ln["frame"])
            if pull:
                return ln["frame"].pull()
        return None

    own_job_s = []  # this rank's own time for the jobs of a region (enqueue -> its GPU idle), before the closing barrier

    def timed_region(first_job, k, pull=False, serial=False):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(first_job, first_job + k):
            frame_job(j, pull, serial)
        torch.cuda.synchronize()
        own_job_s.append((time.perf_counter() - t0) / k)
        barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    job = 0
    est = []
    for _ in range(max(WU + 1, n_lanes)):  # warm-up jobs (every lane at least once: its work buffers get their final size)
        est.append(timed_region(job, 1))
        job += 1
    lane_calibration = None
    if len(lane_sets) > 1 and not args.single_region and min(est) < 0.03:  # (jobs of 100 ms and more gain nothing either way)
        # still warm-up, untimed as far as the result goes: the same K-job region with each set of lanes, twice
        lane_calibration = {}
        for name in list(lane_sets) * 2:
            active[:] = lane_sets[name]
            if name not in lane_calibration:
                timed_region(job, len(active))  # (every lane of the set once first)
                job += len(active)
            t = timed_region(job, K) / K
            job += K
            lane_calibration[name] = min(t, lane_calibration.get(name, t))
        active[:] = lane_sets[min(lane_calibration, key=lane_calibration.get)]
    else:
        active[:] = list(range(n_lanes))
    untimed_jobs = job  # warm-up jobs + the lane calibration's
    est_job = max_over_ranks([min(est)])[0]
    regions = 1 if args.single_region else max(1, min(MAX_REGIONS, int(math.ceil(MIN_TIMED_SECONDS / max(est_job * K, 1e-6)))))
    region_s = []
    for _ in range(regions):
        region_s.append(timed_region(job, K))
        job += K
    # the estimate came from single jobs, which do not overlap: add regions until MIN_TIMED_SECONDS really have been timed
    # (the decision is taken on a value every rank agrees on)
    while not args.single_region and len(region_s) < MAX_REGIONS and max_over_ranks([sum(region_s)])[0] < MIN_TIMED_SECONDS:
        region_s.append(timed_region(job, K))
        job += K
    regions = len(region_s)
    own_ms_headline = statistics.median(own_job_s[-regions:]) * 1e3
    # kernel durations for the roofline: HIP events around every launch, in a separate region in which the frame jobs run
    # strictly one after the other on one stream -- with two frames in flight a launch's events would also span the time it
    # shares the GPU with the other frame's kernels
    k_serial = max(2, min(K, 10))
    ctx.set_timing(True)
    serial_s = timed_region(job, k_serial, serial=True)
    job += k_serial
    timers = ctx.timing_read_all()
    ctx.set_timing(False)
    region_s = max_over_ranks(region_s)  # every region: the slowest rank's time
    elapsed = statistics.median(region_s)

    samples_per_job = W * H * SPP
    value = samples_per_job * K / elapsed / 1e6
    ranks_seen = [{"rank": 0, "backend": "none", "world_size_seen": 1, "device": local_rank, "own_ms_per_job": round(own_ms_headline, 4)}]
    if world > 1:
        # what every rank saw, so that a scaling run explains itself: the collective backend and world size, its device, and ITS
        # OWN median time per job (enqueue to idle, without waiting for the other ranks at the closing barrier)
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                                      "device": local_rank, "own_ms_per_job": round(own_ms_headline, 4)})
        ranks_seen = seen
    result = {
        "metric": "msamples_per_sec",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": K,
        "warmup": WU,
        "ms_per_step": round(elapsed * 1e3 / K, 4),
        "timed_region_s": round(sum(region_s), 4),  # everything between the first and the last timing bracket of the headline value
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "configs[%d]: %d^3 int16 phantom + %dx%d RGBA8 env map, %dx%d frame, %d spp per step "
                        "(1 spp per render pass), SDF empty-space skip, %s TF" % (
                            args.config - 1, N, args.env[0], args.env[1], W, H, SPP, args.tf),
            "step": "one frame job: k_primary (camera changed) + %d passes in %d launch(es) of <= %d + %s resolve to RGBA8" % (
                SPP, (SPP + S - 1) // S, S, "RCCL all-gather of the resolved RGBA8 tiles + " if world > 1 else ""),
            "accumulation": "image-space float4 per pixel, 8x8 tiles interleaved over ranks",
            "passes_per_launch": S,
            "untimed_jobs_before_the_first_region": untimed_jobs,
            "frames_in_flight": len(active),
            "persistent_grid_per_launch": ("half of the library's default (CLWH_TUNE_BLOCKS=1024): two launches co-reside" if active and active[0] in half_grid
                                           else os.environ.get("CLWH_TUNE_BLOCKS", "library default (2048 blocks)")),
            "frames_in_flight_calibration": ({"what": "ms per job of an untimed warm-up region (best of two) with each set of frame lanes; the timed "
                                              "regions use the fastest; the roofline's kernel durations come from single launches on the default grid",
                                              **{k: round(v * 1e3, 4) for k, v in lane_calibration.items()}} if lane_calibration else None),
            "one_frame_at_a_time": {"steps": k_serial, "ms_per_step": round(max_over_ranks([serial_s])[0] * 1e3 / k_serial, 4),
                                    "note": "the same jobs strictly serial on one stream; the per-kernel durations of the roofline come from this region"},
            "timed": {"regions": regions, "steps_per_region": K, "seconds_median": round(elapsed, 6),
                      "seconds_min": round(min(region_s), 6), "seconds_max": round(max(region_s), 6),
                      "seconds_total": round(sum(region_s), 4)},
            "sdf_build_s": round(sdf_build_s, 5),
            "sdf_layers": n_layers,
            "hbm_in_use_gib": round((torch.cuda.mem_get_info(dev)[1] - torch.cuda.mem_get_info(dev)[0]) / 2 ** 30, 2),
            "derived_scene": dict(zip(("id", "bytes", "contexts_sharing_it"), ctx.scene_info())),
            "ranks": ranks_seen,
            "rehearsal_on_one_gpu": bool(rehearsal),
        },
    }

    # ---- roofline + CPU baseline (oracle = checker / reported baseline only; rank 0)
    if rank == 0:
        from oracle import orc_ffi

        sdf_host = d_sdf.pull()
        osc = orc_ffi.Scene(vol, sdf_host, env, orc_ffi.parse_tf(tf_source), (W, H), mode=orc_ffi.MODE_IMAGE_SPACE,
                            tile_rank=0, tile_world=world, threads=threads)
        first = job_seeds(max(WU + 1, n_lanes))
        t0 = time.perf_counter()
        osc.render(pos, cdir, first[0])
        t1 = time.perf_counter() - t0
        passes = 1
        if world == 1 and not args.no_cpu_baseline:
            more = int(max(0, min(15, SPP - 1, args.cpu_seconds / max(t1, 1e-3) - 1)))
            for s in first[1:1 + more]:
                osc.render(pos, cdir, s)
            passes += more
        cpu_s = time.perf_counter() - t0
        own_px = (W * H) // world  # interleaved tiles: equal shares
        c = osc.counter_dict()
        c = {k: v / float(passes) for k, v in c.items()}  # per pass
        c["samples"] = own_px
        b_ms, b_n = timers["bounce"]
        p_ms, p_n = timers["primary"]
        f_ms, f_n = timers["fixup"]
        r_ms, r_n = timers["resolve"]
        passes_per_launch = SPP / float((SPP + S - 1) // S)
        grad = args.tf == "gradient"
        bounce_bytes = bytes_bounce(c, grad) * passes_per_launch    # per launch
        bounce_avg_ms = b_ms / max(b_n, 1)
        bounce_gbs = bounce_bytes / (bounce_avg_ms * 1e-3) / 1e9
        primary_bytes = bytes_primary(c)
        primary_avg_ms = p_ms / max(p_n, 1)
        contract_bps = bytes_contract_image_space(c) / float(own_px)
        contract_gbs = contract_bps * own_px * SPP * K / elapsed / 1e9
        key = "%d^3 %dx%d %s tf, %d passes per launch, world %d" % (N, W, H, args.tf, int(passes_per_launch), world)
        traffic = measured_traffic(key) or {}
        tb = traffic.get("k_bounce")
        result["roofline"] = {
            "bound": "hbm",
            "kernel": "k_bounce",
            "achieved": round(bounce_gbs, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(bounce_gbs / HBM_PEAK_GBS, 5),
            "traffic": round(tb["gb_per_launch"], 3) if tb else None,
            "traffic_rate": ({"unit": "GB/s", "value": round(tb["gb_per_launch"] / (bounce_avg_ms * 1e-3), 1),
                              "frac_of_peak": round(tb["gb_per_launch"] / (bounce_avg_ms * 1e-3) / HBM_PEAK_GBS, 4),
                              "what": "bytes ACTUALLY moved at the L2's memory side per launch (profiled run) / this run's launch "
                                      "duration: every missing 1-byte gather moves a 128-byte line"} if tb else None),
            "traffic_note": (tb or {}).get("note", "GB per launch from rocprofv3 --pmc (profiles/r02_traffic.json); null: this launch shape was not profiled"),
            "limiter": "dependent 1-byte gathers, not algorithmic HBM bytes: every gather that misses L1 moves a 128-byte line over the CU's "
                       "fill path (3.3 ns per thousand, measured), every L2 miss a line from the fabric (50-54 G lines/s, a plain random-load "
                       "probe's rate); on the 512^3 default-TF job those two and VALU issue (85 % busy) are of comparable size, larger jobs run "
                       "at the line rate of their misses (DESIGN.md 4, profiles/r03_k_bounce_memory_pipe_sensitivity.txt)",
            "bytes_per_sample": round(bytes_bounce(c, grad) / float(own_px), 3),
            "survey_formula_with_per_step_gradient_taps": ({
                "what": "SURVEY 8d's formula as written credits a TF that reads `gradient` with 7 volume texels per classified step; "
                        "k_bounce does not perform those fetches (class byte baked by k_repack), so `achieved` / `frac` above leave them out",
                "bytes_per_sample": round(bytes_bounce(c) / float(own_px), 3),
                "frac": round(bytes_bounce(c) * passes_per_launch / (bounce_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)} if grad else None),
            "executed": {"what": "the same launch credited with what k_bounce fetches at texel granularity: 1 step byte per march step, "
                                 "8 B hit record per secondary Hit, 4 B per env texel, 8 B per add (for a TF that reads `gradient`, SURVEY's "
                                 "formula above credits seven volume texels per step that the kernel does not read)",
                         "bytes_per_sample": round(bytes_bounce_executed(c) / float(own_px), 3),
                         "achieved": round(bytes_bounce_executed(c) * passes_per_launch / (bounce_avg_ms * 1e-3) / 1e9, 2),
                         "frac": round(bytes_bounce_executed(c) * passes_per_launch / (bounce_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "samples_per_launch": int(own_px * passes_per_launch),
            "algorithmic_gb_per_launch": round(bounce_bytes / 1e9, 4),
            "avg_launch_ms": round(bounce_avg_ms, 4),
            "launches": b_n,
            "what_is_counted": "only what k_bounce executes per pass: N_sdf + 2 N_vol + 4 N_env of the bounce phase + 8 N_add "
                               "(oracle counters split by phase; N_vol = one texel per classified step + six per Hit normal, whatever "
                               "the TF reads); the primary march, its normal and the miss pixels' env texels belong to k_primary "
                               "(once per camera)",
            "per_kernel": {
                "k_primary": {"algorithmic_gb_per_launch": round(primary_bytes / 1e9, 4), "avg_launch_ms": round(primary_avg_ms, 4),
                              "launches": p_n, "achieved": round(primary_bytes / max(primary_avg_ms, 1e-9) / 1e6, 2),
                              "frac": round(primary_bytes / max(primary_avg_ms, 1e-9) / 1e6 / HBM_PEAK_GBS, 5),
                              "traffic": round(traffic["k_primary"]["gb_per_launch"], 4) if "k_primary" in traffic else None},
                "k_env_fixup+k_commit": {"avg_launch_ms": round(f_ms / max(f_n, 1), 4), "launches": f_n},
                "k_accum_resolve": {"avg_launch_ms": round(r_ms / max(r_n, 1), 4), "launches": r_n,
                                    "algorithmic_gb_per_launch": round(W * H * 20 / 1e9, 4)},
            },
            "contract": {
                "what": "SURVEY 8d formula over the WHOLE path (primary march attributed to every sample although k_primary "
                        "performs it once per camera) divided by the whole-job time",
                "bytes_per_sample": round(contract_bps, 3), "achieved": round(contract_gbs, 2),
                "frac": round(contract_gbs / HBM_PEAK_GBS, 5)},
            "per_sample": {k: round(v / float(own_px), 4) for k, v in c.items() if k != "samples"},
        }
        # what the headline hides: most pixels of the default view miss the volume and cost nothing per pass
        hits_per_pass = c["n_hit"]
        result["hit_samples_per_sec"] = {
            "what": "traced samples only: (hit pixels of this rank's tiles x spp) per second of the same timed jobs; a miss pixel is "
                    "computed once per camera (k_primary) and counted spp times in `value`",
            "value": round(hits_per_pass * world * SPP * K / elapsed / 1e6, 3) if world == 1 else None,
            "unit": "Msamples/s", "hit_pixel_share": round(hits_per_pass / float(own_px), 4)}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = {
                "value": round(W * H * passes / cpu_s / 1e6, 3),
                "unit": "Msamples/s",
                "cores": threads,
                "kind": "port",
                "sample": "%d of the %d passes of one frame job (same scene, camera, seeds), OpenMP over image rows; "
                          "CPU restatement (oracle/), not POCL" % (passes, SPP),
            }
        del osc

    # ---- the same job from a pose whose frame the volume fills (scene.close_camera): its own value and k_bounce roofline
    if world == 1 and not args.no_secondary and rank == 0:
        view["pos"], view["dir"] = scene.close_camera(N)
        timed_region(job, len(active))  # every lane once: work buffers grow to this view's hit count
        job += len(active)
        kf = max(2, min(K, 10))
        t_close = []
        while sum(t_close) < 0.3 and len(t_close) < 20:
            t_close.append(timed_region(job, kf))
            job += kf
        timed_region(job, 1, serial=True)  # (lane 0 may not be among the lanes in use: its buffers meet this view here)
        job += 1
        ctx.set_timing(True)
        timed_region(job, 2, serial=True)
        job += 2
        tm = ctx.timing_read_all()
        ctx.set_timing(False)
        from oracle import orc_ffi

        osc = orc_ffi.Scene(vol, d_sdf.pull(), env, orc_ffi.parse_tf(tf_source), (W, H), mode=orc_ffi.MODE_IMAGE_SPACE, threads=threads)
        osc.render(view["pos"], view["dir"], job_seeds(0)[0])
        cc = osc.counter_dict()
        cc["samples"] = W * H
        del osc
        el_close = statistics.median(t_close)
        b_ms = tm["bounce"][0] / max(tm["bounce"][1], 1)
        ppl = SPP / float((SPP + S - 1) // S)
        gbs = bytes_bounce(cc, args.tf == "gradient") * ppl / (b_ms * 1e-3) / 1e9
        result["frame_filling_view"] = {
            "what": "the same frame job from scene.close_camera (0.6 N in front of the volume's centre, default viewing direction)",
            "camera": {"pos": [round(float(x), 3) for x in view["pos"]], "dir": [round(float(x), 5) for x in view["dir"]]},
            "hit_pixel_share": round(cc["n_hit"] / float(W * H), 4),
            "value": round(samples_per_job * kf / el_close / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(el_close * 1e3 / kf, 4),
            "hit_samples_per_sec": round(cc["n_hit"] * SPP * kf / el_close / 1e6, 3),
            "roofline": {"kernel": "k_bounce", "bound": "hbm", "avg_launch_ms": round(b_ms, 4), "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "bytes_per_sample": round(bytes_bounce(cc, args.tf == "gradient") / float(W * H), 3),
                         "executed_bytes_per_sample": round(bytes_bounce_executed(cc) / float(W * H), 3),
                         "k_primary_ms": round(tm["primary"][0] / max(tm["primary"][1], 1), 4)},
        }
        view["pos"], view["dir"] = pos, cdir

    # ---- end to end incl. the final frame readback (SURVEY 8d metric ii); same jobs, frame pulled after each
    if not args.no_secondary:
        k2 = max(1, min(K, 10))
        # (a blocking pull after every job leaves one job on the GPU at a time: on one GPU they run on lane 0, whose launches take the
        # whole grid, whatever set of lanes the timed regions used)
        t = max_over_ranks([timed_region(job, k2, pull=True, serial=(world == 1))])[0]
        job += k2
        if rank == 0:
            result["end_to_end"] = {
                "what": "the same frame jobs, each followed by the blocking device-to-host copy of the %dx%d RGBA8 frame "
                        "(clw_image::pull, %.1f MiB)" % (W, H, W * H * 4 / 2 ** 20),
                "value": round(samples_per_job * k2 / t / 1e6, 3), "unit": "Msamples/s", "steps": k2,
                "ms_per_step": round(t * 1e3 / k2, 4)}

    if rank == 0 and world == 1 and not args.no_secondary and args.config in (2, 3):
        # cap ON: the reference's own accumulation (256 tokens per voxel), the job's SPP passes one per launch
        result["reference_exact_mode"], exact_frame = run_voxel_single(ctx, kernel, d_vol, d_sdf, d_env, d_frame, pos, cdir, N, W, H,
                                                                       job_seeds(0), torch, ffi)
        # ---- the stated tolerance (north_star): the headline's image-space frame against the reference-exact frame of the SAME
        # job (same seeds); different estimators of the same radiance -- per pixel 64 samples vs per voxel up to 256 shared samples
        ln = lanes[0]
        with torch.cuda.stream(ln["stream"]):
            ln["accum"].zero_()
            ln["kernel"].render(frame=None, volume=ln["vol"], sdf=ln["sdf"], env=ln["env"], accum=ln["m_accum"], cam_pos=pos, cam_dir=cdir,
                                seed=0, seeds=job_seeds(0)[:S], width=W, height=H, mode=ffi.ACCUM_IMAGE_SPACE, write_frame=False)
            for i in range(S, SPP, S):
                ln["kernel"].render(frame=None, volume=ln["vol"], sdf=ln["sdf"], env=ln["env"], accum=ln["m_accum"], cam_pos=pos,
                                    cam_dir=cdir, seed=0, seeds=job_seeds(0)[i:i + S], width=W, height=H, mode=ffi.ACCUM_IMAGE_SPACE,
                                    write_frame=False)
            ln["ctx"].accum_resolve(ln["m_accum"], 1, W, H, ln["frame"], ln["env"], pos, cdir)
            image_frame = ln["frame"].pull()
        both = (image_frame[..., 3] == 1) & (exact_frame[..., 3] == 1)
        diff = np.abs(image_frame[both][:, :3].astype(np.int16) - exact_frame[both][:, :3].astype(np.int16)).max(axis=1)
        result["tolerance_vs_reference_exact"] = {
            "what": "RGBA8 frame of this job, image-space accumulation (the headline) vs the reference's capped voxel cache (one pass "
                    "per launch), same seeds; max over r, g, b per hit pixel; miss pixels and the hit mask are identical",
            "spp": SPP, "hit_pixels": int(both.sum()), "hit_mask_identical": bool(np.array_equal(image_frame[..., 3], exact_frame[..., 3])),
            "max_abs_diff_lsb": int(diff.max()), "mean_abs_diff_lsb": round(float(diff.mean()), 3),
            "share_within_1_lsb": round(float((diff <= 1).mean()), 4), "share_within_4_lsb": round(float((diff <= 4).mean()), 4),
            "share_within_16_lsb": round(float((diff <= 16).mean()), 4)}
        if args.config == 2:
            result["drop_in_path"] = run_dropin(vol, env, tf_source, pos, torch)
    # every rank checks every lane for a fix-up-buffer overflow (its samples would be missing from the gathered frame): no rank
    # prints a number when any of them overflowed
    failed = 0.0
    for ln in lanes:
        try:
            ln["ctx"].finish()
        except ffi.ClwhError as e:
            log("[bench] rank %d: %s" % (rank, e))
            failed = 1.0
    if max_over_ranks([failed])[0] != 0.0:
        raise SystemExit("bench.py: a render overflowed its fix-up buffer on some rank: results are incomplete, no line printed")
    if rank == 0:
        print(json.dumps(result), flush=True)

    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
