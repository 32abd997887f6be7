#!/usr/bin/env python3
"""bench.py -- headline benchmark of the volumetric path-tracing hot path on MI355X.

Metric (BASELINE.json): Msamples/s = width x height x spp / seconds, on configs[1]:
512^3 volume + 4096x2048 env map, 1920x1080, 64 spp (one `render` pass = one sample per pixel,
reference app/renderer.cpp:131-158), synthetic seeded inputs.

A "step" is one render pass over the whole frame.  `--gpus N` (launched by torch.distributed.run,
one rank per GPU) shards the SAME frame as interleaved 8x8 image tiles over the ranks (strong
scaling), each rank accumulating float4 per pixel; the timed region ends with one RCCL all-gather
of the accumulation tiles and the resolve of the RGBA8 frame.  Inputs are resident in HBM before
the timed region starts.

The JSON line also carries
  roofline     -- algorithmic bytes per launch (oracle texel counters) / the render
                  kernel's mean duration measured with HIP events on its stream, vs 8 TB/s HBM;
  cpu_baseline -- the CPU oracle (a port, not POCL) timed on this host on a bounded sample of the
                  same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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


def image_space_bytes(c, samples):
    """Algorithmic bytes of the image-space pass (DESIGN.md 'Measurement'): SURVEY 8d's formula with
    the accumulation done per pixel: no token, float4 read-modify-write (32 B) per granted sample,
    8 B of per-pixel hit scratch instead of the 4 B frame write."""
    return (c["n_sdf"] + 2 * c["n_vol"] + 4 * c["n_env"] + 32 * c["n_add"] + 8 * samples) / float(samples)


def measured_traffic(world, N, W, H, launches, args):
    """HBM-side bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under profiles/
    (bench.py cannot run the profiler on itself); only reported for the launch shape that was profiled."""
    path = os.path.join(ROOT, "profiles", "r01_k_bounce_traffic.json")
    if world != 1 or (N, W, H, args.steps, launches, args.tf) != (512, 1920, 1080, 64, 1, "default") or not os.path.exists(path):
        return None
    t = json.load(open(path))
    # MI355X_MICROARCH.md, HBM / rocprofv3: on gfx950 FETCH_SIZE tallies the L2's 128-byte fabric read requests at
    # 64 bytes -- double it; WRITE_SIZE reads exactly.  (The counters are KiB; the json holds bytes.)
    return round((2.0 * t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]) / 1e9, 3)


def voxel_cache_bytes(c, samples):
    """B = N_sdf + 2 N_vol + 4 N_env + 4 N_tok + 8 N_add + 8 N_read + 4."""
    return (c["n_sdf"] + 2 * c["n_vol"] + 4 * c["n_env"] + 4 * c["n_tok"] + 8 * c["n_add"]
            + 8 * c["n_read"] + 4 * samples) / float(samples)


def run_voxel_mode(args, ctx, kernel, d_vol, d_sdf, d_env, d_frame, pos, cdir, seeds, N, W, H, rank, world, dev, barrier,
                   sdf_build_s, n_layers):
    """Reference-exact accumulation: every rank renders its image tiles into a private world-space cache
    (token cap on, one pass per launch as renderer::render_frame does), then ONE all-reduce(SUM) of the
    caches (tiles.reduce_voxel_caches) and a resolve of the rank's tiles."""
    import torch
    import torch.distributed as dist

    from cl_volume_renderer_amd import ffi, tiles

    n_cache = ffi.cache_len(N, N, N)
    cache = torch.zeros(n_cache // 2, dtype=torch.int32, device=dev)
    m_cache = ctx.wrap(cache.data_ptr(), n_cache * 2)

    def one_pass(seed, write_frame):
        kernel.render(frame=d_frame, volume=d_vol, sdf=d_sdf, env=d_env, buffer_volume=m_cache, cam_pos=pos, cam_dir=cdir,
                      seed=seed, width=W, height=H, mode=ffi.ACCUM_VOXEL_CACHE, tile_rank=rank, tile_world=world,
                      write_frame=write_frame)

    for s in seeds[: args.warmup]:
        one_pass(s, False)
    tiles.reduce_voxel_caches(cache, world)  # untimed: channel set-up
    cache.zero_()
    ctx.invalidate_derived(scene=False, camera=True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in seeds[args.warmup:]:
        one_pass(s, False)
    tiles.reduce_voxel_caches(cache, world)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    counts = (cache.view(-1, 2)[:, 1] >> 16) & 0xFFFF
    if rank == 0:
        print(json.dumps({
            "metric": "msamples_per_sec", "value": round(W * H * args.steps / elapsed / 1e6, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[%d]: %d^3 phantom, %dx%d frame, %d spp, reference-exact voxel cache" % (
                           args.config - 1, N, W, H, args.steps),
                       "accumulation": "world-space voxel cache per rank (token cap on), one pass per launch, "
                                       "one all-reduce(SUM) of the caches at the end",
                       "sdf_build_s": round(sdf_build_s, 4), "sdf_layers": n_layers,
                       "voxels_touched": int((counts > 0).sum().item()), "max_count": int(counts.max().item())},
        }), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed render passes (spp); default from --config")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--env", type=int, nargs=2, default=[4096, 2048])
    ap.add_argument("--tf", choices=["default", "gradient"], default=None)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 5],
                    help="BASELINE.json config: 2 = headline (512^3, 1080p, 64 spp, default TF); 3 = 512^3, 1080p, "
                         "256 spp, gradient-reading TF (the 7-texel step); 5 = 3840x2160, 1024 spp (meant for --gpus 8). "
                         "Explicit --steps/--width/--height/--tf override the preset")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--no-secondary", action="store_true", help="skip the voxel-cache-mode measurement")
    ap.add_argument("--accumulation", choices=["image", "voxel"], default="image",
                    help="image: float4 per pixel + one all-gather (the headline); voxel: the reference's world-space "
                         "cache per rank, one pass per launch, + one all-reduce of the caches (exact below the token cap)")
    ap.add_argument("--seeds-per-launch", type=int, default=64,
                    help="render passes fused into one launch of the persistent bounce kernel (1..64)")
    args = ap.parse_args()
    preset = {2: dict(steps=64, width=1920, height=1080, tf="default"),
              3: dict(steps=256, width=1920, height=1080, tf="gradient"),
              5: dict(steps=1024, width=3840, height=2160, tf="default")}[args.config]
    for key, val in preset.items():
        if getattr(args, key) is None:
            setattr(args, key, val)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one rank per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    from cl_volume_renderer_amd import ffi, scene, tiles

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # rehearsal switches (not used by the driver): BENCH_DIST_BACKEND=gloo and BENCH_ALL_ON_DEVICE=0 run the
    # whole N>1 code path with every rank on one GPU and the gather staged through the host
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_ALL_ON_DEVICE") is not None:
        local_rank = int(os.environ["BENCH_ALL_ON_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    N, W, H = args.volume, args.width, args.height
    t_setup = time.time()
    vol = scene.phantom(N)
    env = scene.env_map(args.env[0], args.env[1])
    tf_source = scene.tf_default_source() if args.tf == "default" else scene.tf_gradient_source()
    pos, cdir = scene.default_camera(N)
    seeds = scene.glibc_rand(args.warmup + args.steps)
    if rank == 0:
        log("[bench] scene: phantom(%d) %.1fs" % (N, time.time() - t_setup))

    # everything below runs on torch's current stream so torch.cuda.synchronize() covers it
    ctx = ffi.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    d_vol = ctx.image_from(vol)
    d_env = ctx.image_from(env, channels=4)
    d_sdf = ctx.image([N, N, N], 1, np.int8, (N, N, N))
    torch.cuda.synchronize()
    t0 = time.time()
    n_layers = ctx.sdf_build(d_vol, tf_source, d_sdf)
    torch.cuda.synchronize()
    sdf_build_s = time.time() - t0
    d_frame = ctx.image([W, H], 4, np.uint8, (H, W, 4))
    kernel = ctx.kernel("ray_marching.cl", "render", tf_source)

    n_acc = ffi.accum_len(W, H, world)
    accum = torch.zeros(n_acc * 4, dtype=torch.float32, device=dev)
    accum_all = torch.zeros(n_acc * 4 * world, dtype=torch.float32, device=dev)
    m_accum = ctx.wrap(accum.data_ptr(), accum.numel() * 4)
    m_accum_all = ctx.wrap(accum_all.data_ptr(), accum_all.numel() * 4)

    S = max(1, min(ffi.MAX_SEEDS, args.seeds_per_launch))
    if args.accumulation == "voxel":
        run_voxel_mode(args, ctx, kernel, d_vol, d_sdf, d_env, d_frame, pos, cdir, seeds, N, W, H, rank, world, dev,
                       barrier, sdf_build_s, n_layers)
        return

    def render_passes(batch):
        """len(batch) render passes (steps) in one launch of the persistent bounce kernel"""
        kernel.render(frame=None, volume=d_vol, sdf=d_sdf, env=d_env, accum=m_accum, cam_pos=pos, cam_dir=cdir,
                      seed=0, seeds=batch, width=W, height=H, mode=ffi.ACCUM_IMAGE_SPACE, tile_rank=rank,
                      tile_world=world, write_frame=False)

    def batches(seq):
        return [seq[i:i + S] for i in range(0, len(seq), S)]

    for b in batches(seeds[: args.warmup]):
        render_passes(b)
    # one more untimed launch of the timed launches' shape, so that the context's work buffers (fix-up
    # records, hit list) have their final size before the clock starts
    first = seeds[args.warmup: args.warmup + S]
    if len(first) > args.warmup:
        render_passes(first)
    tiles.gather_accum(accum, accum_all, world)  # untimed: RCCL sets its channels up on the first collective of a kind
    ctx.accum_resolve(m_accum_all, world, W, H, d_frame, d_env, pos, cdir)
    accum.zero_()
    # the timed region starts like the first frame after a camera move: the per-camera primary hits are
    # rebuilt inside it (once); the packed records are flush-time data like the SDF and stay resident
    ctx.invalidate_derived(scene=False, camera=True)
    ctx.set_timing(True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches(seeds[args.warmup:]):
        render_passes(b)
    tiles.gather_accum(accum, accum_all, world)  # N>1: ONE RCCL all-gather of the float4 tiles over xGMI
    ctx.accum_resolve(m_accum_all, world, W, H, d_frame, d_env, pos, cdir)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = ctx.timing_read()
    ctx.set_timing(False)

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    samples = W * H * args.steps
    value = samples / elapsed / 1e6
    result = {
        "metric": "msamples_per_sec",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "configs[%d]: %d^3 int16 phantom + %dx%d RGBA8 env map, %dx%d frame, %d spp "
                        "(1 spp per render pass), SDF empty-space skip, %s TF" % (
                            args.config - 1, N, args.env[0], args.env[1], W, H, args.steps, args.tf),
            "accumulation": "image-space float4 per pixel, 8x8 tiles interleaved over ranks, "
                            "one RCCL all-gather + resolve at the end of the timed region",
            "passes_per_launch": S,
            "sdf_build_s": round(sdf_build_s, 4),
            "sdf_layers": n_layers,
        },
    }

    # ---- roofline + CPU baseline (oracle = checker / reported baseline only; rank 0)
    if rank == 0:
        from oracle import orc_ffi

        threads = host_threads()
        sdf_host = d_sdf.pull()
        osc = orc_ffi.Scene(vol, sdf_host, env, orc_ffi.parse_tf(tf_source), (W, H), mode=orc_ffi.MODE_IMAGE_SPACE,
                            tile_rank=0, tile_world=world, threads=threads)
        t0 = time.perf_counter()
        osc.render(pos, cdir, seeds[args.warmup])
        t1 = time.perf_counter() - t0
        passes = 1
        if world == 1 and not args.no_cpu_baseline:
            more = int(max(0, min(15, args.cpu_seconds / max(t1, 1e-3) - 1)))
            for s in seeds[args.warmup + 1: args.warmup + 1 + more]:
                osc.render(pos, cdir, s)
            passes += more
        cpu_s = time.perf_counter() - t0
        own_px = (W * H) // world  # interleaved tiles: equal shares (1920x1080 tiles divide evenly)
        counters = osc.counter_dict()
        bps = image_space_bytes(counters, own_px * passes)
        spl = args.steps / float(max(kern_n, 1))  # render passes (seeds) per launch
        per_launch = bps * own_px * spl
        avg_ms = kern_ms / max(kern_n, 1)
        achieved = per_launch / (avg_ms * 1e-3) / 1e9
        result["roofline"] = {
            "bound": "hbm",
            "kernel": "k_bounce",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "traffic_unit": "GB per launch: 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, rocprofv3 --pmc, separate passes",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": measured_traffic(world, N, W, H, kern_n, args),
            "bytes_per_sample": round(bps, 3),
            "samples_per_launch": int(own_px * spl),
            "avg_launch_ms": round(avg_ms, 4),
            "launches": kern_n,
            "per_sample": {k: round(v / float(own_px * passes), 4) for k, v in counters.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = {
                "value": round(W * H * passes / cpu_s / 1e6, 3),
                "unit": "Msamples/s",
                "cores": threads,
                "kind": "port",
                "sample": "%d of the %d timed passes (same scene, camera, seeds), OpenMP over image rows; "
                          "CPU restatement (oracle/), not POCL" % (passes, args.steps),
            }
        # parity spot check of the measured job itself: the GPU's accumulated float4 for the oracle's passes
        # is checked in tests; here only report that the frame was produced
        del osc

        # ---- secondary: the reference-exact voxel-cache mode (token cap on), N=1 only
        if world == 1 and not args.no_secondary:
            n_cache = ffi.cache_len(N, N, N)
            cache = torch.zeros(n_cache // 2, dtype=torch.int32, device=dev)
            m_cache = ctx.wrap(cache.data_ptr(), n_cache * 2)

            def voxel_pass(seed, wf):
                kernel.render(frame=d_frame, volume=d_vol, sdf=d_sdf, env=d_env, buffer_volume=m_cache, cam_pos=pos,
                              cam_dir=cdir, seed=seed, width=W, height=H, mode=ffi.ACCUM_VOXEL_CACHE, write_frame=wf)

            for s in seeds[: args.warmup]:
                voxel_pass(s, False)
            cache.zero_()
            ctx.invalidate_derived(scene=False, camera=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i, s in enumerate(seeds[args.warmup:]):
                voxel_pass(s, i == args.steps - 1)
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t0
            counts = (cache.view(-1, 2)[:, 1] >> 16) & 0xFFFF
            result["reference_exact_mode"] = {
                "what": "world-space voxel cache with the 256-token cap (utility.cl:20-54), fresh cache, same seeds, "
                        "one pass per launch, frame resolved after the last pass",
                "value": round(samples / el2 / 1e6, 3),
                "unit": "Msamples/s",
                "ms_per_step": round(el2 * 1e3 / args.steps, 4),
                "voxels_touched": int((counts > 0).sum().item()),
                "voxels_at_cap": int((counts >= 256).sum().item()),
            }
            del cache
        print(json.dumps(result), flush=True)

    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
