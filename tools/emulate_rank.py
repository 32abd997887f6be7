"""What one rank of an N-GPU strong-scaling run computes, timed alone on one GPU (no collective):
python tools/emulate_rank.py [--width W --height H --spp S --frames-in-flight F] WORLD [WORLD ...]
One frame job = k_primary + S passes in launches of 64, image-space mode, the rank's interleaved 8x8 tiles only;
20 jobs back to back, alternating over F HIP streams exactly as bench.py does."""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from cl_volume_renderer_amd import ffi, scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("worlds", type=int, nargs="+")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--volume", type=int, default=512)
ap.add_argument("--frames-in-flight", type=int, default=2)
ap.add_argument("--jobs", type=int, default=20)
args = ap.parse_args()
N, W, H, SPP, F = args.volume, args.width, args.height, args.spp, max(1, args.frames_in_flight)
vol = scene.phantom(N)
env = scene.env_map(4096, 2048)
tf = scene.tf_default_source()
pos, cdir = scene.default_camera(N)
seeds = scene.glibc_rand(SPP)
torch.cuda.init()
ctx = ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
d_vol = ctx.image_from(vol)
d_env = ctx.image_from(env, channels=4)
d_sdf = ctx.image([N, N, N], 1, np.int8, (N, N, N))
ctx.sdf_build(d_vol, tf, d_sdf)
lanes = []
for li in range(F):
    if li == 0:
        stream, lctx, lvol, lsdf, lenv = torch.cuda.current_stream(), ctx, d_vol, d_sdf, d_env
    else:
        stream = torch.cuda.Stream()
        lctx = ffi.Context(0, stream=stream.cuda_stream)
        lvol = lctx.image_wrap(d_vol.device_ptr, [N, N, N], 1, np.int16, (N, N, N))
        lsdf = lctx.image_wrap(d_sdf.device_ptr, [N, N, N], 1, np.int8, (N, N, N))
        lenv = lctx.image_wrap(d_env.device_ptr, [4096, 2048], 4, np.uint8, (2048, 4096, 4))
    lanes.append(dict(stream=stream, ctx=lctx, vol=lvol, sdf=lsdf, env=lenv, kernel=lctx.kernel("ray_marching.cl", "render", tf)))
print("scene: phantom(%d), %dx%d, %d spp per job, default TF; one MI355X computing ONE rank's share, %d frame job(s) in flight" % (N, W, H, SPP, F))
base = None
for world in args.worlds:
    for ln in lanes:
        with torch.cuda.stream(ln["stream"]):
            ln["accum"] = torch.zeros(ffi.accum_len(W, H, world) * 4, dtype=torch.float32, device="cuda")
        ln["m_accum"] = ln["ctx"].wrap(ln["accum"].data_ptr(), ln["accum"].numel() * 4)

    def job(j, rank=0):
        ln = lanes[j % F]
        with torch.cuda.stream(ln["stream"]):
            ln["ctx"].invalidate_derived(scene=False, camera=True)
            ln["accum"].zero_()
            for i in range(0, SPP, 64):
                ln["kernel"].render(frame=None, volume=ln["vol"], sdf=ln["sdf"], env=ln["env"], accum=ln["m_accum"], cam_pos=pos,
                                    cam_dir=cdir, seed=0, seeds=seeds[i:i + 64], width=W, height=H, mode=ffi.ACCUM_IMAGE_SPACE,
                                    tile_rank=rank, tile_world=world, write_frame=False)

    for j in range(2 * F):
        job(j)
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(args.jobs):
            job(j)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.jobs
        best = dt if best is None else min(best, dt)
    if base is None:
        base = best * world
    print("world %d: %.3f ms per job -> %.1f Gsamples/s if the %d ranks ran like this one; %.0f %% of ideal strong scaling from world %d" % (
        world, best * 1e3, W * H * SPP / best / 1e9, world, 100.0 * base / (best * world), args.worlds[0]))
