"""What one rank of an N-GPU strong-scaling run computes, timed alone on one GPU (no collective):
python tools/emulate_rank.py [--width W --height H --spp S] WORLD [WORLD ...]
One frame job = k_primary + S passes in launches of 64, image-space mode, the rank's interleaved 8x8 tiles only."""
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
args = ap.parse_args()
N, W, H, SPP = args.volume, args.width, args.height, args.spp
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
kernel = ctx.kernel("ray_marching.cl", "render", tf)
print("scene: phantom(%d), %dx%d, %d spp per job, default TF; one MI355X computing ONE rank's share" % (N, W, H, SPP))
base = None
for world in args.worlds:
    accum = torch.zeros(ffi.accum_len(W, H, world) * 4, dtype=torch.float32, device="cuda")
    m_accum = ctx.wrap(accum.data_ptr(), accum.numel() * 4)

    def go(rank=0):
        ctx.invalidate_derived(scene=False, camera=True)
        for i in range(0, SPP, 64):
            kernel.render(frame=None, volume=d_vol, sdf=d_sdf, env=d_env, accum=m_accum, cam_pos=pos, cam_dir=cdir, seed=0,
                          seeds=seeds[i:i + 64], width=W, height=H, mode=ffi.ACCUM_IMAGE_SPACE, tile_rank=rank, tile_world=world,
                          write_frame=False)

    go()
    torch.cuda.synchronize()
    best = None
    for rep in range(5):
        ctx.set_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = ctx.timing_read_all()
        ctx.set_timing(False)
        if best is None or dt < best[0]:
            best = (dt, t["bounce"][0], t["primary"][0])
    if base is None:
        base = best[0] * world
    print("world %d: %.3f ms per job (k_bounce %.3f ms, k_primary %.3f ms) -> %.1f Gsamples/s if the %d ranks ran like this one; "
          "%.0f %% of ideal strong scaling from world %d" % (world, best[0] * 1e3, best[1], best[2], W * H * SPP / best[0] / 1e9, world,
                                                             100.0 * base / (best[0] * world), args.worlds[0]))
    del accum
