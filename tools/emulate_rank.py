"""What one rank of an N-GPU strong-scaling run computes, timed alone on one GPU (no collective):
python tools/emulate_rank.py WORLD [RANK]   -- the headline scene, 64 seeds in one launch, image-space mode."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from cl_volume_renderer_amd import ffi, scene  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, W, H, S = 512, 1920, 1080, 64
vol = scene.phantom(N)
env = scene.env_map(4096, 2048)
tf = scene.tf_default_source()
pos, cdir = scene.default_camera(N)
seeds = scene.glibc_rand(S)
ctx = ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
d_vol = ctx.image_from(vol)
d_env = ctx.image_from(env, channels=4)
d_sdf = ctx.image([N, N, N], 1, np.int8, (N, N, N))
ctx.sdf_build(d_vol, tf, d_sdf)
kernel = ctx.kernel("ray_marching.cl", "render", tf)
accum = torch.zeros(ffi.accum_len(W, H, world) * 4, dtype=torch.float32, device="cuda")
m_accum = ctx.wrap(accum.data_ptr(), accum.numel() * 4)


def go():
    kernel.render(frame=None, volume=d_vol, sdf=d_sdf, env=d_env, accum=m_accum, cam_pos=pos, cam_dir=cdir, seed=0,
                  seeds=seeds, width=W, height=H, mode=ffi.ACCUM_IMAGE_SPACE, tile_rank=rank, tile_world=world,
                  write_frame=False)


go()
torch.cuda.synchronize()
for rep in range(3):
    ctx.invalidate_derived(scene=False, camera=True)
    ctx.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms, n = ctx.timing_read()
    print("world %d rank %d: %.3f ms wall for 64 passes (k_bounce %.3f ms) -> x%d ranks = %.0f Msamples/s if nothing else cost time"
          % (world, rank, dt * 1e3, kern_ms, world, W * H * S / dt / 1e6))
