"""Times k_repack (volume + SDF + TF -> bricked step bytes + hit records) on the headline phantom: python tools/time_repack.py [N]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from cl_volume_renderer_amd import ffi, scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.cuda.init()
ctx = ffi.Context(0)
vol = scene.phantom(n) if n < 1024 else scene.phantom_mt(n, threads=16)
env = scene.env_map(256, 128)
tf = scene.tf_default_source()
d_vol = ctx.image_from(vol)
d_env = ctx.image_from(env, channels=4)
d_sdf = ctx.image([n, n, n], 1, np.int8, (n, n, n))
ctx.sdf_build(d_vol, tf, d_sdf)
k = ctx.kernel("ray_marching.cl", "render", tf)
w = h = 64
acc = ctx.buffer(ffi.accum_len(w, h, 1) * 16, np.float32)
pos, d = scene.default_camera(n)
for rep in range(4):
    ctx.invalidate_derived(scene=True, camera=True)
    ctx.set_timing(True)
    k.render(frame=None, volume=d_vol, sdf=d_sdf, env=d_env, accum=acc, cam_pos=pos, cam_dir=d, seed=1, width=w, height=h,
             mode=ffi.ACCUM_IMAGE_SPACE, write_frame=False)
    ctx.finish()
    t = ctx.timing_read_all()
    ctx.set_timing(False)
print("k_repack %d^3: %.3f ms (%.2f GB/s of the 12 bytes per voxel it must read and write)" % (n, t["repack"][0], 12.0 * n ** 3 / t["repack"][0] / 1e6))
