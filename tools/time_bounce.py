"""k_bounce alone, for A/B comparisons on ONE box (box-to-box differences are larger than most kernel changes):
python tools/time_bounce.py [--tf default|gradient] [--close] [--mode image|voxel] [--seeds 64] [--jobs 20] [--volume 512]
Prints the average launch duration (HIP events around the kernel) and the time per frame job.  CLWH_LIBRARY picks the build."""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from cl_volume_renderer_amd import ffi, scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tf", default="default")
ap.add_argument("--close", action="store_true")
ap.add_argument("--mode", default="image")
ap.add_argument("--seeds", type=int, default=64)
ap.add_argument("--jobs", type=int, default=20)
ap.add_argument("--volume", type=int, default=512)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
args = ap.parse_args()
N, W, H = args.volume, args.width, args.height
vol = scene.phantom(N) if N < 1024 else scene.phantom_mt(N, threads=16)
env = scene.env_map(4096, 2048)
tf = scene.tf_default_source() if args.tf == "default" else scene.tf_gradient_source()
pos, cdir = scene.close_camera(N) if args.close else scene.default_camera(N)
seeds = scene.glibc_rand(args.seeds)
torch.cuda.init()
ctx = ffi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
d_vol = ctx.image_from(vol)
d_env = ctx.image_from(env, channels=4)
d_sdf = ctx.image([N, N, N], 1, np.int8, (N, N, N))
ctx.sdf_build(d_vol, tf, d_sdf)
k = ctx.kernel("ray_marching.cl", "render", tf)
acc = torch.zeros(ffi.accum_len(W, H, 1) * 4, dtype=torch.float32, device="cuda")
m_acc = ctx.wrap(acc.data_ptr(), acc.numel() * 4)
image = args.mode == "image"
if not image:
    n_cache = ffi.cache_len(N, N, N)
    cache = torch.zeros(n_cache // 2, dtype=torch.int32, device="cuda")
    m_cache = ctx.wrap(cache.data_ptr(), n_cache * 2)


def job():
    ctx.invalidate_derived(scene=False, camera=True)
    if image:
        acc.zero_()
        k.render(frame=None, volume=d_vol, sdf=d_sdf, env=d_env, accum=m_acc, cam_pos=pos, cam_dir=cdir, seed=0, seeds=seeds,
                 width=W, height=H, mode=ffi.ACCUM_IMAGE_SPACE, write_frame=False)
    else:
        cache.zero_()
        k.render(frame=None, volume=d_vol, sdf=d_sdf, env=d_env, buffer_volume=m_cache, cam_pos=pos, cam_dir=cdir, seed=0, seeds=seeds,
                 width=W, height=H, mode=ffi.ACCUM_VOXEL_CACHE, write_frame=False)


for _ in range(3):
    job()
torch.cuda.synchronize()
ctx.set_timing(True)
t0 = time.perf_counter()
for _ in range(args.jobs):
    job()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / args.jobs
t = ctx.timing_read_all()
print("%s: k_bounce %.4f ms per launch (%d launches), k_primary %.4f ms, %.4f ms per job  [%s TF, %s camera, %s, %d seeds, %d^3]" % (
    ffi.LIB_PATH.split("/")[-1], t["bounce"][0] / max(t["bounce"][1], 1), t["bounce"][1], t["primary"][0] / max(t["primary"][1], 1), el * 1e3,
    args.tf, "close" if args.close else "default", args.mode, args.seeds, N))
