"""Times the pre-processing kernels next to the hot path and the volume upload transforms on the GPU, through the
generic launch exactly as the mirrored host classes call them:  python tools/time_volume_kernels.py [N]
Each line: time of the 3rd repetition (host clock around launch + sync), voxels/s, and the rate on the bytes the kernel must move
(unique voxels read + results written; the stencils' re-reads are cache hits by design)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from cl_volume_renderer_amd import ffi, scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.cuda.init()  # torch's HIP runtime first (see tests/conftest.py)
ctx = ffi.Context(0)
vol = scene.phantom(n)
v = ctx.image_from(vol)
ev = lambda g, l=8: (g + l - 1) // l * l  # noqa: E731
G = [ev(n)] * 3


def timed(name, fn, bytes_moved, reps=4):
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ctx.finish()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%-34s %d^3: %8.3f ms  %6.1f Gvoxel/s  %7.1f GB/s on %.2f GB (%.3f of 8 TB/s)" % (
        name, n, dt * 1e3, n**3 / dt / 1e9, bytes_moved / dt / 1e9, bytes_moved / 1e9, bytes_moved / dt / 8e12))


INIT = np.array([2**31 - 1, -2**31, 2**31 - 1, -2**31, -2**31], dtype=np.int32)
stats = ctx.buffer_from(INIT)
k_stats = ctx.kernel("reference_volume_figures.cl", "fetch_stats")
timed("fetch_stats (min/max value, gradient)", lambda: k_stats.launch(G, [4, 4, 4], v, stats), 2 * n**3)
st = stats.pull()

W = H = 500
bins = ctx.buffer_from(np.zeros(W * H, np.uint32))
k_hist = ctx.kernel("histogram.cl", "tf_sort_values")
timed("tf_sort_values (500x500 histogram)", lambda: k_hist.launch(G, [4, 4, 4], v, bins, np.uint32(W), np.uint32(H), float(st[0]), float(st[1]),
                                                                  float(st[2]), float(st[3])), 2 * n**3)

dst = ctx.image([n, n, n], 1, np.int16, (n, n, n))
k_bil = ctx.kernel("volume_filter.cl", "bilateral_filter")
timed("bilateral_filter (5x5x5 taps)", lambda: k_bil.launch(G, [4, 4, 4], v, dst), 4 * n**3)

c = n * 3 // 4
clip = ctx.image([c, c, c], 1, np.int16, (c, c, c))
b_start = ctx.buffer_from(np.array([n // 8] * 3, np.uint32))
b_len = ctx.buffer_from(np.array([c, c, c, 4], np.uint32))
k_clip = ctx.kernel("reference_volume_clip.cl", "apply_clip")
gc = [ev(c, 4)] * 3
t_bytes = 4 * c**3
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k_clip.launch(gc, [4, 4, 4], v, clip, b_start, b_len)
    ctx.finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("%-34s %d^3 -> %d^3: %8.3f ms  %7.1f GB/s on %.2f GB (%.3f of 8 TB/s)" % ("apply_clip", n, c, dt * 1e3, t_bytes / dt / 1e9, t_bytes / 1e9, t_bytes / dt / 8e12))
