"""Times the bilateral volume filter (volume_filter.cl drop-in) on the GPU: python tools/time_filter.py [N]"""
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
dst = ctx.image([n, n, n], 1, np.int16, (n, n, n))
k = ctx.kernel("volume_filter.cl", "bilateral_filter")
g = [(n + 7) // 8 * 8] * 3
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k.launch(g, [4, 4, 4], v, dst)
    ctx.finish() if hasattr(ctx, "finish") else None
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("bilateral_filter %d^3: %.3f ms  (%.1f Gvoxel/s, %.1f Gtap/s, %.0f GB/s of the 4 B/voxel stream)" % (
        n, dt * 1e3, n**3 / dt / 1e9, 125 * n**3 / dt / 1e9, 4 * n**3 / dt / 1e9))
