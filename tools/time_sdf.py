"""Times clwh_sdf_build on the headline phantom and on a nearly empty volume (fixed per-layer cost vs per-tile cost):
python tools/time_sdf.py [N]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from cl_volume_renderer_amd import ffi, scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.cuda.init()  # torch's HIP runtime first (see tests/conftest.py)
ctx = ffi.Context(0)
tf = scene.tf_default_source()
small = np.zeros((n, n, n), np.int16)
small[n // 2 - 8: n // 2 + 8, n // 2 - 8: n // 2 + 8, n // 2 - 8: n // 2 + 8] = 800
for name, vol in (("phantom", scene.phantom(n)), ("16^3 cube in an empty volume", small)):
    d_vol = ctx.image_from(vol)
    d_sdf = ctx.image([n, n, n], 1, np.int8, (n, n, n))
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        layers = ctx.sdf_build(d_vol, tf, d_sdf)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%s %d^3: %.2f ms, %d layers (%.1f us per layer)" % (name, n, dt * 1e3, layers, dt * 1e6 / max(layers, 1)))
    d_vol.release()
    d_sdf.release()
