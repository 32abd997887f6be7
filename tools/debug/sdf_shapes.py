"""debug: GPU SDF build against the oracle on awkward shapes, mismatch statistics"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
torch.cuda.init()
from cl_volume_renderer_amd import ffi, scene
from oracle import orc_ffi as orc
from tests.test_gpu_sdf import _blobs
orc.build(); orc.lib()
ctx = ffi.Context(0)
tf = scene.tf_default_source()
for dims in [(2, 40, 40), (130, 3, 7), (31, 97, 129), (3, 64, 64)]:
    vol = _blobs(dims, seed=sum(dims))
    want, n_want, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    Z, Y, X = vol.shape
    v = ctx.image_from(vol)
    s = ctx.image([X, Y, Z], 1, np.int8, (Z, Y, X))
    n = ctx.sdf_build(v, tf, s)
    got = s.pull()
    bad = np.argwhere(got != want)
    print(dims, "launches", n, n_want, "mismatches", len(bad), "of", got.size)
    for b in bad[:12]:
        print("   z,y,x", tuple(b), "got", got[tuple(b)], "want", want[tuple(b)])
