"""Generates the env-map fixtures of tests/golden/hostio/: small .hdr / .png / .jpg files and the RGBA8 the REFERENCE's
own loader (oracle/_ref/libref_hostio.so = the reference's app/hdre_loader.cpp + stb_image.h, built in place by
oracle/ref/Makefile) decodes them to.  Run from the repo root in a container that holds /root/reference:
    python tests/golden/make_hostio_golden.py
The fixtures let tests/test_host_mirror.py::test_env_map_golden_fixtures check the loaders where neither the reference
tree nor oracle/_ref exists."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cl_volume_renderer_amd import scene  # noqa: E402
from tests.jpeg_writer import Encoder, quant_table  # noqa: E402
from tests.test_host_mirror import _png_bytes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "hostio")
os.makedirs(OUT, exist_ok=True)
L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_hostio.so"))
L.ref_env_load.restype = C.c_longlong
L.ref_env_load.argtypes = [C.c_char_p, C.POINTER(C.c_uint), C.c_void_p, C.c_longlong]

rng = np.random.default_rng(2024)
w, h = 24, 16
files = {}
rgb = (rng.random((h, w, 3), dtype=np.float32) ** 2 * 3.0).astype(np.float32)
rgb[:, :8] = rgb[:, :1]
p = os.path.join(OUT, "env_rle.hdr")
scene.write_hdr(p, scene.float_to_rgbe(rgb), rle=True)
files["env_rle.hdr"] = p
img = rng.integers(0, 256, (h, w, 4))
p = os.path.join(OUT, "env_rgba_adam7.png")
open(p, "wb").write(_png_bytes(img, 6, 8, True))
files["env_rgba_adam7.png"] = p
yy, xx = np.mgrid[0:h, 0:w]
planes = [np.clip(128 + 80 * np.sin(xx / (3.0 + c)) * np.cos(yy / 4.0) + rng.normal(0, 10, (h, w)), 0, 255).astype(np.uint8) for c in range(3)]
for name, prog in (("env_420_baseline.jpg", False), ("env_420_progressive.jpg", True)):
    enc = Encoder(planes, [(2, 2), (1, 1), (1, 1)], [quant_table(0.7), quant_table(1.2)], [0, 1, 1], progressive=prog, restart=2 if not prog else 0, style=1)
    p = os.path.join(OUT, name)
    open(p, "wb").write(enc.encode_progressive() if prog else enc.encode_baseline())
    files[name] = p
for name, p in files.items():
    out = np.zeros((h, w, 4), np.uint8)
    d = (C.c_uint * 2)()
    assert L.ref_env_load(p.encode(), d, out.ctypes.data, out.nbytes) == out.nbytes and (d[0], d[1]) == (w, h)
    np.save(os.path.join(OUT, name + ".rgba.npy"), out)
    print(name, os.path.getsize(p), "bytes ->", out.shape)
