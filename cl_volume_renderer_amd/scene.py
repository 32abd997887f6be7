"""Synthetic scenes for tests and bench: seeded CT-like phantom, procedural
environment map, the reference's transfer-function source strings, camera and seed stream.

No dataset ships with the reference except the 38x35x38 SDF test block, so every render-path
input is synthetic and reproducible from a seed.
"""
from __future__ import annotations

import gzip
import math
import zlib

import numpy as np

# ---------------------------------------------------------------------------------------------
# volume


def phantom(n: int, seed: int = 1234, dims=None) -> np.ndarray:
    """int16 [z][y][x] (x fastest) phantom: background -1000, ball r<0.42N = 40, shell
    0.30N<r<0.36N = 900 ("bone", inside the default TF [500,1200]), slab |x-c|<0.05N carved back
    to -1000 so secondary rays see interior, plus uniform integer noise in [-20, 20]."""
    if dims is None:
        dims = (n, n, n)
    X, Y, Z = dims
    rng = np.random.default_rng(seed)
    cx, cy, cz = (X - 1) / 2.0, (Y - 1) / 2.0, (Z - 1) / 2.0
    xs = (np.arange(X, dtype=np.float32) - cx)[None, :]
    ys = (np.arange(Y, dtype=np.float32) - cy)[:, None]
    r2_xy = xs * xs + ys * ys
    slab = np.abs(xs) < 0.05 * n
    out = np.empty((Z, Y, X), dtype=np.int16)
    for z in range(Z):
        r = np.sqrt(r2_xy + np.float32((z - cz) ** 2))
        v = np.full((Y, X), -1000, dtype=np.int16)
        v[r < 0.42 * n] = 40
        v[(r > 0.30 * n) & (r < 0.36 * n)] = 900
        v[np.broadcast_to(slab, v.shape)] = -1000
        v += rng.integers(-20, 21, size=(Y, X), dtype=np.int16)
        out[z] = v
    return out


def env_map(width: int = 4096, height: int = 2048, seed: int = 7) -> np.ndarray:
    """uint8 [h][w][4] procedural sky: vertical gradient, a sun disc and a few seeded soft blobs
    so that neighbouring texels differ (an env-texel flip is visible in parity tests)."""
    rng = np.random.default_rng(seed)
    v = (np.arange(height, dtype=np.float32) + 0.5)[:, None] / height
    u = (np.arange(width, dtype=np.float32) + 0.5)[None, :] / width
    r = 90 + 120 * (1 - v) + 0 * u
    g = 120 + 110 * (1 - v) + 10 * np.sin(2 * np.pi * u)
    b = 170 + 80 * (1 - v) + 0 * u
    su, sv = 0.31, 0.28
    d2 = ((u - su) * 2.0) ** 2 + (v - sv) ** 2
    sun = np.exp(-d2 / 0.0009)
    r = r + 255 * sun
    g = g + 240 * sun
    b = b + 200 * sun
    for _ in range(6):
        bu, bv, bs = rng.random(), rng.random(), 0.01 + 0.03 * rng.random()
        amp = rng.integers(-40, 41, size=3)
        blob = np.exp(-(((u - bu) * 2.0) ** 2 + (v - bv) ** 2) / bs)
        r = r + amp[0] * blob
        g = g + amp[1] * blob
        b = b + amp[2] * blob
    img = np.empty((height, width, 4), dtype=np.uint8)
    img[..., 0] = np.clip(r, 0, 255).astype(np.uint8)
    img[..., 1] = np.clip(g, 0, 255).astype(np.uint8)
    img[..., 2] = np.clip(b, 0, 255).astype(np.uint8)
    img[..., 3] = 255
    return img


# ---------------------------------------------------------------------------------------------
# transfer-function source strings, exactly as the reference's host generates them


def _ostream_float(v: float) -> str:
    """std::ostream << float with default flags: %g with 6 significant digits."""
    return "%g" % np.float32(v)


def tf_rect_source(rects, stats=(-2000.0, 3000.0, 0.0, 4000.0)) -> str:
    """The string ui::flush_tf builds (app/ui.cpp:160-168) from tf_rect_selection::create_cl_condition
    (app/tf_part.cpp:55-79).  rects: iterable of (min_v, max_v, min_g, max_g, (r,g,b,a) floats 0..1).
    stats = (min_v, max_v, min_g, max_g) of the volume; the gradient clause is emitted only when the
    rectangle is narrower than the stats range (tf_part.cpp:65)."""
    code = "inline bool is_event_gen(short value, short gradient, int4 *color){\n"
    for (min_v, max_v, min_g, max_g, color) in rects:
        code += "  if(value >= " + _ostream_float(min_v) + " && value <= " + _ostream_float(max_v)
        if min_g > stats[2] or max_g < stats[3]:
            code += " && gradient > " + _ostream_float(min_g) + " && gradient < " + _ostream_float(max_g)
        code += ")\n {\n"
        c = [int(np.float32(x) * np.float32(255)) for x in color]
        code += "    int4 tmp_color = {%d,%d,%d,%d};\n" % tuple(c)
        code += "    *color = tmp_color;\n    return true;\n }\n"
    code += "  \n  return false;\n}\n"
    return code


def tf_default_source() -> str:
    """Default selection of the app: rect value in [500,1200], gradient [0,4000] (== stats clip, so
    no gradient clause), colour (1,1,1,1)  (app/ui.cpp:195, app/tf_part.cpp:8-16)."""
    return tf_rect_source([(500.0, 1200.0, 0.0, 4000.0, (1.0, 1.0, 1.0, 1.0))])


def tf_gradient_source() -> str:
    """C3's gradient-dependent TF: exercises the 7-texel step."""
    return tf_rect_source([(500.0, 1200.0, 100.0, 4000.0, (1.0, 0.8, 0.6, 0.5))])


TF_TEST_VALUE_GT_800 = (
    "inline bool is_event_gen(short value, short gradient, uint4 *color){ return (value > 800); }"
)  # tests/sdf/sdf_test.cpp:22, app/sdf_benchmark.cpp:18


# ---------------------------------------------------------------------------------------------
# camera and seeds


def camera_direction(alpha: float, beta: float) -> np.ndarray:
    """Position3D(alpha, beta, 0, {1,0,0}) of app/common.hpp:5-12,44-56 (double math, float store,
    float normalise).  The angles arrive as floats (ui_state::direction_look is float[2])."""
    a, b = float(np.float32(alpha)), float(np.float32(beta))
    v = np.array(
        [math.cos(a) * math.cos(b), -math.sin(b), math.sin(a) * math.cos(b)], dtype=np.float64
    ).astype(np.float32)
    s = float(v[0]) ** 2 + float(v[1]) ** 2 + float(v[2]) ** 2
    ln = np.float32(np.sqrt(np.float32(s)))
    return (v / ln).astype(np.float32)


def default_camera(n: int):
    """Reference default (app/ui.cpp:178): position (-200,200,-200), look (0.9, 6.183), scaled N/512."""
    s = n / 512.0
    pos = np.array([-200.0 * s, 200.0 * s, -200.0 * s], dtype=np.float32)
    return pos, camera_direction(0.9, 6.183)


def glibc_rand(count: int, seed: int = 1):
    """glibc rand() (TYPE_3 additive feedback) -- what the never-srand'ed reference draws its
    per-frame seeds from (app/renderer.cpp:142): 1804289383, 846930886, 1681692777, ..."""
    r = [0] * (344 + count)
    r[0] = seed
    for i in range(1, 31):
        hi, lo = divmod(r[i - 1], 127773)
        w = 16807 * lo - 2836 * hi
        if w < 0:
            w += 2147483647
        r[i] = w
    for i in range(31, 34):
        r[i] = r[i - 31]
    for i in range(34, 344 + count):
        r[i] = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
    return [(r[344 + k] >> 1) for k in range(count)]


# ---------------------------------------------------------------------------------------------
# NRRD (only the header forms app/nrrd_loader.cpp:55-110 accepts)


def read_nrrd(path: str) -> np.ndarray:
    raw = open(path, "rb").read()
    sep = raw.find(b"\n\n")
    if sep < 0:
        raise ValueError("NRRD: no blank line after header")
    header, payload = raw[:sep].decode("ascii", "replace"), raw[sep + 2:]
    fields = {}
    for line in header.splitlines():
        if ":" in line and not line.startswith("#"):
            k, v = line.split(":", 1)
            fields[k.strip()] = v.strip()
    if fields.get("type") not in ("short", "signed short", "int16"):
        raise ValueError("NRRD: only type short is supported")
    sizes = [int(s) for s in fields["sizes"].split()]
    enc = fields.get("encoding", "raw")
    if enc in ("gzip", "gz"):
        payload = zlib.decompress(payload, 15 + 32)
    elif enc != "raw":
        raise ValueError("NRRD: unsupported encoding " + enc)
    dt = np.dtype("<i2") if fields.get("endian", "little") == "little" else np.dtype(">i2")
    n = sizes[0] * sizes[1] * sizes[2]
    vol = np.frombuffer(payload, dtype=dt, count=n).astype(np.int16)
    return vol.reshape(sizes[2], sizes[1], sizes[0])


def write_nrrd(path: str, vol: np.ndarray, use_gzip: bool = False) -> None:
    Z, Y, X = vol.shape
    hdr = (
        "NRRD0004\ntype: short\ndimension: 3\nsizes: %d %d %d\n"
        "space directions: (1,0,0) (0,1,0) (0,0,1)\nendian: little\nencoding: %s\n\n"
        % (X, Y, Z, "gzip" if use_gzip else "raw")
    )
    data = np.ascontiguousarray(vol, dtype="<i2").tobytes()
    if use_gzip:
        data = gzip.compress(data, 6)
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(data)


# ---------------------------------------------------------------------------------------------
# Radiance RGBE (.hdr) writer + the LDR conversion the reference applies through stb_image
# (app/hdre_loader.cpp:11-13: gamma 2.2, scale 1, 4 channels) -- used to test the C++ loader


def float_to_rgbe(rgb: np.ndarray) -> np.ndarray:
    """[h][w][3] float32 -> [h][w][4] uint8 (shared exponent)"""
    m = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), dtype=np.uint8)
    nz = m > 1e-32
    mant, exp = np.frexp(m[nz])
    scale = (mant * 256.0 / m[nz])[:, None]
    out[nz, :3] = np.clip(rgb[nz] * scale, 0, 255).astype(np.uint8)
    out[nz, 3] = (exp + 128).astype(np.uint8)
    return out


def write_hdr(path: str, rgbe: np.ndarray, rle: bool = True) -> None:
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        if not rle or w < 8 or w >= 32768:
            f.write(rgbe.tobytes())
            return
        for j in range(h):
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for k in range(4):
                ch = rgbe[j, :, k]
                i = 0
                while i < w:
                    run = 1
                    while i + run < w and run < 127 and ch[i + run] == ch[i]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(ch[i])]))
                        i += run
                    else:
                        lit = 1
                        while i + lit < w and lit < 128 and not (
                                i + lit + 2 < w and ch[i + lit] == ch[i + lit + 1] == ch[i + lit + 2]):
                            lit += 1
                        f.write(bytes([lit]) + ch[i:i + lit].tobytes())
                        i += lit


def rgbe_to_ldr(rgbe: np.ndarray) -> np.ndarray:
    """what hdre_loader must produce: [h][w][4] uint8"""
    e = rgbe[..., 3].astype(np.int32)
    f = np.ldexp(np.float32(1.0), e - 136).astype(np.float32)
    lin = rgbe[..., :3].astype(np.float32) * f[..., None]
    lin[e == 0] = 0
    inv_gamma = np.float32(1) / np.float32(2.2)
    z = np.power(lin.astype(np.float64), np.float64(inv_gamma)).astype(np.float32) * np.float32(255) + np.float32(0.5)
    z = np.clip(z, 0, 255)
    out = np.empty(rgbe.shape, dtype=np.uint8)
    out[..., :3] = z.astype(np.int32).astype(np.uint8)
    out[..., 3] = 255
    return out
