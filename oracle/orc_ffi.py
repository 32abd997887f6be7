"""ctypes binding of the CPU oracle (oracle/_build/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product never imports this module.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liborc.so")

TF_MAX_RULES = 16
N_COUNTERS = 12
LOCALITY_TOTAL, LOCALITY_HIST_RAY, LOCALITY_HIST_ITEM, LOCALITY_CERT_WRONG = 16 + 256 + 512 + 8, 16, 16 + 256, 16 + 256 + 512
LOCALITY_NAMES = ("fetches", "same_sub4", "same_brick8", "near_8", "near_16", "near_32", "near_64", "uniform4",
                  "step_le_1", "step_le_2", "step_le_8", "step_le_32", "steps", "cert_tried", "cert_granted", "cert_saved")
COUNTER_NAMES = ("n_sdf", "n_vol", "n_env", "n_tok", "n_add", "n_read", "n_hit", "n_step",
                 "n_sdf_primary", "n_vol_primary", "n_env_primary", "n_hit_bounce")
MODE_VOXEL_CACHE, MODE_IMAGE_SPACE = 0, 1
SHADE_LIGHT, SHADE_AO = 0, 1


class TfRule(C.Structure):
    _fields_ = [
        ("v_lo", C.c_int32), ("v_hi", C.c_int32), ("g_lo", C.c_int32), ("g_hi", C.c_int32),
        ("use_gradient", C.c_int32), ("writes_color", C.c_int32), ("terminal", C.c_int32),
        ("color", C.c_int32 * 4),
    ]


class Tf(C.Structure):
    _fields_ = [("n", C.c_int32), ("rules", TfRule * TF_MAX_RULES)]

    def as_tuples(self):
        return [
            (r.v_lo, r.v_hi, r.g_lo, r.g_hi, r.use_gradient, r.writes_color, r.terminal, tuple(r.color))
            for r in list(self.rules)[: self.n]
        ]


class RenderParams(C.Structure):
    _fields_ = [
        ("volume", C.c_void_p), ("X", C.c_int32), ("Y", C.c_int32), ("Z", C.c_int32),
        ("sdf", C.c_void_p),
        ("env", C.c_void_p), ("env_w", C.c_int32), ("env_h", C.c_int32),
        ("cache", C.c_void_p),
        ("frame", C.c_void_p), ("frame_w", C.c_int32), ("frame_h", C.c_int32),
        ("launch_w", C.c_int32), ("launch_h", C.c_int32),
        ("cam_pos", C.c_float * 3), ("cam_dir", C.c_float * 3), ("seed", C.c_int32),
        ("tf", C.POINTER(Tf)),
        ("mode", C.c_int32),
        ("accum", C.c_void_p),
        ("hit_index", C.c_void_p),
        ("contrib", C.c_void_p),
        ("counters", C.c_void_p),
        ("tile_rank", C.c_int32), ("tile_world", C.c_int32),
        ("threads", C.c_int32),
        ("shading", C.c_int32),
        ("locality", C.c_void_p), ("uniform4", C.c_void_p),
        ("macro_free_min", C.c_void_p), ("macro_m", C.c_int32), ("cert_t", C.c_int32),
        ("cert_mode", C.c_int32), ("cert_min_free", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("orc_render.c", "orc_sdf.c", "orc_filter.c", "orc.h", "Makefile")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_cache_len.restype = C.c_int64
        L.orc_cache_len.argtypes = [C.c_int32] * 3
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.POINTER(RenderParams)]
        L.orc_resolve.restype = C.c_int
        L.orc_resolve.argtypes = [C.POINTER(RenderParams)]
        L.orc_sdf_build.restype = C.c_int
        L.orc_sdf_build.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Tf),
                                    C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
        L.orc_buffer_reset.restype = None
        L.orc_buffer_reset.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.orc_bilateral_filter.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.orc_bilateral_filter.restype = None
        L.orc_camera_direction.restype = None
        L.orc_camera_direction.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_float)]
        L.orc_hash.restype = C.c_uint32
        L.orc_hash.argtypes = [C.c_uint32]
        L.orc_hemisphere_reflective.restype = None
        L.orc_hemisphere_reflective.argtypes = [C.POINTER(C.c_float), C.c_int32, C.c_uint32, C.c_uint32,
                                                C.c_float, C.POINTER(C.c_float)]
        L.orc_generate_ray.restype = None
        L.orc_generate_ray.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.POINTER(C.c_float)]
        L.orc_cut.restype = C.c_int
        L.orc_cut.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float),
                              C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_env_texel.restype = None
        L.orc_env_texel.argtypes = [C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        L.orc_tf_eval.restype = C.c_int
        L.orc_tf_eval.argtypes = [C.POINTER(Tf), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        _lib = L
    return _lib


# -------------------------------------------------------------------------------------------------
# Independent parser of the reference's generated TF source (app/ui.cpp:160-168,
# app/tf_part.cpp:55-79, tests/sdf/sdf_test.cpp:22).  The product has its own C++ parser
# (csrc/tf_parse.cpp); tests compare the two.

_NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|inf|nan)"
_CMP = re.compile(r"\s*(value|gradient)\s*(>=|<=|>|<|==)\s*(" + _NUM + r")[fF]?\s*")
_SHORT_MIN, _SHORT_MAX = -32768, 32767


def _apply_cmp(bounds, var, op, lit):
    lo, hi = bounds[var]
    x = float(lit)
    if op == ">=":
        lo = max(lo, math.ceil(x))
    elif op == ">":
        lo = max(lo, math.floor(x) + 1)
    elif op == "<=":
        hi = min(hi, math.floor(x))
    elif op == "<":
        hi = min(hi, math.ceil(x) - 1)
    elif op == "==":
        if x == math.floor(x):
            lo, hi = max(lo, int(x)), min(hi, int(x))
        else:
            lo, hi = 1, 0
    bounds[var] = (int(max(lo, _SHORT_MIN)), int(min(hi, _SHORT_MAX)))


def _parse_cond(cond: str):
    bounds = {"value": (_SHORT_MIN, _SHORT_MAX), "gradient": (_SHORT_MIN, _SHORT_MAX)}
    uses_g = False
    cond = cond.strip()
    while cond.startswith("(") and cond.endswith(")") and cond.count("(") == 1:
        cond = cond[1:-1].strip()
    for term in cond.split("&&"):
        t = term.strip()
        while t.startswith("(") and t.endswith(")"):
            t = t[1:-1].strip()
        m = _CMP.fullmatch(t)
        if not m:
            raise ValueError("unsupported TF condition: %r" % term)
        var, op, lit = m.groups()
        uses_g |= var == "gradient"
        _apply_cmp(bounds, var, op, lit)
    return bounds, uses_g


def parse_tf(source: str) -> Tf:
    body = source[source.index("{") + 1: source.rindex("}")]
    tf = Tf()
    n = 0
    pos = 0
    stmt = re.compile(
        r"\s*(?:if\s*\((?P<cond>[^{]*?)\)\s*\{\s*int4\s+tmp_color\s*=\s*\{\s*(?P<r>-?\d+)\s*,\s*(?P<g>-?\d+)\s*,"
        r"\s*(?P<b>-?\d+)\s*,\s*(?P<a>-?\d+)\s*\}\s*;\s*\*color\s*=\s*tmp_color\s*;\s*return\s+true\s*;\s*\}"
        r"|return\s+(?P<ret>[^;]+);)"
    )
    while pos < len(body) and body[pos:].strip():
        m = stmt.match(body, pos)
        if not m:
            raise ValueError("unsupported TF statement at: %r" % body[pos: pos + 60])
        pos = m.end()
        if m.group("cond") is not None:
            bounds, uses_g = _parse_cond(m.group("cond"))
            r = tf.rules[n]
            r.v_lo, r.v_hi = bounds["value"]
            r.g_lo, r.g_hi = bounds["gradient"]
            r.use_gradient, r.writes_color, r.terminal = int(uses_g), 1, 0
            for k, ch in enumerate("rgba"):
                r.color[k] = int(m.group(ch))
            n += 1
        else:
            ret = m.group("ret").strip()
            if ret == "false":
                break
            if ret == "true":
                bounds, uses_g = _parse_cond("value >= -32768")
            else:
                bounds, uses_g = _parse_cond(ret)
            r = tf.rules[n]
            r.v_lo, r.v_hi = bounds["value"]
            r.g_lo, r.g_hi = bounds["gradient"]
            r.use_gradient, r.writes_color, r.terminal = int(uses_g), 0, 1
            n += 1
            break
        if n >= TF_MAX_RULES:
            raise ValueError("too many TF rules")
    tf.n = n
    return tf


# -------------------------------------------------------------------------------------------------
# numpy-level convenience wrappers


def cache_len(X, Y, Z) -> int:
    return int(lib().orc_cache_len(X, Y, Z))


def bilateral_filter(volume: np.ndarray) -> np.ndarray:
    """volume_filter.cl:5-11 over a (Z, Y, X) int16 array"""
    v = np.ascontiguousarray(volume, dtype=np.int16)
    out = np.empty_like(v)
    Z, Y, X = v.shape
    lib().orc_bilateral_filter(v.ctypes.data, X, Y, Z, out.ctypes.data)
    return out


def sdf_build(volume: np.ndarray, tf: Tf):
    """volume: int16 [Z][Y][X].  Returns (sdf int8 [Z][Y][X], n_launches, per-launch write counts)."""
    vol = np.ascontiguousarray(volume, dtype=np.int16)
    Z, Y, X = vol.shape
    out = np.zeros((Z, Y, X), dtype=np.int8)
    n = C.c_int32(0)
    counts = np.zeros(160, dtype=np.int32)
    rc = lib().orc_sdf_build(vol.ctypes.data, X, Y, Z, C.byref(tf), out.ctypes.data, C.byref(n),
                             counts.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_sdf_build failed: %d" % rc)
    return out, n.value, counts[: n.value].copy()


class Scene:
    """Holds the numpy buffers of one oracle render job (so ctypes pointers stay alive)."""

    def __init__(self, volume, sdf, env, tf, frame_wh, launch_wh=None, mode=MODE_VOXEL_CACHE,
                 tile_rank=0, tile_world=1, threads=1, shading=SHADE_LIGHT):
        self.volume = np.ascontiguousarray(volume, dtype=np.int16)
        self.sdf = np.ascontiguousarray(sdf, dtype=np.int8)
        self.env = np.ascontiguousarray(env, dtype=np.uint8)
        assert self.volume.shape == self.sdf.shape and self.env.ndim == 3 and self.env.shape[2] == 4
        self.tf = tf
        Z, Y, X = self.volume.shape
        self.dims = (X, Y, Z)
        self.frame_w, self.frame_h = frame_wh
        self.launch_w, self.launch_h = launch_wh or frame_wh
        self.mode = mode
        self.shading = shading
        # image-space mode never touches the world-space cache (64 GiB at 2048^3): only allocated when used
        self.cache = np.zeros(cache_len(X, Y, Z), dtype=np.uint16) if (mode == MODE_VOXEL_CACHE or shading == SHADE_AO) else None
        self.frame = np.zeros((self.frame_h, self.frame_w, 4), dtype=np.uint8)
        npx = self.launch_w * self.launch_h
        self.accum = np.zeros((self.launch_h, self.launch_w, 4), dtype=np.float32)
        self.hit_index = np.full(npx, -1, dtype=np.int64)
        self.contrib = np.zeros((npx, 4), dtype=np.uint32)
        self.counters = np.zeros(N_COUNTERS, dtype=np.uint64)
        self.tile_rank, self.tile_world, self.threads = tile_rank, tile_world, threads
        self.locality = None   # set to np.zeros(LOCALITY_TOTAL, uint64) to collect the bounce phase's step-locality counters
        self.uniform4 = None   # optional uint8 [ceil(Z/4)][ceil(Y/4)][ceil(X/4)] flags for the same instrumentation
        self.macro_free_min, self.macro_m, self.cert_t = None, 0, 0   # exit-certificate experiment (orc.h)
        self.cert_mode, self.cert_min_free = 0, 1

    def _params(self, cam_pos, cam_dir, seed):
        X, Y, Z = self.dims
        p = RenderParams()
        p.volume, p.X, p.Y, p.Z = self.volume.ctypes.data, X, Y, Z
        p.sdf = self.sdf.ctypes.data
        p.env, p.env_w, p.env_h = self.env.ctypes.data, self.env.shape[1], self.env.shape[0]
        p.cache = self.cache.ctypes.data if self.cache is not None else None
        p.frame, p.frame_w, p.frame_h = self.frame.ctypes.data, self.frame_w, self.frame_h
        p.launch_w, p.launch_h = self.launch_w, self.launch_h
        for k in range(3):
            p.cam_pos[k] = float(cam_pos[k])
            p.cam_dir[k] = float(cam_dir[k])
        p.seed = int(seed)
        p.tf = C.pointer(self.tf)
        p.mode = self.mode
        p.accum = self.accum.ctypes.data
        p.hit_index = self.hit_index.ctypes.data
        p.contrib = self.contrib.ctypes.data
        p.counters = self.counters.ctypes.data
        p.tile_rank, p.tile_world, p.threads = self.tile_rank, self.tile_world, self.threads
        p.shading = self.shading
        p.locality = self.locality.ctypes.data if self.locality is not None else None
        p.uniform4 = self.uniform4.ctypes.data if self.uniform4 is not None else None
        p.macro_free_min = self.macro_free_min.ctypes.data if self.macro_free_min is not None else None
        p.macro_m, p.cert_t = int(self.macro_m), int(self.cert_t)
        p.cert_mode, p.cert_min_free = int(self.cert_mode), int(self.cert_min_free)
        return p

    def render(self, cam_pos, cam_dir, seed):
        p = self._params(cam_pos, cam_dir, seed)
        rc = lib().orc_render(C.byref(p))
        if rc != 0:
            raise RuntimeError("orc_render failed: %d" % rc)

    def resolve(self, cam_pos, cam_dir, seed=0):
        p = self._params(cam_pos, cam_dir, seed)
        rc = lib().orc_resolve(C.byref(p))
        if rc != 0:
            raise RuntimeError("orc_resolve failed: %d" % rc)

    def reset(self):
        if self.cache is not None:
            self.cache[:] = 0
        self.accum[:] = 0
        self.counters[:] = 0

    def counter_dict(self):
        return {k: int(v) for k, v in zip(COUNTER_NAMES, self.counters) if k != "reserved"}


def algorithmic_bytes(counters: dict, samples: int) -> float:
    """B = N_sdf + 2 N_vol + 4 N_env + 4 N_tok + 8 N_add + 8 N_read + 4 per sample."""
    c = counters
    total = (c["n_sdf"] + 2 * c["n_vol"] + 4 * c["n_env"] + 4 * c["n_tok"] + 8 * c["n_add"]
             + 8 * c["n_read"] + 4 * samples)
    return total / float(samples)
