"""ctypes binding of the product's C ABI (include/clwh.h, libclwhip.so).

There is no CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLWH_LIBRARY: another build of the same library (tools/ experiments: timing probes, tuning variants)
LIB_PATH = os.environ.get("CLWH_LIBRARY") or os.path.join(_HERE, "libclwhip.so")

OK = 0
ELEM_S8, ELEM_S16, ELEM_S32, ELEM_U8, ELEM_U16, ELEM_U32, ELEM_F32 = range(7)
ARG_MEM, ARG_I32, ARG_U32, ARG_F32, ARG_I64, ARG_U64, ARG_F64 = range(7)
ACCUM_VOXEL_CACHE, ACCUM_IMAGE_SPACE = 0, 1
DERIVED_SCENE, DERIVED_CAMERA = 1, 2
SHADE_LIGHT, SHADE_AO = 0, 1
TIMERS = ("bounce", "primary", "fixup", "resolve", "repack", "ao")
MAX_SEEDS = 64
TF_MAX_RULES = 16

_ELEM_OF_DTYPE = {
    np.dtype(np.int8): ELEM_S8, np.dtype(np.int16): ELEM_S16, np.dtype(np.int32): ELEM_S32,
    np.dtype(np.uint8): ELEM_U8, np.dtype(np.uint16): ELEM_U16, np.dtype(np.uint32): ELEM_U32,
    np.dtype(np.float32): ELEM_F32,
}


class ClwhError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        name = lib().clwh_strerror(status).decode()
        super().__init__("%s failed: %s (hip error %d)" % (where, name, lib().clwh_last_hip_error()))


class _ArgValue(C.Union):
    _fields_ = [("mem", C.c_void_p), ("i32", C.c_int32), ("u32", C.c_uint32), ("f32", C.c_float),
                ("i64", C.c_int64), ("u64", C.c_uint64), ("f64", C.c_double)]


class Arg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("v", _ArgValue)]


class TfRule(C.Structure):
    _fields_ = [("v_lo", C.c_int32), ("v_hi", C.c_int32), ("g_lo", C.c_int32), ("g_hi", C.c_int32),
                ("use_gradient", C.c_int32), ("writes_color", C.c_int32), ("terminal", C.c_int32),
                ("color", C.c_int32 * 4)]


class Tf(C.Structure):
    _fields_ = [("n", C.c_int32), ("rules", TfRule * TF_MAX_RULES)]

    def as_tuples(self):
        return [(r.v_lo, r.v_hi, r.g_lo, r.g_hi, r.use_gradient, r.writes_color, r.terminal, tuple(r.color))
                for r in list(self.rules)[: self.n]]


class RenderDesc(C.Structure):
    _fields_ = [
        ("frame", C.c_void_p), ("volume", C.c_void_p), ("sdf", C.c_void_p), ("env", C.c_void_p),
        ("buffer_volume", C.c_void_p),
        ("cam_pos", C.c_float * 3), ("cam_dir", C.c_float * 3), ("seed", C.c_int32),
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("accum_mode", C.c_int32), ("accum", C.c_void_p),
        ("tile_rank", C.c_int32), ("tile_world", C.c_int32),
        ("write_frame", C.c_int32),
        ("hit_index", C.c_void_p), ("contrib", C.c_void_p),
        ("n_seeds", C.c_int32), ("seeds", C.c_int32 * 64),
        ("shading", C.c_int32), ("resolve_only", C.c_int32),
    ]


# every symbol include/clwh.h declares: (name, restype, argtypes)
_SIZE3 = C.POINTER(C.c_size_t)
_PROTOTYPES = [
    ("clwh_ctx_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("clwh_ctx_create_on_stream", C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("clwh_ctx_destroy", C.c_int, [C.c_void_p]),
    ("clwh_ctx_finish", C.c_int, [C.c_void_p]),
    ("clwh_ctx_stream", C.c_void_p, [C.c_void_p]),
    ("clwh_ctx_device", C.c_int, [C.c_void_p]),
    ("clwh_mem_create", C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    ("clwh_mem_wrap", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("clwh_image_create", C.c_int, [C.c_void_p, _SIZE3, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("clwh_image_wrap", C.c_int, [C.c_void_p, C.c_void_p, _SIZE3, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("clwh_mem_push", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("clwh_mem_pull", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("clwh_mem_release", C.c_int, [C.c_void_p]),
    ("clwh_host_register", C.c_int, [C.c_void_p, C.c_size_t]),
    ("clwh_host_unregister", C.c_int, [C.c_void_p]),
    ("clwh_mem_device_ptr", C.c_void_p, [C.c_void_p]),
    ("clwh_mem_size", C.c_size_t, [C.c_void_p]),
    ("clwh_mem_mark_dirty", C.c_int, [C.c_void_p]),
    ("clwh_kernel_get", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    ("clwh_kernel_release", C.c_int, [C.c_void_p]),
    ("clwh_launch", C.c_int, [C.c_void_p, _SIZE3, _SIZE3, C.POINTER(Arg), C.c_int]),
    ("clwh_render", C.c_int, [C.c_void_p, C.POINTER(RenderDesc)]),
    ("clwh_cache_len", C.c_int64, [C.c_uint32, C.c_uint32, C.c_uint32]),
    ("clwh_accum_len", C.c_int64, [C.c_uint32, C.c_uint32, C.c_int32]),
    ("clwh_accum_resolve", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("clwh_accum_resolve_tiles", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("clwh_frame_from_tiles", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("clwh_ctx_invalidate_derived", C.c_int, [C.c_void_p, C.c_int]),
    ("clwh_ctx_scene_info", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    ("clwh_sdf_build", C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int32)]),
    ("clwh_buffer_reset", C.c_int, [C.c_void_p, C.c_void_p]),
    ("clwh_cache_exchange_plan", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    ("clwh_cache_apply_contributions", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    ("clwh_cache_exchange_plan_release", C.c_int, [C.c_void_p]),
    ("clwh_tf_parse", C.c_int, [C.c_char_p, C.POINTER(Tf)]),
    ("clwh_debug_float_conversions", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("clwh_debug_wave_min", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("clwh_strerror", C.c_char_p, [C.c_int]),
    ("clwh_last_hip_error", C.c_int, []),
    ("clwh_version", C.c_char_p, []),
    ("clwh_ctx_acquire_from", C.c_int, [C.c_void_p, C.c_void_p]),
    ("clwh_ctx_release_to", C.c_int, [C.c_void_p, C.c_void_p]),
    ("clwh_ctx_set_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("clwh_ctx_timing_read", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    ("clwh_ctx_timing_read_all", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int32]),
]
EXPORTED_SYMBOLS = [p[0] for p in _PROTOTYPES]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libclwhip.so is missing (%s): build it with `python -m cl_volume_renderer_amd.build`; "
                "there is no CPU fallback for the product path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in _PROTOTYPES:
            if not hasattr(L, name) and os.environ.get("CLWH_LIBRARY"):
                continue  # an older build named explicitly for an A/B timing run (tools/ab_bounce.sh)
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _check(status, where):
    if status != OK:
        raise ClwhError(status, where)


def parse_tf(source: str) -> Tf:
    tf = Tf()
    _check(lib().clwh_tf_parse(source.encode(), C.byref(tf)), "clwh_tf_parse")
    return tf


def cache_len(X, Y, Z) -> int:
    return int(lib().clwh_cache_len(X, Y, Z))


def accum_len(w, h, world=1) -> int:
    return int(lib().clwh_accum_len(w, h, world))


class Mem:
    def __init__(self, ctx, handle, nbytes, dtype=None, shape=None):
        self.ctx, self.h, self.nbytes, self.dtype, self.shape = ctx, C.c_void_p(handle), nbytes, dtype, shape

    def push(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        _check(lib().clwh_mem_push(self.ctx.h, self.h, a.ctypes.data, a.nbytes), "clwh_mem_push")

    def pull(self, dtype=None, shape=None) -> np.ndarray:
        dtype = np.dtype(dtype or self.dtype or np.uint8)
        out = np.empty(self.nbytes // dtype.itemsize, dtype=dtype)
        _check(lib().clwh_mem_pull(self.ctx.h, self.h, out.ctypes.data, out.nbytes), "clwh_mem_pull")
        shape = shape or self.shape
        return out.reshape(shape) if shape else out

    @property
    def device_ptr(self) -> int:
        return int(lib().clwh_mem_device_ptr(self.h) or 0)

    def release(self):
        if self.h:
            _check(lib().clwh_mem_release(self.h), "clwh_mem_release")
            self.h = None


class Kernel:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, C.c_void_p(handle)

    def launch(self, global_size, local_size, *args):
        g = (C.c_size_t * 3)(*(list(global_size) + [0, 0, 0])[:3])
        l = (C.c_size_t * 3)(*(list(local_size) + [0, 0, 0])[:3])
        arr = (Arg * max(len(args), 1))()
        for i, a in enumerate(args):
            if isinstance(a, Mem):
                arr[i].kind, arr[i].v.mem = ARG_MEM, a.h.value
            elif isinstance(a, (float, np.floating)):
                arr[i].kind, arr[i].v.f32 = ARG_F32, float(a)
            elif isinstance(a, np.unsignedinteger):
                arr[i].kind, arr[i].v.u32 = ARG_U32, int(a)
            elif isinstance(a, (int, np.integer)):
                arr[i].kind, arr[i].v.i32 = ARG_I32, int(a)
            else:
                raise TypeError("unsupported kernel argument %r" % (a,))
        _check(lib().clwh_launch(self.h, g, l, arr, len(args)), "clwh_launch")

    def render(self, *, frame, volume, sdf, env, cam_pos, cam_dir, seed, width, height, buffer_volume=None,
               accum=None, mode=ACCUM_VOXEL_CACHE, tile_rank=0, tile_world=1, write_frame=True,
               hit_index=None, contrib=None, seeds=None, shading=SHADE_LIGHT, resolve_only=False):
        d = RenderDesc()
        d.frame = frame.h if frame is not None else None
        d.volume, d.sdf, d.env = volume.h, sdf.h, env.h
        d.buffer_volume = buffer_volume.h if buffer_volume is not None else None
        for k in range(3):
            d.cam_pos[k] = float(cam_pos[k])
            d.cam_dir[k] = float(cam_dir[k])
        d.seed, d.width, d.height = int(seed), int(width), int(height)
        d.accum_mode = mode
        d.accum = accum.h if accum is not None else None
        d.tile_rank, d.tile_world = tile_rank, tile_world
        d.write_frame = 1 if write_frame else 0
        d.hit_index = hit_index.h if hit_index is not None else None
        d.contrib = contrib.h if contrib is not None else None
        d.shading = shading
        d.resolve_only = 1 if resolve_only else 0
        if seeds is not None:
            d.n_seeds = len(seeds)
            for i, sd in enumerate(seeds):
                d.seeds[i] = int(sd)
        _check(lib().clwh_render(self.h, C.byref(d)), "clwh_render")

    def release(self):
        if self.h:
            _check(lib().clwh_kernel_release(self.h), "clwh_kernel_release")
            self.h = None


class ExchangePlan:
    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, C.c_void_p(handle)

    def apply(self, buffer_volume: Mem, rgb: Mem, rgb_stride=3):
        """one pass's contributions (int32[n][rgb_stride], the plan's order) into the cache under the 256-token rule"""
        _check(lib().clwh_cache_apply_contributions(self.ctx.h, self.h, buffer_volume.h, rgb.h, rgb_stride),
               "clwh_cache_apply_contributions")

    def release(self):
        if self.h:
            _check(lib().clwh_cache_exchange_plan_release(self.h), "clwh_cache_exchange_plan_release")
            self.h = None


class Context:
    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        if stream is None:
            _check(lib().clwh_ctx_create(device, C.byref(h)), "clwh_ctx_create")
        else:
            _check(lib().clwh_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h)),
                   "clwh_ctx_create_on_stream")
        self.h = h

    def buffer(self, nbytes, dtype=None, shape=None) -> Mem:
        h = C.c_void_p()
        _check(lib().clwh_mem_create(self.h, nbytes, 0, C.byref(h)), "clwh_mem_create")
        return Mem(self, h.value, nbytes, dtype, shape)

    def buffer_from(self, arr: np.ndarray) -> Mem:
        a = np.ascontiguousarray(arr)
        m = self.buffer(a.nbytes, a.dtype, a.shape)
        m.push(a)
        return m

    def wrap(self, device_ptr: int, nbytes: int, dtype=None, shape=None) -> Mem:
        h = C.c_void_p()
        _check(lib().clwh_mem_wrap(self.h, C.c_void_p(device_ptr), nbytes, C.byref(h)), "clwh_mem_wrap")
        return Mem(self, h.value, nbytes, dtype, shape)

    def image(self, dims, channels, dtype, shape=None) -> Mem:
        dtype = np.dtype(dtype)
        d = (C.c_size_t * 3)(*(list(dims) + [1, 1, 1])[:3])
        h = C.c_void_p()
        _check(lib().clwh_image_create(self.h, d, channels, _ELEM_OF_DTYPE[dtype], 0, C.byref(h)),
               "clwh_image_create")
        n = int(np.prod([max(int(x), 1) for x in list(dims)[:3]])) * channels * dtype.itemsize
        return Mem(self, h.value, n, dtype, shape)

    def image_from(self, arr: np.ndarray, channels=1) -> Mem:
        """arr: [z][y][x] (3-D, 1 channel) or [h][w][c] (2-D, c channels)."""
        a = np.ascontiguousarray(arr)
        if channels == 1:
            dims = list(a.shape[::-1])
        else:
            assert a.shape[-1] == channels
            dims = list(a.shape[:-1][::-1])
        m = self.image(dims, channels, a.dtype, a.shape)
        m.push(a)
        return m

    def kernel(self, file: str, entry: str, prepend: str = "") -> Kernel:
        h = C.c_void_p()
        _check(lib().clwh_kernel_get(self.h, file.encode(), entry.encode(), prepend.encode(), C.byref(h)),
               "clwh_kernel_get")
        return Kernel(self, h.value)

    def sdf_build(self, volume: Mem, tf_source: str, sdf: Mem) -> int:
        n = C.c_int32(0)
        _check(lib().clwh_sdf_build(self.h, volume.h, tf_source.encode(), sdf.h, C.byref(n)), "clwh_sdf_build")
        return n.value

    def buffer_reset(self, buffer_volume: Mem):
        _check(lib().clwh_buffer_reset(self.h, buffer_volume.h), "clwh_buffer_reset")

    def exchange_plan(self, entries: Mem, n: int) -> "ExchangePlan":
        """group the listed cache entries (int64[n], (rank, pixel) order) by voxel: once per camera"""
        h = C.c_void_p()
        _check(lib().clwh_cache_exchange_plan(self.h, entries.h, int(n), C.byref(h)), "clwh_cache_exchange_plan")
        return ExchangePlan(self, h.value)

    def accum_resolve(self, accum_all: Mem, tile_world, width, height, frame: Mem, env: Mem, cam_pos, cam_dir):
        p = (C.c_float * 3)(*[float(x) for x in cam_pos])
        d = (C.c_float * 3)(*[float(x) for x in cam_dir])
        _check(lib().clwh_accum_resolve(self.h, accum_all.h, tile_world, width, height, frame.h, env.h, p, d),
               "clwh_accum_resolve")

    def accum_resolve_tiles(self, accum: Mem, tile_rank, tile_world, width, height, tiles_rgba8: Mem, env: Mem, cam_pos, cam_dir):
        """this rank's tiles -> RGBA8, tile-major (4 bytes per pixel to exchange instead of 16)"""
        p = (C.c_float * 3)(*[float(x) for x in cam_pos])
        d = (C.c_float * 3)(*[float(x) for x in cam_dir])
        _check(lib().clwh_accum_resolve_tiles(self.h, accum.h, tile_rank, tile_world, width, height, tiles_rgba8.h, env.h, p, d),
               "clwh_accum_resolve_tiles")

    def frame_from_tiles(self, tiles_all: Mem, tile_world, width, height, frame: Mem):
        _check(lib().clwh_frame_from_tiles(self.h, tiles_all.h, tile_world, width, height, frame.h), "clwh_frame_from_tiles")

    def invalidate_derived(self, scene=True, camera=True):
        what = (DERIVED_SCENE if scene else 0) | (DERIVED_CAMERA if camera else 0)
        _check(lib().clwh_ctx_invalidate_derived(self.h, what), "clwh_ctx_invalidate_derived")

    def scene_info(self):
        """(id, bytes, holders) of the derived scene data this context renders from"""
        sid, nb, holders = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        _check(lib().clwh_ctx_scene_info(self.h, C.byref(sid), C.byref(nb), C.byref(holders)), "clwh_ctx_scene_info")
        return int(sid.value), int(nb.value), int(holders.value)

    def image_wrap(self, device_ptr: int, dims, channels, dtype, shape=None) -> Mem:
        """adopt device memory somebody else allocated (a graphics-interop mapping, a torch tensor) as an image"""
        dtype = np.dtype(dtype)
        d = (C.c_size_t * 3)(*(list(dims) + [1, 1, 1])[:3])
        h = C.c_void_p()
        _check(lib().clwh_image_wrap(self.h, C.c_void_p(device_ptr), d, channels, _ELEM_OF_DTYPE[dtype], C.byref(h)),
               "clwh_image_wrap")
        n = int(np.prod([max(int(x), 1) for x in list(dims)[:3]])) * channels * dtype.itemsize
        return Mem(self, h.value, n, dtype, shape)

    def acquire_from(self, stream: int):
        """the context's later work waits for what is queued on `stream` now (display done with the old frame)"""
        _check(lib().clwh_ctx_acquire_from(self.h, C.c_void_p(stream)), "clwh_ctx_acquire_from")

    def release_to(self, stream: int):
        """later work on `stream` waits for what is queued on the context's stream now (frame complete)"""
        _check(lib().clwh_ctx_release_to(self.h, C.c_void_p(stream)), "clwh_ctx_release_to")

    def finish(self):
        _check(lib().clwh_ctx_finish(self.h), "clwh_ctx_finish")

    def set_timing(self, enabled=True):
        _check(lib().clwh_ctx_set_timing(self.h, 1 if enabled else 0), "clwh_ctx_set_timing")

    def timing_read(self):
        """(total ms, launches) of the dominant render kernel since the last read (HIP events)."""
        ms, n = C.c_float(0), C.c_int32(0)
        _check(lib().clwh_ctx_timing_read(self.h, C.byref(ms), C.byref(n)), "clwh_ctx_timing_read")
        return float(ms.value), int(n.value)

    def timing_read_all(self):
        """{kernel: (total ms, launches)} per kernel of the render path since the last read (HIP events)."""
        n = len(TIMERS)
        ms, cnt = (C.c_float * n)(), (C.c_int32 * n)()
        _check(lib().clwh_ctx_timing_read_all(self.h, ms, cnt, n), "clwh_ctx_timing_read_all")
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(TIMERS)}

    @property
    def stream(self) -> int:
        return int(lib().clwh_ctx_stream(self.h) or 0)

    def destroy(self):
        if self.h:
            _check(lib().clwh_ctx_destroy(self.h), "clwh_ctx_destroy")
            self.h = None
