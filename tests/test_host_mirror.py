"""The C++ host mirror of the reference's application-facing classes (renderer : frame_emitter,
reference_volume, signed_distance_field, env_map over include/clw_*.hpp), driven through the small
extern "C" facade of app/host_c_api.cpp in the order ui::run drives them (app/ui.cpp:170-199, 296)."""
import ctypes as C
import os

import numpy as np
import pytest

from cl_volume_renderer_amd import scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(ROOT, "cl_volume_renderer_amd", "libclvr_host.so")


def _host():
    L = C.CDLL(HOST_LIB)
    L.clvr_host_create.restype = C.c_void_p
    L.clvr_host_destroy.argtypes = [C.c_void_p]
    L.clvr_host_load.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_uint, C.c_uint]
    L.clvr_host_flush.argtypes = [C.c_void_p, C.c_char_p]
    L.clvr_host_render_frame.restype = C.c_void_p
    L.clvr_host_render_frame.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int,
                                         C.c_int, C.POINTER(C.c_int)]
    L.clvr_host_cache_len.restype = C.c_size_t
    L.clvr_host_cache_len.argtypes = [C.c_void_p]
    L.clvr_host_pull_cache.argtypes = [C.c_void_p, C.c_void_p]
    L.clvr_host_sdf_len.restype = C.c_size_t
    L.clvr_host_sdf_len.argtypes = [C.c_void_p]
    L.clvr_host_pull_sdf.argtypes = [C.c_void_p, C.c_void_p]
    L.clvr_host_sdf_layers.restype = C.c_int
    L.clvr_host_sdf_layers.argtypes = [C.c_void_p]
    L.clvr_host_camera_direction.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float)]
    L.clvr_host_volume_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.clvr_host_set_clipping.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    L.clvr_host_filter.argtypes = [C.c_void_p]
    L.clvr_host_render_tf.restype = C.c_void_p
    L.clvr_host_render_tf.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
    return L


def test_host_library_loads_and_position3d_matches_the_oracle(orc):
    L = _host()
    out = (C.c_float * 3)()
    want = (C.c_float * 3)()
    for a, b in [(0.9, 6.183), (0.0, 0.0), (2.5, 1.1), (-0.7, 3.3)]:
        L.clvr_host_camera_direction(a, b, out)
        orc.lib().orc_camera_direction(float(np.float32(a)), float(np.float32(b)), want)
        assert list(out) == list(want)


@pytest.mark.gpu
def test_renderer_frame_emitter_sequence_matches_oracle(orc):
    """load -> flush_tf/flush_changes -> N x render_frame, with the renderer drawing its own seeds from
    std::rand() (never seeded: 1804289383, 846930886, ...), compared with the oracle given the same seeds."""
    L = _host()
    n = 48
    vol = scene.phantom(n)
    env = scene.env_map(256, 128)
    tf = scene.tf_default_source()
    pos = np.array([-110.0, 150.0, -110.0], np.float32)
    look = np.array([0.1, 6.6], np.float32)
    W, H = 512, 256           # launch size (state.width/height); the frame image stays 2048x1024

    h = L.clvr_host_create()
    L.clvr_host_load(h, vol.ctypes.data, n, n, n, env.ctypes.data, env.shape[1], env.shape[0])
    L.clvr_host_flush(h, tf.encode())
    # the state a fresh process has (app/renderer.cpp:142 never calls srand).  Set AFTER the context exists: the
    # ROCm runtime draws from rand() while it initialises, which would shift the renderer's seed sequence.
    libc = C.CDLL("libc.so.6")
    libc.srand(1)

    sdf = np.empty(L.clvr_host_sdf_len(h), np.int8)
    L.clvr_host_pull_sdf(h, sdf.ctypes.data)
    want_sdf, n_layers, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    assert np.array_equal(sdf.reshape(vol.shape), want_sdf)
    assert L.clvr_host_sdf_layers(h) == n_layers

    o = orc.Scene(vol, want_sdf, env, orc.parse_tf(tf), (2048, 1024), (W, H))
    d = scene.camera_direction(look[0], look[1])
    changed = C.c_int(0)
    seeds = scene.glibc_rand(3)
    frame_ptr = None
    for s in seeds:
        frame_ptr = L.clvr_host_render_frame(h, pos.ctypes.data_as(C.POINTER(C.c_float)),
                                             look.ctypes.data_as(C.POINTER(C.c_float)), W, H, 1, C.byref(changed))
        assert changed.value == 1
        o.render(pos, d, s)
    # unchanged camera: the cached host frame comes back and nothing is rendered (renderer.cpp:134-135)
    again = L.clvr_host_render_frame(h, pos.ctypes.data_as(C.POINTER(C.c_float)),
                                     look.ctypes.data_as(C.POINTER(C.c_float)), W, H, 0, C.byref(changed))
    assert changed.value == 0 and again == frame_ptr

    cache = np.empty(L.clvr_host_cache_len(h), np.uint16)
    L.clvr_host_pull_cache(h, cache.ctypes.data)
    assert np.array_equal(cache, o.cache)
    assert 0 < cache.reshape(-1, 4)[:, 3].max() < 256  # below the token cap the cache is order-independent
    o.resolve(pos, d)
    frame = np.ctypeslib.as_array(C.cast(frame_ptr, C.POINTER(C.c_uint8)), shape=(1024, 2048, 4))
    assert np.array_equal(frame[:H, :W], o.frame[:H, :W])
    L.clvr_host_destroy(h)


@pytest.mark.gpu
def test_reference_volume_stats_and_clipping(orc):
    """reference_volume: fetch_stats at construction with the app's clip ranges (ui.cpp:187-188), then
    set_clipping -> flush -> render on the cropped copy, against the oracle on the cropped numpy array."""
    from oracle import orc_volume

    L = _host()
    n = 40
    vol = scene.phantom(n)
    env = scene.env_map(128, 64)
    tf = scene.tf_default_source()
    h = L.clvr_host_create()
    L.clvr_host_load(h, vol.ctypes.data, n, n, n, env.ctypes.data, env.shape[1], env.shape[0])
    st = (C.c_float * 4)()
    L.clvr_host_volume_stats(h, st)
    want = orc_volume.fetch_stats(vol)
    assert list(st) == [max(-2000, want[0]), min(3000, want[1]), max(0, want[2]), min(4000, want[3])]

    lo, hi = (4, 2, 6), (36, 34, 38)
    L.clvr_host_set_clipping(h, (C.c_uint * 3)(*lo), (C.c_uint * 3)(*hi))
    L.clvr_host_flush(h, tf.encode())
    crop = np.ascontiguousarray(vol[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]])
    sdf = np.empty(L.clvr_host_sdf_len(h), np.int8)
    L.clvr_host_pull_sdf(h, sdf.ctypes.data)
    want_sdf, _, _ = orc.sdf_build(crop, orc.parse_tf(tf))
    assert np.array_equal(sdf.reshape(crop.shape), want_sdf)

    libc = C.CDLL("libc.so.6")
    libc.srand(1)
    pos = np.array([-110.0, 150.0, -110.0], np.float32)
    look = np.array([0.1, 6.6], np.float32)
    W, H = 512, 256
    changed = C.c_int(0)
    L.clvr_host_render_frame(h, pos.ctypes.data_as(C.POINTER(C.c_float)), look.ctypes.data_as(C.POINTER(C.c_float)),
                             W, H, 1, C.byref(changed))
    o = orc.Scene(crop, want_sdf, env, orc.parse_tf(tf), (2048, 1024), (W, H))
    o.render(pos, scene.camera_direction(look[0], look[1]), scene.glibc_rand(1)[0])
    cache = np.empty(L.clvr_host_cache_len(h), np.uint16)
    L.clvr_host_pull_cache(h, cache.ctypes.data)
    assert np.array_equal(cache, o.cache)
    assert 0 < cache.reshape(-1, 4)[:, 3].max() < 256  # below the token cap the cache is order-independent
    L.clvr_host_destroy(h)


@pytest.mark.gpu
def test_reference_volume_filter_replaces_the_device_volume(orc):
    """reference_volume::filter (app/reference_volume.cpp:70-80): after "Apply Filter" + flush, SDF and render
    work on the bilaterally filtered volume"""
    L = _host()
    n = 48
    vol = scene.phantom(n)
    env = scene.env_map(128, 64)
    # a threshold inside the phantom's noise band, so that filtering changes which voxels are surface
    tf = scene.tf_default_source().replace("value >= 500", "value >= -1000")
    assert "value >= -1000" in tf
    h = L.clvr_host_create()
    L.clvr_host_load(h, vol.ctypes.data, n, n, n, env.ctypes.data, env.shape[1], env.shape[0])
    L.clvr_host_filter(h)
    L.clvr_host_flush(h, tf.encode())
    filtered = orc.bilateral_filter(vol)
    assert not np.array_equal(filtered, vol)
    sdf = np.empty(L.clvr_host_sdf_len(h), np.int8)
    L.clvr_host_pull_sdf(h, sdf.ctypes.data)
    want_sdf, _, _ = orc.sdf_build(filtered, orc.parse_tf(tf))
    assert np.array_equal(sdf.reshape(vol.shape), want_sdf)
    assert not np.array_equal(want_sdf, orc.sdf_build(vol, orc.parse_tf(tf))[0])
    libc = C.CDLL("libc.so.6")
    libc.srand(1)
    pos = np.array([-110.0, 150.0, -110.0], np.float32)
    look = np.array([0.1, 6.6], np.float32)
    W, H = 512, 256
    changed = C.c_int(0)
    L.clvr_host_render_frame(h, pos.ctypes.data_as(C.POINTER(C.c_float)), look.ctypes.data_as(C.POINTER(C.c_float)),
                             W, H, 1, C.byref(changed))
    o = orc.Scene(filtered, want_sdf, env, orc.parse_tf(tf), (2048, 1024), (W, H))
    o.render(pos, scene.camera_direction(look[0], look[1]), scene.glibc_rand(1)[0])
    cache = np.empty(L.clvr_host_cache_len(h), np.uint16)
    L.clvr_host_pull_cache(h, cache.ctypes.data)
    assert np.array_equal(cache, o.cache)
    assert 0 < cache.reshape(-1, 4)[:, 3].max() < 256  # below the token cap the cache is order-independent
    L.clvr_host_destroy(h)


@pytest.mark.gpu
def test_render_tf_histogram_texture(orc):
    """frame_emitter::render_tf (ui.cpp:151-158 -> renderer.cpp:45-124): the 2-D histogram texture."""
    from oracle import orc_volume

    L = _host()
    n = 40
    vol = scene.phantom(n)
    env = scene.env_map(64, 32)
    h = L.clvr_host_create()
    L.clvr_host_load(h, vol.ctypes.data, n, n, n, env.ctypes.data, env.shape[1], env.shape[0])
    st = (C.c_float * 4)()
    L.clvr_host_volume_stats(h, st)
    w = hgt = 100
    ptr = L.clvr_host_render_tf(h, w, hgt)
    got = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(hgt, w, 4)).copy()
    want = orc_volume.render_tf(vol, w, hgt, st[0], st[1], st[2], st[3])
    assert (want[..., 0] > 0).sum() > 20
    assert np.array_equal(got, want)
    L.clvr_host_destroy(h)


def test_nrrd_loader_reads_the_reference_test_block(tmp_path):
    """app/nrrd_loader.cpp mirror (CPU only): gzip payload of the reference's testdata.nrrd and a raw file
    written by this project's writer decode to the same voxels as the Python reader."""
    L = _host()
    L.clvr_host_nrrd_probe.restype = C.c_longlong
    L.clvr_host_nrrd_probe.argtypes = [C.c_char_p, C.POINTER(C.c_uint), C.POINTER(C.c_longlong)]
    dims = (C.c_uint * 3)()
    checksum = C.c_longlong(0)
    path = os.path.join(ROOT, "tests", "golden", "sdf_testdata.nrrd")
    n = L.clvr_host_nrrd_probe(path.encode(), dims, C.byref(checksum))
    vol = scene.read_nrrd(path)
    assert list(dims) == [38, 35, 38] and n == 50540
    assert checksum.value == int(vol.astype(np.int64).sum()) == 26559114
    raw = str(tmp_path / "raw.nrrd")
    v2 = scene.phantom(20, dims=(20, 12, 9))
    scene.write_nrrd(raw, v2, use_gzip=False)
    assert L.clvr_host_nrrd_probe(raw.encode(), dims, C.byref(checksum)) == v2.size
    assert list(dims) == [20, 12, 9] and checksum.value == int(v2.astype(np.int64).sum())
    gz = str(tmp_path / "gz.nrrd")
    scene.write_nrrd(gz, v2, use_gzip=True)
    assert L.clvr_host_nrrd_probe(gz.encode(), dims, C.byref(checksum)) == v2.size
    assert checksum.value == int(v2.astype(np.int64).sum())


@pytest.mark.gpu
def test_reference_sdf_test_program():
    """the reference's only test, tests/sdf/sdf_test.cpp, as a program over this project's headers"""
    import subprocess

    exe = os.path.join(ROOT, "cl_volume_renderer_amd", "sdf_test")
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "sdf_testdata.nrrd"),
                          os.path.join(ROOT, "tests", "golden", "sdf_values.x")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "EVERYTHING FINE" in out.stdout and "13 layers" in out.stdout


def _png_bytes(img, color, depth, interlace=False, palette=None, trns=None, filters=(0, 1, 2, 3, 4)):
    """a small PNG writer for the reader's test: img [h][w][channels] of sample values at `depth` bits"""
    import struct
    import zlib

    h, w, ch = img.shape

    def pack_row(row):  # row [w][ch] -> bytes at the bit depth
        flat = row.reshape(-1).astype(np.uint32)
        if depth == 16:
            return b"".join(struct.pack(">H", int(v)) for v in flat)
        if depth == 8:
            return bytes(int(v) for v in flat)
        bits = "".join(format(int(v), "0%db" % depth) for v in flat)
        bits += "0" * (-len(bits) % 8)
        return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))

    bpp = max(1, ch * depth // 8)

    def filter_rows(rows):
        out, prev = b"", None
        for y, cur in enumerate(rows):
            t = filters[y % len(filters)]
            f = bytearray(len(cur))
            for i in range(len(cur)):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i] if prev is not None else 0
                c = prev[i - bpp] if (prev is not None and i >= bpp) else 0
                if t == 0:
                    pr = 0
                elif t == 1:
                    pr = a
                elif t == 2:
                    pr = b
                elif t == 3:
                    pr = (a + b) >> 1
                else:
                    pp = a + b - c
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                f[i] = (cur[i] - pr) & 255
            out += bytes([t]) + bytes(f)
            prev = cur
        return out

    if interlace:
        raw = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = img[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filter_rows([pack_row(r) for r in sub])
    else:
        raw = filter_rows([pack_row(r) for r in img])

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    z = zlib.compress(raw, 6)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette.reshape(-1).tolist()))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    half = len(z) // 2  # two IDAT chunks: the stream may be split anywhere
    return out + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")


PNG_CASES = [  # colour type, bit depth, interlace
    (2, 8, False), (6, 8, False), (0, 8, False), (4, 8, False), (3, 8, False), (2, 16, False), (6, 16, True),
    (0, 1, False), (0, 2, True), (0, 4, False), (3, 2, False), (3, 4, True), (2, 8, True), (0, 16, False), (4, 16, True),
]


@pytest.mark.parametrize("color,depth,interlace", PNG_CASES)
def test_env_map_loader_decodes_png(tmp_path, color, depth, interlace):
    """PNG environment maps (the reference loads any stb_image format with 4 requested channels,
    app/hdre_loader.cpp:13): every colour type / bit depth, all five filters, Adam7, PLTE + tRNS, two IDAT
    chunks; expected RGBA8 stated in numpy from the PNG specification (lossless format)."""
    L = _host()
    L.clvr_host_hdr_probe.restype = C.c_longlong
    L.clvr_host_hdr_probe.argtypes = [C.c_char_p, C.POINTER(C.c_uint), C.c_void_p, C.c_longlong]
    rng = np.random.default_rng(color * 100 + depth + interlace)
    w, h = 37, 19
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    palette = trns = None
    if color == 3:
        n_pal = 1 << depth
        palette = rng.integers(0, 256, (n_pal, 3)).astype(np.uint8)
        trns = rng.integers(0, 256, n_pal // 2).astype(np.uint8).tolist()  # shorter than the palette: the rest is opaque
        img = rng.integers(0, n_pal, (h, w, 1))
    else:
        img = rng.integers(0, 1 << depth, (h, w, ch))
        img[:, :5] = img[:1, :1]  # flat areas for the filters
    if color == 0 and depth <= 8:
        key = int(img[3, 7, 0])
        trns = [0, key]  # one fully transparent grey level
    if color == 2 and depth == 8:
        key3 = img[2, 9].tolist()
        trns = [0, key3[0], 0, key3[1], 0, key3[2]]
    data = _png_bytes(img, color, depth, interlace, palette, trns)
    path = str(tmp_path / "env.png")
    open(path, "wb").write(data)

    def to8(v):
        if depth == 16:
            return v >> 8
        if depth == 8:
            return v
        return v * {1: 255, 2: 85, 4: 17}[depth]

    want = np.zeros((h, w, 4), np.uint8)
    want[..., 3] = 255
    if color == 3:
        want[..., :3] = palette[img[..., 0]]
        a = np.full(1 << depth, 255)
        a[: len(trns)] = trns
        want[..., 3] = a[img[..., 0]]
    elif color == 0:
        want[..., :3] = to8(img[..., :1])
        if trns is not None:
            want[..., 3] = np.where(img[..., 0] == key, 0, 255)
    elif color == 4:
        want[..., :3] = to8(img[..., :1])
        want[..., 3] = to8(img[..., 1])
    elif color == 2:
        want[..., :3] = to8(img)
        if trns is not None:
            want[..., 3] = np.where((img == np.array(key3)).all(-1), 0, 255)
    else:
        want[...] = to8(img)
    dims = (C.c_uint * 2)()
    out = np.zeros((h, w, 4), np.uint8)
    n = L.clvr_host_hdr_probe(path.encode(), dims, out.ctypes.data, out.nbytes)
    assert n == out.nbytes and (dims[0], dims[1]) == (w, h)
    assert np.array_equal(out, want)


def test_env_map_golden_fixtures():
    """tests/golden/hostio: files + the RGBA8 the reference's own loader gave for them (made by
    tests/golden/make_hostio_golden.py from oracle/_ref); usable where the reference tree does not exist"""
    L = _host()
    L.clvr_host_hdr_probe.restype = C.c_longlong
    L.clvr_host_hdr_probe.argtypes = [C.c_char_p, C.POINTER(C.c_uint), C.c_void_p, C.c_longlong]
    d = os.path.join(ROOT, "tests", "golden", "hostio")
    names = [n for n in sorted(os.listdir(d)) if not n.endswith(".npy")]
    assert len(names) >= 4
    for name in names:
        want = np.load(os.path.join(d, name + ".rgba.npy"))
        out = np.zeros_like(want)
        dims = (C.c_uint * 2)()
        assert L.clvr_host_hdr_probe(os.path.join(d, name).encode(), dims, out.ctypes.data, out.nbytes) == out.nbytes
        assert (dims[1], dims[0]) == want.shape[:2]
        assert np.array_equal(out, want), name


@pytest.mark.parametrize("rle,w", [(True, 64), (False, 64), (True, 5), (True, 300)])
def test_hdre_loader_decodes_radiance_files(tmp_path, rle, w):
    """app/hdre_loader.cpp mirror (CPU only): RGBE decode + gamma-2.2 LDR conversion, checked against the numpy
    statement of the same formula (tests/test_ref_hostio.py checks it against the reference's own loader)."""
    L = _host()
    L.clvr_host_hdr_probe.restype = C.c_longlong
    L.clvr_host_hdr_probe.argtypes = [C.c_char_p, C.POINTER(C.c_uint), C.c_void_p, C.c_longlong]
    rng = np.random.default_rng(w)
    h = 17
    rgb = (rng.random((h, w, 3), dtype=np.float32) ** 3 * 4.0).astype(np.float32)
    rgb[:, : w // 3] = rgb[:, :1]      # long runs for the RLE path
    rgb[0, 0] = 0.0
    rgbe = scene.float_to_rgbe(rgb)
    path = str(tmp_path / "env.hdr")
    scene.write_hdr(path, rgbe, rle=rle)
    dims = (C.c_uint * 2)()
    out = np.zeros((h, w, 4), np.uint8)
    n = L.clvr_host_hdr_probe(path.encode(), dims, out.ctypes.data, out.nbytes)
    assert n == out.nbytes and list(dims) == [w, h]
    assert np.array_equal(out, scene.rgbe_to_ldr(rgbe))


def test_tf_source_generator_matches_the_reference_format():
    """tf_rect_selection::create_cl_condition + ui::flush_tf (app/tf_part.cpp:55-79, app/ui.cpp:160-168):
    the C++ generator and the Python statement of the same format produce the same text, and the product's
    parser accepts it."""
    from cl_volume_renderer_amd import ffi

    L = _host()
    L.clvr_host_tf_source.restype = C.c_longlong
    L.clvr_host_tf_source.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_char_p, C.c_longlong]
    rects = [(500.0, 1200.0, 0.0, 4000.0, (1.0, 1.0, 1.0, 1.0)),
             (812.5, 900.25, 100.0, 4000.0, (0.5, 0.25, 1.0, 0.0)),
             (-100.0, 1e-3, 10.5, 20.5, (0.1, 0.2, 0.3, 0.4))]
    stats = (-2000.0, 3000.0, 0.0, 4000.0)
    flat = []
    for r in rects:
        flat += [r[0], r[1], r[2], r[3], *r[4]]
    buf = C.create_string_buffer(4096)
    n = L.clvr_host_tf_source((C.c_float * len(flat))(*flat), len(rects), (C.c_float * 4)(*stats), buf, 4096)
    code = buf.value.decode()
    assert n == len(code)
    assert code == scene.tf_rect_source(rects, stats)
    assert code.startswith("inline bool is_event_gen(short value, short gradient, int4 *color){\n  if(value >= 500 && value <= 1200)\n {")
    assert ffi.parse_tf(code).n == 3


@pytest.mark.gpu
def test_headless_application_renders_the_oracles_frame(orc, tmp_path):
    """main.cpp + the frame_emitter part of ui::run without the window: NRRD + .hdr from disk, default TF,
    N frames with std::rand() seeds; the frame's checksum equals the oracle's resolved frame."""
    import json
    import subprocess

    n = 192  # large enough that the start-up view stays below the 256-token cap (order-independent cache)
    vol = scene.phantom(n)
    scene.write_nrrd(str(tmp_path / "v.nrrd"), vol, use_gzip=True)
    rng = np.random.default_rng(5)
    rgbe = scene.float_to_rgbe((rng.random((64, 128, 3), dtype=np.float32) * 1.5).astype(np.float32))
    scene.write_hdr(str(tmp_path / "e.hdr"), rgbe)
    env = scene.rgbe_to_ldr(rgbe)
    W, H, frames = 1024, 512, 2
    exe = os.path.join(ROOT, "cl_volume_renderer_amd", "clvr_headless")
    out = subprocess.run([exe, str(tmp_path / "v.nrrd"), str(tmp_path / "e.hdr"), str(frames), str(W), str(H)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip().splitlines()[-1])

    tf = scene.tf_default_source()
    sdf, _, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (2048, 1024), (W, H))
    s = n / 512.0
    pos = np.array([-200 * s, 200 * s, -200 * s], np.float32)
    d = scene.camera_direction(0.9, 6.183)
    for seed in scene.glibc_rand(frames):
        o.render(pos, d, seed)
    assert 0 < o.cache.reshape(-1, 4)[:, 3].max() < 256 and (o.hit_index >= 0).sum() > 10000
    o.resolve(pos, d)
    h = 1469598103934665603
    for b in o.frame[:H, :W].tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert got["frame_fnv1a"] == "%016x" % h


def test_fail_hard_error_convention_without_a_device():
    """clw_fail_hard_on_error (reference opencl_wrapper/include/clw_helper.hpp:293-309): any failing call
    prints the call site and the error name to stderr and exits with status 1.  Without a GPU the very
    first call -- clw_context's constructor -- fails, which is exactly the convention to observe."""
    import subprocess

    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = os.path.join(ROOT, "cl_volume_renderer_amd", "sdf_test")
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "sdf_testdata.nrrd"),
                          os.path.join(ROOT, "tests", "golden", "sdf_values.x")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1
    assert "CLWH_ERR_NO_DEVICE" in out.stderr and "Exiting application" in out.stderr
    assert "File      :" in out.stderr and "Line      :" in out.stderr


@pytest.mark.gpu
def test_render_frame_device_hands_the_frame_over_without_a_readback(orc):
    """SURVEY 8f rank 4 (clw_foreign_memory.hpp:10-51, ui.cpp:296-308): renderer::render_frame_device writes the
    frame into device memory the DISPLAY owns (here a torch tensor stands in for the mapped GL buffer), ordered to
    the display's stream by an event; the host never pulls the frame.  One pass, then four passes batched into one
    launch: the cache and the frame equal the oracle given the same std::rand() seeds."""
    import torch

    L = _host()
    L.clvr_host_render_frame_device.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int,
                                                C.c_int, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_int]
    n = 48
    vol = scene.phantom(n)
    env = scene.env_map(256, 128)
    tf = scene.tf_default_source()
    pos = np.array([-110.0, 150.0, -110.0], np.float32)
    look = np.array([0.1, 6.6], np.float32)
    W, H = 512, 256
    h = L.clvr_host_create()
    L.clvr_host_load(h, vol.ctypes.data, n, n, n, env.ctypes.data, env.shape[1], env.shape[0])
    L.clvr_host_flush(h, tf.encode())

    display = torch.cuda.Stream()
    shown = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")    # the display's buffer
    torch.cuda.synchronize()
    # the seed state of a fresh process, set AFTER everything that may initialise a piece of the ROCm runtime (the first side stream
    # of a process draws from rand(): the test passed inside the whole suite and failed when run alone)
    C.CDLL("libc.so.6").srand(1)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    want_sdf, _, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    o = orc.Scene(vol, want_sdf, env, orc.parse_tf(tf), (W, H))
    d = scene.camera_direction(look[0], look[1])
    seeds = scene.glibc_rand(5)

    L.clvr_host_render_frame_device(h, fp(pos), fp(look), W, H, 1, C.c_void_p(shown.data_ptr()), W, H,
                                    C.c_void_p(display.cuda_stream), 1)
    with torch.cuda.stream(display):
        first = shown.clone()                 # "the blit": runs on the display's stream, after the frame is complete
    o.render(pos, d, seeds[0])
    o.resolve(pos, d)
    display.synchronize()
    assert np.array_equal(first.cpu().numpy(), o.frame)

    L.clvr_host_render_frame_device(h, fp(pos), fp(look), W, H, 0, C.c_void_p(shown.data_ptr()), W, H,
                                    C.c_void_p(display.cuda_stream), 4)
    with torch.cuda.stream(display):
        second = shown.clone()
    for s in seeds[1:]:
        o.render(pos, d, s)
    o.resolve(pos, d)
    display.synchronize()
    cache = np.empty(L.clvr_host_cache_len(h), np.uint16)
    L.clvr_host_pull_cache(h, cache.ctypes.data)
    assert 0 < cache.reshape(-1, 4)[:, 3].max() < 256
    assert np.array_equal(cache, o.cache)
    assert np.array_equal(second.cpu().numpy(), o.frame)
    L.clvr_host_destroy(h)
