"""-m "not gpu": mutation fuzzing of the parsers that take caller-controlled bytes (PNG environment maps, the
transfer-function source) under AddressSanitizer + UBSan on the CPU build (tests/csrc/fuzz_parsers.cpp)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from cl_volume_renderer_amd import scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fuzz") / "fuzz_parsers")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "csrc", "fuzz_parsers.cpp"),
                           os.path.join(ROOT, "cl_volume_renderer_amd", "app", "png_reader.cpp"),
                           os.path.join(ROOT, "cl_volume_renderer_amd", "app", "jpeg_reader.cpp"),
                           os.path.join(ROOT, "cl_volume_renderer_amd", "app", "hdre_loader.cpp"),
                           os.path.join(ROOT, "cl_volume_renderer_amd", "app", "nrrd_loader.cpp"),
                           os.path.join(ROOT, "cl_volume_renderer_amd", "csrc", "tf_parse.cpp"),
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cl_volume_renderer_amd", "app"),
                           "-lz", "-o", exe])
    return exe


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))


def _png(w, h, color, depth, interlace, rng):
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    bits = ch * depth
    raw = b""
    if interlace:
        subs = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    else:
        subs = [(0, 0, 1, 1)]
    for x0, y0, dx, dy in subs:
        pw = (w - x0 + dx - 1) // dx if w > x0 else 0
        ph = (h - y0 + dy - 1) // dy if h > y0 else 0
        for _ in range(ph if pw else 0):
            raw += bytes([int(rng.integers(0, 5))]) + rng.integers(0, 256, (pw * bits + 7) // 8).astype(np.uint8).tobytes()
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, int(interlace)))
    if color == 3:
        out += _chunk(b"PLTE", rng.integers(0, 256, 3 << depth).astype(np.uint8).tobytes())
        out += _chunk(b"tRNS", rng.integers(0, 256, 3).astype(np.uint8).tobytes())
    return out + _chunk(b"IDAT", zlib.compress(raw)) + _chunk(b"IEND", b"")


def test_png_reader_survives_mutated_files(fuzzer, tmp_path):
    rng = np.random.default_rng(1)
    paths = []
    for k, (color, depth, il) in enumerate([(2, 8, False), (6, 16, True), (3, 4, False), (0, 1, True), (4, 8, False), (3, 8, True)]):
        p = str(tmp_path / ("s%d.png" % k))
        open(p, "wb").write(_png(13 + k, 9 + 2 * k, color, depth, il, rng))
        paths.append(p)
    out = subprocess.run([fuzzer, "60000", "png"] + paths, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    accepted = int(out.stdout.split(",")[1].split()[0])
    assert accepted >= len(paths)  # the unmodified seeds decode


def test_jpeg_reader_survives_mutated_files(fuzzer, tmp_path):
    from tests.jpeg_writer import Encoder, quant_table

    rng = np.random.default_rng(5)
    paths = []
    for k, (samp, prog, rst) in enumerate([([(1, 1)], False, 0), ([(2, 2), (1, 1), (1, 1)], False, 3), ([(2, 1), (1, 1), (1, 1)], True, 0),
                                           ([(1, 1)] * 4, False, 0), ([(2, 2), (1, 1), (1, 1)], True, 2)]):
        planes = [np.clip(rng.normal(128, 50, (21, 27)), 0, 255).astype(np.uint8) for _ in samp]
        enc = Encoder(planes, samp, [quant_table(0.8), quant_table(1.5)], [0] + [1] * (len(samp) - 1), progressive=prog,
                      restart=rst, style=k % 2)
        p = str(tmp_path / ("s%d.jpg" % k))
        open(p, "wb").write(enc.encode_progressive() if prog else enc.encode_baseline())
        paths.append(p)
    out = subprocess.run([fuzzer, "40000", "jpg"] + paths, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert int(out.stdout.split(",")[1].split()[0]) >= len(paths)


def test_tf_parser_survives_mutated_sources(fuzzer, tmp_path):
    sources = [scene.tf_default_source(), scene.tf_gradient_source(),
               scene.tf_rect_source([(20.0, 60.0, 0.0, 4000.0, (0.9, 0.5, 0.2, 0.6)), (100.0, 900.0, 5.0, 50.0, (0.1, 0.2, 0.3, 1.0))]),
               "inline bool is_event_gen(short value, short gradient, int4 *color){ return (value >= -5 && gradient < 7.5f); }"]
    paths = []
    for k, src in enumerate(sources):
        p = str(tmp_path / ("tf%d.cl" % k))
        open(p, "w").write(src)
        paths.append(p)
    out = subprocess.run([fuzzer, "200000", "tf"] + paths, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert int(out.stdout.split(",")[1].split()[0]) >= len(paths)


def _loader_env():
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "exitcode=77:detect_leaks=0"  # 1 is the loaders' own fail-hard status
    return env


def test_nrrd_loader_survives_mutated_files(fuzzer, tmp_path):
    """the fail-hard loaders run in forked children: exit 0 (loaded) or 1 (rejected) only"""
    rng = np.random.default_rng(3)
    vol = rng.integers(-1000, 3000, (6, 5, 7)).astype(np.int16)
    seeds = []
    for k, gz in enumerate((True, False)):
        p = str(tmp_path / ("v%d.nrrd" % k))
        scene.write_nrrd(p, vol, use_gzip=gz)
        seeds.append(p)
    seeds.append(os.path.join(ROOT, "tests", "golden", "sdf_testdata.nrrd"))
    out = subprocess.run([fuzzer, "3000", "nrrd", str(tmp_path / "scratch.nrrd")] + seeds, capture_output=True, text=True,
                         timeout=900, env=_loader_env())
    assert out.returncode == 0, out.stderr[-3000:]
    assert int(out.stdout.split(",")[1].split()[0]) >= len(seeds)


def test_hdr_loader_survives_mutated_files(fuzzer, tmp_path):
    rng = np.random.default_rng(4)
    seeds = []
    for k, (w, rle) in enumerate(((40, True), (40, False), (5, True))):
        rgb = (rng.random((9, w, 3), dtype=np.float32) * 2).astype(np.float32)
        rgb[:, : w // 2] = rgb[:, :1]
        p = str(tmp_path / ("e%d.hdr" % k))
        scene.write_hdr(p, scene.float_to_rgbe(rgb), rle=rle)
        seeds.append(p)
    out = subprocess.run([fuzzer, "3000", "hdr", str(tmp_path / "scratch.hdr")] + seeds, capture_output=True, text=True,
                         timeout=900, env=_loader_env())
    assert out.returncode == 0, out.stderr[-3000:]
    assert int(out.stdout.split(",")[1].split()[0]) >= len(seeds)
