"""The certified fast path of the environment-map lookup (csrc/env_fast.hpp), measured on the CPU:
its approximations use IEEE +,*,/,sqrt,fma only, so this host build computes what gfx950 computes.

1. the approximations' worst absolute error against binary64 libm stays well inside the bracket;
2. whenever the fast path claims a texel, it is the texel of the exact contract (oracle)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("envfast") / "env_probe.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "csrc", "env_fast_probe.cpp"), "-o", so])
    L = C.CDLL(so)
    L.probe_bracket.restype = C.c_float
    L.probe_atan2_approx.restype = C.c_float
    L.probe_atan2_approx.argtypes = [C.c_float, C.c_float]
    L.probe_sweep.argtypes = [C.c_long, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.probe_env_texel_fast.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int)]
    return L


def test_approximation_error_is_well_inside_the_bracket(probe):
    ea, es = C.c_double(), C.c_double()
    probe.probe_sweep(4_000_000, C.byref(ea), C.byref(es))
    bracket = probe.probe_bracket()
    half_ulp_pi = 2.0 ** -23  # RN(angle) may sit half an ulp away from the true angle
    assert ea.value < 4e-7 and es.value < 4e-7
    assert 4 * max(ea.value, es.value) + half_ulp_pi < bracket


def test_sign_conventions_of_atan2(probe):
    f = probe.probe_atan2_approx
    assert abs(f(0.0, -1.0) - np.pi) < 1e-6
    assert abs(f(-0.0, -1.0) + np.pi) < 1e-6       # atan2(-0, x<0) = -pi: the texel column flips side
    assert abs(f(1.0, -0.0) - np.pi / 2) < 1e-6
    assert np.isnan(f(0.0, 0.0))                   # undecidable -> the caller goes exact


@pytest.mark.parametrize("w,h", [(4096, 2048), (64, 32), (1000, 333)])
def test_certified_texels_equal_the_exact_contract(probe, orc, w, h):
    rng = np.random.default_rng(w)
    n = 150_000
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    # adversarial: directions sitting on texel-column boundaries and at the poles / seam
    k = rng.integers(0, w, size=5000)
    ang = ((k / w) - 0.5) * 2 * np.pi
    edge = np.stack([np.sin(ang), rng.uniform(-0.9, 0.9, size=ang.size), np.cos(ang)], axis=1).astype(np.float32)
    special = np.array([[0, 1, 0], [0, -1, 0], [0, 0, -1], [-1e-9, 0, -1], [1e-9, 0, -1], [-0.0, 0.3, -1.0],
                        [0, 1.0000001, 0], [0, 0, 0], [np.nan, 0, 1]], np.float32)
    dirs = np.concatenate([d, edge, special])
    ij = (C.c_int * 2)()
    want = (C.c_int32 * 2)()
    undecided = 0
    for v in dirs:
        ok = probe.probe_env_texel_fast(float(v[0]), float(v[1]), float(v[2]), w, h, ij)
        if not ok:
            undecided += 1
            continue
        orc.lib().orc_env_texel((C.c_float * 3)(*[float(x) for x in v]), w, h, want)
        assert (ij[0], ij[1]) == (want[0], want[1]), (v, list(ij), list(want))
    # the fast path must decide almost everything at random directions (edge cases aside)
    assert undecided < 0.02 * n + edge.shape[0] + special.shape[0]
