"""-m gpu: the float -> integer conversions of the contract (DESIGN.md "Semantics": truncate toward zero, saturate, NaN -> 0) are
single hardware instructions in the kernels (csrc/device_math.hpp); here the instruction is compared with the written-out
definition -- on the device -- and with the oracle's C definition on the host."""
import ctypes as C

import numpy as np
import pytest

from cl_volume_renderer_amd import ffi

pytestmark = pytest.mark.gpu


def _host_f2i(v):
    out = np.zeros(v.shape, np.int64)
    ok = ~np.isnan(v)
    big, small = ok & (v >= 2147483648.0), ok & (v <= -2147483648.0)
    mid = ok & ~big & ~small
    out[big], out[small] = 2147483647, -2147483648
    out[mid] = np.trunc(v[mid].astype(np.float64)).astype(np.int64)
    return out.astype(np.int32)


def _host_f2u(v):
    out = np.zeros(v.shape, np.uint64)
    ok = ~np.isnan(v)
    big = ok & (v >= 4294967296.0)
    mid = ok & ~big & (v > 0.0)
    out[big] = 0xFFFFFFFF
    out[mid] = np.trunc(v[mid].astype(np.float64)).astype(np.uint64)
    return out.astype(np.uint32)


def test_conversion_instructions_equal_their_definition(gpu_ctx):
    special = np.array([0.0, -0.0, 0.5, -0.5, 0.99999994, -0.99999994, 1.0, -1.0, 1.5, -1.5, 2147483520.0, 2147483648.0, -2147483648.0,
                        -2147483904.0, 4294967040.0, 4294967296.0, 1e30, -1e30, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 8388607.5,
                        -8388607.5, 16777216.0, 3e9, -3e9, 255.99998, 510.0003], np.float32)
    rng = np.random.default_rng(11)
    bits = rng.integers(0, 2 ** 32, 1 << 20, dtype=np.uint64).astype(np.uint32).view(np.float32)   # every exponent, NaN payloads, denormals
    near = (rng.integers(-70000, 70000, 1 << 18).astype(np.float32) + rng.random(1 << 18, dtype=np.float32))
    neg_nan = np.array([0xFFC00000, 0x7FC00001, 0xFF800001], np.uint32).view(np.float32)
    v = np.concatenate([special, bits, near, neg_nan]).astype(np.float32)
    n = v.size
    m_in = gpu_ctx.buffer_from(v)
    m_i = gpu_ctx.buffer(n * 8, np.int32)
    m_u = gpu_ctx.buffer(n * 8, np.uint32)
    ffi._check(ffi.lib().clwh_debug_float_conversions(gpu_ctx.h, m_in.h, C.c_uint64(n), m_i.h, m_u.h), "clwh_debug_float_conversions")
    gpu_ctx.finish()
    gi, gu = m_i.pull(), m_u.pull()
    assert np.array_equal(gi[:n], gi[n:])          # instruction == definition, on the device
    assert np.array_equal(gu[:n], gu[n:])
    with np.errstate(invalid="ignore"):
        assert np.array_equal(gi[:n], _host_f2i(v))    # == the oracle's definition, on the host
        assert np.array_equal(gu[:n], _host_f2u(v))
    for m in (m_in, m_i, m_u):
        m.release()


def test_wave_minimum_by_dpp_equals_the_shuffle_loop_and_numpy(gpu_ctx):
    """k_repack's minimum over a sub-brick's 64 voxels (the exit-certificate table's input) is taken with cross-lane DPP operations:
    the minimum in every lane position, ties, the values the kernel meets (0..127, 255) and arbitrary 32-bit patterns"""
    rng = np.random.default_rng(5)
    blocks = [rng.integers(0, 2 ** 32, (4096, 64), dtype=np.uint64).astype(np.uint32),
              rng.choice(np.array(list(range(128)) + [255], np.uint32), (4096, 64)),
              np.full((64, 64), 255, np.uint32)]
    for lane in range(64):   # a single small value at every lane position
        blocks[2][lane, lane] = lane
    ties = np.full((64, 64), 7, np.uint32)
    ties[:, ::3] = 3
    v = np.concatenate(blocks + [ties, np.zeros((1, 64), np.uint32), np.full((1, 64), 0xFFFFFFFF, np.uint32)])
    n_waves = v.shape[0]
    m_in = gpu_ctx.buffer_from(v.reshape(-1))
    m_out = gpu_ctx.buffer(n_waves * 8, np.uint32)
    ffi._check(ffi.lib().clwh_debug_wave_min(gpu_ctx.h, m_in.h, C.c_uint64(v.size), m_out.h), "clwh_debug_wave_min")
    gpu_ctx.finish()
    got = m_out.pull()
    want = v.min(axis=1)
    assert np.array_equal(got[:n_waves], want) and np.array_equal(got[n_waves:], want)
    m_in.release(); m_out.release()
