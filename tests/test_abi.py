"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/clwh.h declares, and its host-only logic (TF parser, size helpers, error paths) behaves.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from cl_volume_renderer_amd import ffi, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "clwh.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clwh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    declared = _declared_symbols()
    assert len(declared) >= 30
    raw = ctypes.CDLL(ffi.LIB_PATH)
    missing = [s for s in declared if not hasattr(raw, s)]
    assert missing == []
    assert sorted(ffi.EXPORTED_SYMBOLS) == declared  # the Python binding covers the whole header


def test_version_and_strerror():
    L = ffi.lib()
    assert b"gfx950" in L.clwh_version()
    assert L.clwh_strerror(0) == b"CLWH_OK"
    assert L.clwh_strerror(6) == b"CLWH_ERR_TF_UNSUPPORTED"
    assert L.clwh_strerror(12345) == b"CLWH_ERR_UNKNOWN"


def test_size_helpers():
    # reference allocation X*Y*Z*4 plus the one-row overrun of utility.cl:21
    assert ffi.cache_len(38, 35, 38) == (38 * 35 * 38 + 38 * 38 + 38 + 1) * 4
    assert ffi.cache_len(2048, 2048, 2048) == (2048 ** 3 + 2048 ** 2 + 2048 + 1) * 4  # needs 64-bit
    assert ffi.accum_len(1920, 1080, 1) == 1920 * 1080
    assert ffi.accum_len(1920, 1080, 8) == 1920 * 1080 // 8
    assert ffi.accum_len(64, 64, 3) == 8 * 3 * 64  # 8 tiles per row over 3 ranks -> 3 slots per row


@pytest.mark.parametrize("source", [
    scene.TF_TEST_VALUE_GT_800,
    scene.tf_default_source(),
    scene.tf_gradient_source(),
    scene.tf_rect_source([(812.5, 900.25, 0.0, 4000.0, (0.5, 0.25, 1.0, 0.0)),
                          (-100.0, 100.0, 10.5, 20.5, (1.0, 1.0, 1.0, 1.0)),
                          (1e-3, 2.5e3, 0.0, 4000.0, (0.1, 0.2, 0.3, 0.4))]),
    "inline bool is_event_gen(short value, short gradient, int4 *color){ return (value >= -5 && gradient < 7.5f); }",
    "inline bool is_event_gen(short value, short gradient, int4 *color){\n  \n  return false;\n}\n",
])
def test_product_tf_parser_agrees_with_the_independent_one(orc, source):
    assert ffi.parse_tf(source).as_tuples() == orc.parse_tf(source).as_tuples()


@pytest.mark.parametrize("source", [
    "",
    "inline bool is_event_gen(short value, short gradient, int4 *color){ return sin(value) > 0; }",
    "inline bool is_event_gen(short value, short gradient, int4 *color){ if(value > 3 || value < 1) { int4 tmp_color = {1,2,3,4}; *color = tmp_color; return true; } return false; }",
])
def test_tf_outside_the_grammar_is_refused(source):
    with pytest.raises(ffi.ClwhError) as e:
        ffi.parse_tf(source)
    assert e.value.status == 6  # CLWH_ERR_TF_UNSUPPORTED


def test_no_silent_fallback_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ffi.ClwhError) as e:
        ffi.Context(0)
    assert e.value.status == 2  # CLWH_ERR_NO_DEVICE
