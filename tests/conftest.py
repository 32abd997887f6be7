import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build the product libraries when they are missing (hipcc cross-compiles for gfx950 without a GPU);
    on the GPU box the prebuilt files travel with the snapshot and nothing happens here."""
    from cl_volume_renderer_amd import build as product_build

    try:
        if not os.path.exists(product_build.LIB):
            product_build.build()
        if not os.path.exists(product_build.HOST_LIB):
            product_build.build_host()
    except Exception as e:  # the tests that need the libraries will fail loudly with the real reason
        print("warning: could not build the product libraries:", e, file=sys.stderr)
    # On a GPU box PyTorch (ROCm 7.0 runtime inside the wheel) and libclwhip.so (the image's ROCm 7.2 runtime) both live
    # in the test process (torch only as a stand-in for "another owner of device memory" in the hand-off tests and for
    # the 64 GiB cache of the 2048^3 test).  Whichever HIP runtime comes SECOND finds the device when torch's came first,
    # but torch reports "no ROCm-capable device" when libclwhip.so initialised the GPU before it -- so torch goes first,
    # exactly as in bench.py.  (No GPU in the build container: nothing happens there.)
    if os.path.exists("/dev/kfd"):
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception as e:
            print("warning: torch could not initialise the GPU:", e, file=sys.stderr)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import orc_ffi

    orc_ffi.build()
    orc_ffi.lib()
    return orc_ffi


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def sdf_golden(golden_dir):
    """The reference's own SDF test data: tests/sdf/testdata.nrrd + tests/sdf/values.x (copied verbatim)."""
    import numpy as np

    from cl_volume_renderer_amd import scene

    vol = scene.read_nrrd(os.path.join(golden_dir, "sdf_testdata.nrrd"))
    vals = np.array(
        [int(line.strip().rstrip(",")) for line in open(os.path.join(golden_dir, "sdf_values.x")) if line.strip()],
        dtype=np.int32,
    )
    return vol, vals


@pytest.fixture(scope="session")
def gpu_ctx():
    """A clwh context on cuda:0 -- fails loudly if the HIP library or the GPU is missing."""
    from cl_volume_renderer_amd import ffi

    ctx = ffi.Context(0)
    yield ctx
    ctx.destroy()


@pytest.fixture(scope="session")
def gpu_ctx_long():
    """A second context that schedules EVERY k_bounce launch like a long one (refill / step thresholds 16 / 16, exit
    certificates on: csrc/render_kernels.hip launch_bounce), so that small scenes exercise the code the 64-pass headline
    launch runs.  The knob is read when the context is created."""
    from cl_volume_renderer_amd import ffi

    os.environ["CLWH_TUNE_LONG_LAUNCH"] = "1"
    try:
        ctx = ffi.Context(0)
    finally:
        del os.environ["CLWH_TUNE_LONG_LAUNCH"]
    yield ctx
    ctx.destroy()


@pytest.fixture(scope="session")
def gpu_ctx_long_big_cells():
    """gpu_ctx_long with the exit certificates' macro cells forced to 64^3 voxels -- what a volume beyond 2048^3 gets by itself (2048^3: 32^3, tests/test_gpu_fullsize.py)"""
    from cl_volume_renderer_amd import ffi

    os.environ["CLWH_TUNE_LONG_LAUNCH"] = "1"
    os.environ["CLWH_TUNE_MACRO_SHIFT"] = "6"
    try:
        ctx = ffi.Context(0)
    finally:
        del os.environ["CLWH_TUNE_LONG_LAUNCH"]
        del os.environ["CLWH_TUNE_MACRO_SHIFT"]
    yield ctx
    ctx.destroy()


def _ctx_with_env(env):
    from cl_volume_renderer_amd import ffi

    os.environ.update(env)
    try:
        return ffi.Context(0)
    finally:
        for k in env:
            del os.environ[k]


@pytest.fixture(scope="session")
def gpu_ctx_two_rays():
    """every launch scheduled like a long one AND run by k_bounce2 (two rays per lane, CLWH_TUNE_BOUNCE_RAYS=2: the round-3
    lane-utilisation experiment, csrc/render_kernels.hip)"""
    ctx = _ctx_with_env({"CLWH_TUNE_LONG_LAUNCH": "1", "CLWH_TUNE_BOUNCE_RAYS": "2"})
    yield ctx
    ctx.destroy()


@pytest.fixture(scope="session")
def gpu_ctx_sdf_front():
    """a context whose clwh_sdf_build runs the byte front (one launch per layer) instead of the bit-parallel build"""
    ctx = _ctx_with_env({"CLWH_TUNE_SDF": "front"})
    yield ctx
    ctx.destroy()


@pytest.fixture(scope="session")
def gpu_ctx_sdf_waves16():
    """the bit-parallel SDF build with 16-wave blocks (regions of 64 x 48 x 48 voxels instead of 64 x 48 x 16)"""
    ctx = _ctx_with_env({"CLWH_TUNE_SDFBIT_WAVES": "16"})
    yield ctx
    ctx.destroy()


@pytest.fixture(scope="session")
def gpu_ctx_sdf_rec_lds():
    """the bit-parallel SDF build with the layer records in LDS (80 VGPRs, three blocks per CU; CLWH_TUNE_SDFBIT_REC=lds: the round-3
    occupancy experiment, csrc/sdf_kernels.hip)"""
    ctx = _ctx_with_env({"CLWH_TUNE_SDFBIT_REC": "lds"})
    yield ctx
    ctx.destroy()


@pytest.fixture(scope="session")
def gpu_ctx_half_grid():
    """k_bounce on half the persistent grid (CLWH_TUNE_BLOCKS=1024): what bench.py asks for when two frame jobs of a multi-rank run
    are in flight"""
    ctx = _ctx_with_env({"CLWH_TUNE_BLOCKS": "1024"})
    yield ctx
    ctx.destroy()
