"""-m gpu: the volume pre-processing kernels next to the hot path through the
generic launch, against their numpy restatement."""
import numpy as np
import pytest

from cl_volume_renderer_amd import scene
from oracle import orc_volume

pytestmark = pytest.mark.gpu

INIT = np.array([2**31 - 1, -2**31, 2**31 - 1, -2**31, -2**31], dtype=np.int32)  # app/reference_volume.cpp:23-28


@pytest.mark.parametrize("dims", [(64, 64, 64), (70, 33, 45), (130, 20, 9), (5, 4, 3)])
def test_fetch_stats(gpu_ctx, dims):
    vol = scene.phantom(max(dims), dims=dims)
    v = gpu_ctx.image_from(vol)
    stats = gpu_ctx.buffer_from(INIT)
    ev = lambda g, l: (g + l - 1) // l * l  # noqa: E731
    k = gpu_ctx.kernel("reference_volume_figures.cl", "fetch_stats")
    k.launch([ev(dims[0], 8), ev(dims[1], 8), ev(dims[2], 8)], [4, 4, 4], v, stats)
    got = stats.pull()
    want = orc_volume.fetch_stats(vol)
    assert np.array_equal(got[:4], want[:4]), (got, want)
    assert got[4] == INIT[4]  # MAX_COUNT_VALUE is never written by the kernel
    for m in (v, stats):
        m.release()
    k.release()


def test_apply_clip(gpu_ctx):
    vol = scene.phantom(48, dims=(48, 40, 36))
    v = gpu_ctx.image_from(vol)
    start, length = (5, 7, 3), (24, 20, 16)
    dst = gpu_ctx.image(list(length), 1, np.int16, (length[2], length[1], length[0]))
    b_start = gpu_ctx.buffer_from(np.array(start, np.uint32))
    b_len = gpu_ctx.buffer_from(np.array(list(length) + [4], np.uint32))
    k = gpu_ctx.kernel("reference_volume_clip.cl", "apply_clip")
    k.launch(list(length), [4, 4, 4], v, dst, b_start, b_len)
    assert np.array_equal(dst.pull(), orc_volume.apply_clip(vol, start, length))
    for m in (v, dst, b_start, b_len):
        m.release()
    k.release()
