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


def _bilateral(gpu_ctx, vol):
    Z, Y, X = vol.shape
    v = gpu_ctx.image_from(vol)
    dst = gpu_ctx.image([X, Y, Z], 1, np.int16, (Z, Y, X))
    ev = lambda g: (g + 7) // 8 * 8  # noqa: E731  get_volume_size_evenness(8), app/reference_volume.cpp:76
    k = gpu_ctx.kernel("volume_filter.cl", "bilateral_filter")
    k.launch([ev(X), ev(Y), ev(Z)], [4, 4, 4], v, dst)
    got = dst.pull()
    for m in (v, dst):
        m.release()
    k.release()
    return got


@pytest.mark.parametrize("dims", [(32, 32, 32), (37, 21, 13), (9, 5, 3), (2, 2, 2), (2, 1, 1), (70, 8, 8)])
def test_bilateral_filter_noise(gpu_ctx, orc, dims):
    """volume_filter.cl:5-11 on noise with small differences, so that most of the 125 range weights are non-zero
    (incl. denormal ones), ragged sizes, and the zero border"""
    rng = np.random.default_rng(sum(dims))
    X, Y, Z = dims
    vol = (rng.integers(-6, 7, (Z, Y, X)) + rng.integers(-3, 4, (Z, 1, 1)) * 5).astype(np.int16)
    assert np.array_equal(_bilateral(gpu_ctx, vol), orc.bilateral_filter(vol))


def test_bilateral_filter_phantom_and_extremes(gpu_ctx, orc):
    vol = scene.phantom(72, dims=(72, 64, 56))
    got = _bilateral(gpu_ctx, vol)
    assert np.array_equal(got, orc.bilateral_filter(vol))
    assert not np.array_equal(got, vol)
    ext = np.full((12, 12, 12), 32767, np.int16)
    ext[::2] = -32768
    ext[5, 5, 5] = 17
    assert np.array_equal(_bilateral(gpu_ctx, ext), orc.bilateral_filter(ext))


def test_bilateral_filter_rejects_in_place_and_mismatched_images(gpu_ctx):
    from cl_volume_renderer_amd import ffi

    a = gpu_ctx.image([8, 8, 8], 1, np.int16, (8, 8, 8))
    b = gpu_ctx.image([8, 8, 4], 1, np.int16, (4, 8, 8))
    k = gpu_ctx.kernel("volume_filter.cl", "bilateral_filter")
    for src, dst in ((a, a), (a, b)):
        with pytest.raises(ffi.ClwhError):
            k.launch([8, 8, 8], [4, 4, 4], src, dst)
    for m in (a, b):
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


@pytest.mark.parametrize("start,length,global_size", [
    ((5, 7, 3), (24, 20, 16), None),      # eight voxels per lane, source runs at odd voxels
    ((30, 25, 20), (24, 20, 20), None),   # runs that cross the image's far faces: border 0 (x voxel by voxel, rows and slices whole)
    ((0, 0, 0), (48, 40, 36), None),      # the whole volume
    ((3, 1, 2), (20, 12, 8), None),       # rows that are no multiple of eight: the one-voxel kernel
    ((4, 4, 4), (20, 12, 8), (24, 16, 8)),  # a clip length shorter than the destination image: the voxels beyond it keep their value
])
def test_apply_clip_shapes(gpu_ctx, start, length, global_size):
    vol = scene.phantom(48, dims=(48, 40, 36))
    v = gpu_ctx.image_from(vol)
    dims = global_size or length
    before = np.full((dims[2], dims[1], dims[0]), 77, np.int16)
    dst = gpu_ctx.image_from(before)
    b_start = gpu_ctx.buffer_from(np.array(start, np.uint32))
    b_len = gpu_ctx.buffer_from(np.array(list(length) + [4], np.uint32))
    k = gpu_ctx.kernel("reference_volume_clip.cl", "apply_clip")
    k.launch(list(dims), [4, 4, 4], v, dst, b_start, b_len)
    want = before.copy()
    want[: length[2], : length[1], : length[0]] = orc_volume.apply_clip(vol, start, length)
    assert np.array_equal(dst.pull(), want)
    for m in (v, dst, b_start, b_len):
        m.release()
    k.release()


@pytest.mark.parametrize("dims", [(192, 160, 144), (200, 170, 150), (2056, 6, 5)])
def test_stats_and_histogram_at_scale(gpu_ctx, dims):
    """fetch_stats and tf_sort_values are persistent grids (a lane takes eight voxels, 16-byte loads when the row length allows; the
    histogram counts into a per-block hash table in LDS): volumes with more work items than blocks, rows longer than a block's chunk,
    both load paths -- every statistic and every one of the 500 x 500 bins against the numpy restatement"""
    vol = scene.phantom(max(dims[:3]) if max(dims) < 1000 else 64, dims=dims)
    v = gpu_ctx.image_from(vol)
    ev = lambda g, l=8: (g + l - 1) // l * l  # noqa: E731
    G = [ev(dims[0]), ev(dims[1]), ev(dims[2])]
    stats = gpu_ctx.buffer_from(INIT)
    k = gpu_ctx.kernel("reference_volume_figures.cl", "fetch_stats")
    k.launch(G, [4, 4, 4], v, stats)
    st = stats.pull()
    want = orc_volume.fetch_stats(vol)
    assert np.array_equal(st[:4], want[:4]), (st, want)
    W = H = 500
    bins = gpu_ctx.buffer_from(np.zeros(W * H, np.uint32))
    kh = gpu_ctx.kernel("histogram.cl", "tf_sort_values")
    kh.launch(G, [4, 4, 4], v, bins, np.uint32(W), np.uint32(H), float(st[0]), float(st[1]), float(st[2]), float(st[3]))
    got = bins.pull()
    want_bins = orc_volume.tf_sort_values(vol, W, H, float(st[0]), float(st[1]), float(st[2]), float(st[3]))
    assert np.array_equal(got.reshape(-1), np.asarray(want_bins).reshape(-1).astype(np.uint32))
    assert int(got.sum()) > 0.9 * vol.size   # only the voxels on the top row / column of the frame fall outside
    for m in (v, stats, bins):
        m.release()
    k.release(); kh.release()
