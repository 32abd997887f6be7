"""-m gpu: the scheduling of the LONG k_bounce launch (lanes refill at 16 idle, the step loop ends at 16 marching lanes,
exit certificates on: csrc/render_kernels.hip launch_bounce) against the oracle.

Small scenes are short launches, where every wave marches its samples to completion and certificates are off, so the
other parity tests do not reach that code.  `gpu_ctx_long` (conftest.py) is a context created under
CLWH_TUNE_LONG_LAUNCH=1: every launch is scheduled like the 64-pass headline launch.  Exit certificates replace the tail of
a march by its outcome, so the parity targets are the same and still bit-exact; the scenes are chosen so that certificates
are granted (an object in a mostly empty volume), refused (rays that graze it) and switched off (a table that contains the
border value 0), and so that the rare voxel-less positions of the lean step loop (far face, NaN, -0.0) occur.  (A CLVR_BOUNCE_STATS build of the library
prints per-launch counts: these scenes grant between 0.3 and 2 certificates per item.)"""
import numpy as np
import pytest

from cl_volume_renderer_amd import ffi, scene
from tests.gpu_util import GpuScene, look_at_centre, small_scene
from tests.test_gpu_edge_cases import _parity
from tests.test_gpu_render import _compare_passes

pytestmark = pytest.mark.gpu


def _ball_in_empty_space(n, radius, centre=None):
    """an object the rays leave for good: macro cells (16^3 voxels) around it are free, certificates are granted"""
    z, y, x = np.mgrid[0:n, 0:n, 0:n].astype(np.float32)
    c = np.array(centre if centre is not None else [n / 2, n / 2, n / 2], np.float32)
    r = np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2)
    vol = np.where(r < radius, 900.0 - 6.0 * r, -900.0 + 3.0 * np.sin(0.3 * x) * np.cos(0.2 * y + 0.1 * z))
    return np.ascontiguousarray(vol.astype(np.int16))


def test_phantom_parity_under_long_launch_scheduling(gpu_ctx_long, orc):
    vol, sdf, env, tf = small_scene(orc, 128, env_wh=(1024, 512))
    pos, d = scene.default_camera(128)
    st = _compare_passes(orc, gpu_ctx_long, vol, sdf, env, tf, (512, 288), pos, d, scene.glibc_rand(5))
    assert st["hits"] > 10000


@pytest.mark.parametrize("mode", ["voxel", "image"])
def test_ball_in_empty_space_certificates_granted(gpu_ctx_long, orc, mode):
    vol = _ball_in_empty_space(160, 30.0, centre=[70, 90, 80])
    env = scene.env_map(512, 256)
    pos, d = look_at_centre(vol, [-40, 200, -60])
    hits = _parity(orc, gpu_ctx_long, vol, env, scene.tf_default_source(), (320, 200), pos, d, scene.glibc_rand(4), mode=mode)
    assert hits > 2000


def test_macro_cells_of_64_voxels(gpu_ctx_long_big_cells, orc):
    """large volumes get larger cells (clwh_internal.hpp macro_cell_shift); forced here on a 200 x 170 x 150 volume"""
    vol = np.full((150, 170, 200), -800, np.int16)
    z, y, x = np.mgrid[0:150, 0:170, 0:200].astype(np.float32)
    r = np.sqrt((x - 150) ** 2 + (y - 120) ** 2 + (z - 100) ** 2)
    vol[r < 28] = 900
    env = scene.env_map(256, 128)
    pos, d = look_at_centre(vol, [-40, 230, -50])
    hits = _parity(orc, gpu_ctx_long_big_cells, vol, env, scene.tf_default_source(), (256, 160), pos, d, scene.glibc_rand(4))
    assert hits > 500


def test_non_cubic_volume_with_ragged_macro_cells(gpu_ctx_long, orc):
    """dimensions that are no multiples of the brick (8) or the macro cell (16): 150 x 70 x 41"""
    rng = np.random.default_rng(9)
    vol = np.full((41, 70, 150), -800, np.int16)
    vol[10:30, 20:50, 40:100] = rng.integers(600, 1100, size=(20, 30, 60), dtype=np.int16)
    vol[35:41, 60:70, 140:150] = 1000   # a second object in the far corner: boxes towards that corner are not free
    env = scene.env_map(256, 128)
    pos, d = look_at_centre(vol, [-30, 120, -40])
    hits = _parity(orc, gpu_ctx_long, vol, env, scene.tf_default_source(), (256, 160), pos, d, scene.glibc_rand(4))
    assert hits > 500


def test_gradient_tf_under_long_launch_scheduling(gpu_ctx_long, orc):
    vol = _ball_in_empty_space(96, 22.0)
    env = scene.env_map(256, 128)
    pos, d = look_at_centre(vol, [-30, 120, -35])
    hits = _parity(orc, gpu_ctx_long, vol, env, scene.tf_gradient_source(), (192, 128), pos, d, scene.glibc_rand(3))
    assert hits > 500


def test_nan_rays_and_border_hits_under_long_launch_scheduling(gpu_ctx_long, orc):
    """piecewise-constant volume: zero gradients -> NaN normals -> NaN rays, whose positions have no voxel ever (EV_CHECK finishes
    their marches); then the same with a table that contains the border value 0 (certificates off, border Hits)"""
    rng = np.random.default_rng(77)
    coarse = rng.choice(np.array([-1000, 700, 900, 1100, 40], np.int16), size=(8, 8, 8))
    vol = np.ascontiguousarray(np.kron(coarse, np.ones((6, 6, 6), np.int16)).astype(np.int16))
    env = scene.env_map(64, 32)
    pos, d = look_at_centre(vol, [-15, 30, -20])
    hits = _parity(orc, gpu_ctx_long, vol, env, scene.tf_default_source(), (128, 96), pos, d, scene.glibc_rand(3))
    assert hits > 3000
    rng = np.random.default_rng(123)
    coarse = rng.choice(np.array([-1000, 700, 150, 1100, 40, -50], np.int16), size=(8, 8, 8))
    vol = np.ascontiguousarray(np.kron(coarse, np.ones((6, 6, 6), np.int16)).astype(np.int16))
    tf = scene.tf_rect_source([(-100.0, 300.0, 0.0, 4000.0, (0.9, 0.6, 0.3, 0.7))])
    hits = _parity(orc, gpu_ctx_long, vol, env, tf, (128, 96), pos, d, scene.glibc_rand(3))
    assert hits > 3000


def test_positions_exactly_on_the_faces(gpu_ctx_long, orc):
    """a camera on an integer lattice looking along an axis of a small volume: steps of integer length land exactly on the far
    face (coordinate == dimension: still inside for exited_volume, border texel) and on coordinate 0"""
    vol = np.full((32, 32, 32), -700, np.int16)
    vol[12:20, 12:20, 12:20] = 1000
    env = scene.env_map(64, 32)
    pos = np.array([16.0, 16.0, -24.0], np.float32)
    d = np.array([0.0, 0.0, 1.0], np.float32)
    for tf in (scene.tf_default_source(), scene.tf_rect_source([(-100.0, 300.0, 0.0, 4000.0, (0.9, 0.6, 0.3, 0.7))])):
        _parity(orc, gpu_ctx_long, vol, env, tf, (64, 64), pos, d, scene.glibc_rand(3))


def test_fused_long_launch_equals_the_passes_one_by_one(gpu_ctx, gpu_ctx_long, orc):
    """the same 12 passes as one launch of the long-launch context and as 12 short launches of the ordinary one"""
    vol = _ball_in_empty_space(128, 26.0, centre=[60, 70, 58])
    env = scene.env_map(512, 256)
    tf = scene.tf_default_source()
    sdf, _, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    pos, d = look_at_centre(vol, [-40, 170, -50])
    seeds = scene.glibc_rand(12)
    acc = []
    for ctx, fused in ((gpu_ctx, False), (gpu_ctx_long, True)):
        g = GpuScene(ctx, vol, sdf, env, tf, (384, 240))
        ctx.buffer_reset(g.accum[0])
        if fused:
            g.render(pos, d, None, mode=ffi.ACCUM_IMAGE_SPACE, seeds=seeds, debug=False, write_frame=False)
        else:
            for s in seeds:
                g.render(pos, d, s, mode=ffi.ACCUM_IMAGE_SPACE, debug=False, write_frame=False)
        acc.append(g.accum[0].pull(np.float32).reshape(-1, 4).copy())
        g.release()
    assert acc[0][:, 3].max() == 12.0
    assert np.array_equal(acc[0], acc[1])


@pytest.mark.parametrize("case", range(8))
def test_random_scenes_under_long_launch_scheduling(gpu_ctx_long, gpu_ctx_long_big_cells, orc, case):
    """random blobs in a mostly empty, non-cubic volume, random rectangle tables (with and without `gradient` clauses, some
    containing the border value 0), random cameras inside and outside: certificates granted, refused and switched off"""
    rng = np.random.default_rng(4000 + case)
    X, Y, Z = (int(rng.integers(40, 150)) for _ in range(3))
    vol = np.full((Z, Y, X), -900, np.int16)
    z, y, x = np.mgrid[0:Z, 0:Y, 0:X].astype(np.float32)
    for _ in range(int(rng.integers(1, 5))):
        c = rng.random(3) * np.array([X, Y, Z])
        r = float(rng.integers(4, max(6, min(X, Y, Z) // 3)))
        d2 = (x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2
        vol[d2 < r * r] = np.int16(rng.integers(200, 1300))
    vol += rng.integers(-40, 40, size=vol.shape, dtype=np.int16)
    rects = []
    for _ in range(int(rng.integers(1, 3))):
        lo = float(rng.integers(100, 900)) if rng.random() < 0.8 else float(rng.integers(-1000, -100))
        hi = lo + float(rng.integers(100, 1200))
        glo, ghi = (float(rng.integers(0, 300)), float(rng.integers(800, 3999))) if rng.random() < 0.4 else (0.0, 4000.0)
        rects.append((lo, hi, glo, ghi, tuple(float(v) for v in rng.random(4))))
    tf = scene.tf_rect_source(rects)
    env = scene.env_map(128, 64, seed=case)
    eye = [float(rng.integers(-60, X + 60)), float(rng.integers(-60, Y + 60)), float(rng.integers(-60, Z + 60))]
    pos, d = look_at_centre(vol, eye)
    ctx = gpu_ctx_long_big_cells if case % 4 == 3 else gpu_ctx_long
    _parity(orc, ctx, vol, env, tf, (160, 96), pos, d, scene.glibc_rand(3), mode="voxel" if case % 2 else "image")
