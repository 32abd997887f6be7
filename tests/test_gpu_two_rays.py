"""-m gpu: k_bounce2 -- two rays per lane (CLWH_TUNE_BOUNCE_RAYS=2), the round-3 experiment on lane utilisation -- against the oracle,
on the scenes of tests/test_gpu_long_launch.py: a ray never leaves its lane and its arithmetic is k_bounce's, so every parity target is
the same and still bit-exact; what differs is which of a lane's two rays a phase works on, and where the event-only state lives (LDS
slots addressed through a per-lane slot bit, radiance sums packed 10 : 10 : 10, o and i inside the fix-up word)."""
import numpy as np
import pytest

from cl_volume_renderer_amd import ffi, scene
from tests.gpu_util import GpuScene, look_at_centre, small_scene
from tests.test_gpu_edge_cases import _parity
from tests.test_gpu_long_launch import _ball_in_empty_space
from tests.test_gpu_render import _compare_passes

pytestmark = pytest.mark.gpu


def test_phantom_parity(gpu_ctx_two_rays, orc):
    vol, sdf, env, tf = small_scene(orc, 128, env_wh=(1024, 512))
    pos, d = scene.default_camera(128)
    st = _compare_passes(orc, gpu_ctx_two_rays, vol, sdf, env, tf, (512, 288), pos, d, scene.glibc_rand(5))
    assert st["hits"] > 10000


@pytest.mark.parametrize("mode", ["voxel", "image"])
def test_ball_in_empty_space(gpu_ctx_two_rays, orc, mode):
    vol = _ball_in_empty_space(160, 30.0, centre=[70, 90, 80])
    env = scene.env_map(512, 256)
    pos, d = look_at_centre(vol, [-40, 200, -60])
    hits = _parity(orc, gpu_ctx_two_rays, vol, env, scene.tf_default_source(), (320, 200), pos, d, scene.glibc_rand(4), mode=mode)
    assert hits > 2000


def test_gradient_tf(gpu_ctx_two_rays, orc):
    vol = _ball_in_empty_space(96, 22.0)
    env = scene.env_map(256, 128)
    pos, d = look_at_centre(vol, [-30, 120, -35])
    assert _parity(orc, gpu_ctx_two_rays, vol, env, scene.tf_gradient_source(), (192, 128), pos, d, scene.glibc_rand(3)) > 500


def test_nan_rays_border_hits_and_faces(gpu_ctx_two_rays, orc):
    rng = np.random.default_rng(77)
    coarse = rng.choice(np.array([-1000, 700, 900, 1100, 40], np.int16), size=(8, 8, 8))
    vol = np.ascontiguousarray(np.kron(coarse, np.ones((6, 6, 6), np.int16)).astype(np.int16))
    env = scene.env_map(64, 32)
    pos, d = look_at_centre(vol, [-15, 30, -20])
    assert _parity(orc, gpu_ctx_two_rays, vol, env, scene.tf_default_source(), (128, 96), pos, d, scene.glibc_rand(3)) > 3000
    tf0 = scene.tf_rect_source([(-100.0, 300.0, 0.0, 4000.0, (0.9, 0.6, 0.3, 0.7))])
    rng = np.random.default_rng(123)
    coarse = rng.choice(np.array([-1000, 700, 150, 1100, 40, -50], np.int16), size=(8, 8, 8))
    vol = np.ascontiguousarray(np.kron(coarse, np.ones((6, 6, 6), np.int16)).astype(np.int16))
    assert _parity(orc, gpu_ctx_two_rays, vol, env, tf0, (128, 96), pos, d, scene.glibc_rand(3)) > 3000
    cube = np.full((32, 32, 32), -700, np.int16)
    cube[12:20, 12:20, 12:20] = 1000
    for tf in (scene.tf_default_source(), tf0):
        _parity(orc, gpu_ctx_two_rays, cube, env, tf, (64, 64), np.array([16.0, 16.0, -24.0], np.float32), np.array([0.0, 0.0, 1.0], np.float32),
                scene.glibc_rand(3))


@pytest.mark.parametrize("mode", ["image", "voxel"])
def test_fused_launch_equals_the_passes_one_by_one(gpu_ctx, gpu_ctx_two_rays, orc, mode):
    """12 passes as one launch of k_bounce2 (image space; voxel cache: a planned launch) and as 12 short launches of k_bounce"""
    vol = _ball_in_empty_space(128, 26.0, centre=[60, 70, 58])
    env = scene.env_map(512, 256)
    tf = scene.tf_default_source()
    sdf, _, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    pos, d = look_at_centre(vol, [-40, 170, -50])
    seeds = scene.glibc_rand(12)
    gmode = ffi.ACCUM_IMAGE_SPACE if mode == "image" else ffi.ACCUM_VOXEL_CACHE
    got = []
    for ctx, fused in ((gpu_ctx, False), (gpu_ctx_two_rays, True)):
        g = GpuScene(ctx, vol, sdf, env, tf, (384, 240))
        if fused:
            g.render(pos, d, None, mode=gmode, seeds=seeds, debug=False, write_frame=False)
        else:
            for s in seeds:
                g.render(pos, d, s, mode=gmode, debug=False, write_frame=False)
        got.append((g.accum[0].pull(np.float32).reshape(-1, 4) if mode == "image" else g.cache.pull().reshape(-1, 4)).copy())
        g.release()
    assert got[0][:, 3].max() >= 12
    if mode == "voxel":
        assert got[0][:, 3].max() < 256   # below the cap: the entries are order-independent
    assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize("case", range(8))
def test_random_scenes(gpu_ctx_two_rays, orc, case):
    rng = np.random.default_rng(4000 + case)
    X, Y, Z = (int(rng.integers(40, 150)) for _ in range(3))
    vol = np.full((Z, Y, X), -900, np.int16)
    z, y, x = np.mgrid[0:Z, 0:Y, 0:X].astype(np.float32)
    for _ in range(int(rng.integers(1, 5))):
        c = rng.random(3) * np.array([X, Y, Z])
        r = float(rng.integers(4, max(6, min(X, Y, Z) // 3)))
        vol[(x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2 < r * r] = np.int16(rng.integers(200, 1300))
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
    _parity(orc, gpu_ctx_two_rays, vol, env, tf, (160, 96), pos, d, scene.glibc_rand(3), mode="voxel" if case % 2 else "image")
