"""-m gpu: degenerate and adversarial inputs through the C ABI against the oracle: tiny volumes and frames,
zero-gradient hits (NaN normals: normalize(0), SURVEY "hard parts"), random multi-rectangle transfer functions."""
import numpy as np
import pytest

from cl_volume_renderer_amd import ffi, scene
from tests.gpu_util import GpuScene, look_at_centre

pytestmark = pytest.mark.gpu


def _parity(orc, ctx, vol, env, tf, frame_wh, pos, d, seeds, mode="voxel"):
    sdf, _, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    g = GpuScene(ctx, vol, None, env, tf, frame_wh)
    ctx.sdf_build(g.volume, tf, g.sdf)
    assert np.array_equal(g.sdf.pull(), sdf)
    omode = orc.MODE_VOXEL_CACHE if mode == "voxel" else orc.MODE_IMAGE_SPACE
    gmode = ffi.ACCUM_VOXEL_CACHE if mode == "voxel" else ffi.ACCUM_IMAGE_SPACE
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), frame_wh, mode=omode)
    for s in seeds:
        g.render(pos, d, s, mode=gmode)
        o.render(pos, d, s)
        assert np.array_equal(g.hit_index.pull(), o.hit_index)
        assert np.array_equal(g.contrib.pull(), o.contrib)
    if mode == "voxel":
        assert np.array_equal(g.cache.pull(), o.cache)
    else:
        hit = o.hit_index.reshape(frame_wh[1], frame_wh[0]) >= 0
        assert np.array_equal(g.accum_row_major(0)[hit], o.accum[hit])
    o.resolve(pos, d)
    assert np.array_equal(g.frame.pull(), o.frame)
    hits = int((o.hit_index >= 0).sum())
    g.release()
    return hits


@pytest.mark.parametrize("dims", [(2, 2, 2), (3, 2, 5), (9, 5, 3), (16, 1, 16)])
def test_tiny_volumes_and_frames(gpu_ctx, orc, dims):
    """the reference's placeholder SDF is 2x2x2 (signed_distance_field.cpp:43-45); max_iterations is 0 or 1 here"""
    X, Y, Z = dims
    rng = np.random.default_rng(X * 100 + Y * 10 + Z)
    vol = rng.integers(400, 1400, size=(Z, Y, X), dtype=np.int16)
    env = scene.env_map(4, 2)
    pos = np.array([-3.0, Y + 4.0, -2.5], np.float32)
    _, d = look_at_centre(vol, pos)
    _parity(orc, gpu_ctx, vol, env, scene.tf_default_source(), (8, 8), pos, d, [1, 2, 3])
    _parity(orc, gpu_ctx, vol, env, scene.tf_default_source(), (16, 8), pos, d, [5], mode="image")


def test_zero_gradient_hits_give_nan_normals_like_the_oracle(gpu_ctx, orc):
    """piecewise-constant volume: hits inside constant blocks have a zero central-difference gradient, so
    normal = -normalize(0) = NaN; the NaN ray never exits and contributes nothing (ray_marching.cl:42,64;
    utility_ray.cl:114-116).  Both sides must agree bit for bit on what that does to the cache."""
    rng = np.random.default_rng(77)
    coarse = rng.choice(np.array([-1000, 700, 900, 1100, 40], np.int16), size=(8, 8, 8))
    vol = np.ascontiguousarray(np.kron(coarse, np.ones((6, 6, 6), np.int16)).astype(np.int16))   # 48^3, 6^3 blocks
    env = scene.env_map(64, 32)
    pos, d = look_at_centre(vol, [-15, 30, -20])
    hits = _parity(orc, gpu_ctx, vol, env, scene.tf_default_source(), (128, 96), pos, d, scene.glibc_rand(3))
    assert hits > 3000


@pytest.mark.parametrize("case", range(6))
def test_random_rectangle_transfer_functions(gpu_ctx, orc, case):
    """1-3 overlapping rectangles with random value / gradient windows, colours and roughness: first match
    wins, gradient clauses are baked or evaluated literally, colours carry over between distribution rays."""
    rng = np.random.default_rng(1000 + case)
    rects = []
    for _ in range(int(rng.integers(1, 4))):
        lo = float(rng.integers(-1100, 1000))
        hi = lo + float(rng.integers(20, 900))
        if rng.random() < 0.5:
            glo, ghi = float(rng.integers(0, 400)), float(rng.integers(600, 3999))
        else:
            glo, ghi = 0.0, 4000.0
        rects.append((lo + 0.5 * float(rng.integers(0, 2)), hi, glo, ghi, tuple(float(x) for x in rng.random(4))))
    tf = scene.tf_rect_source(rects)
    vol = scene.phantom(40, seed=case)
    vol[5:15, 20:30, 8:30] = 300                      # a second, flat material
    env = scene.env_map(128, 64, seed=case)
    pos, d = look_at_centre(vol, [float(rng.integers(-30, -5)), float(rng.integers(20, 60)), float(rng.integers(-30, -5))])
    _parity(orc, gpu_ctx, vol, env, tf, (96, 64), pos, d, scene.glibc_rand(2))
    _parity(orc, gpu_ctx, vol, env, tf, (96, 64), pos, d, [int(rng.integers(0, 2**31 - 1))], mode="image")


def test_env_map_swapped_under_a_fixed_camera(gpu_ctx, orc):
    """the per-camera primary hits keep the env colour of every miss pixel; a different environment
    map -- a new object, or new content pushed into the same object -- with the same camera, volume and frame
    size must not resolve the misses with the old map's colours."""
    vol = scene.phantom(40)
    tf = scene.tf_default_source()
    sdf, _, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    env_a, env_b = scene.env_map(128, 64, seed=1), scene.env_map(128, 64, seed=2)[:, ::-1].copy()
    assert not np.array_equal(env_a, env_b)
    pos, d = look_at_centre(vol, [-30, 35, -25])
    w, h = 96, 64
    g = GpuScene(gpu_ctx, vol, sdf, env_a, tf, (w, h))

    def want(env, seeds):
        o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h))
        for s in seeds:
            o.render(pos, d, s)
        o.resolve(pos, d)
        return o

    g.render(pos, d, 11, debug=False)   # debug outputs would force a primary rebuild
    oa = want(env_a, [11])
    assert np.array_equal(g.frame.pull(), oa.frame)
    misses = oa.frame[..., 3] == 200
    assert misses.sum() > 500
    # (1) new content pushed into the SAME image object
    g.env.push(env_b)
    gpu_ctx.buffer_reset(g.cache)
    g.render(pos, d, 11, debug=False)   # debug outputs would force a primary rebuild
    ob = want(env_b, [11])
    assert not np.array_equal(oa.frame[misses], ob.frame[misses])
    assert np.array_equal(g.frame.pull(), ob.frame)
    assert np.array_equal(g.cache.pull(), ob.cache)
    # (2) a different image object (renderer::image_set with another env_map)
    other = gpu_ctx.image_from(env_a, channels=4)
    old, g.env = g.env, other
    gpu_ctx.buffer_reset(g.cache)
    g.render(pos, d, 11, debug=False)   # debug outputs would force a primary rebuild
    assert np.array_equal(g.frame.pull(), oa.frame)
    old.release()
    g.release()


def test_extreme_voxel_values_fill_the_17_bit_hit_record_fields(gpu_ctx, orc):
    """hit records keep the voxel's central differences in 17 signed bits (csrc/packed_volume.hpp): a volume that
    alternates between the int16 extremes produces differences of +-65535 and +-32768 against the zero border; normals,
    contributions and cache must still equal the oracle's six literal taps."""
    rng = np.random.default_rng(5)
    vol = rng.choice(np.array([-32768, 32767, -32768, 32767, 0, 700, 1100], np.int16), size=(24, 24, 24)).astype(np.int16)
    vol[8:16, 8:16, 8:16] = 32767
    env = scene.env_map(64, 32)
    tf = scene.tf_rect_source([(20000.0, 32767.0, 0.0, 4000.0, (1.0, 0.5, 0.25, 0.8)),
                               (600.0, 1200.0, 0.0, 4000.0, (0.2, 0.9, 0.4, 0.3))])
    pos, d = look_at_centre(vol, [-10, 30, -14])
    hits = _parity(orc, gpu_ctx, vol, env, tf, (96, 96), pos, d, scene.glibc_rand(3))
    assert hits > 2000
    _parity(orc, gpu_ctx, vol, env, tf, (96, 96), pos, d, [77], mode="image")


def test_transfer_function_that_matches_the_border_value_zero(gpu_ctx, orc):
    """read_imagei outside the volume returns the border value 0 (CLK_ADDRESS_CLAMP), and a position with a NaN
    coordinate or with a coordinate exactly equal to the dimension is still 'inside' for exited_volume
    (utility_ray.cl:112-117) -- so a transfer function whose value range contains 0 turns such a position into a Hit
    on the border.  Piecewise-constant volume (zero gradients -> NaN normals -> NaN rays) with such a TF."""
    rng = np.random.default_rng(123)
    coarse = rng.choice(np.array([-1000, 700, 150, 1100, 40, -50], np.int16), size=(8, 8, 8))
    vol = np.ascontiguousarray(np.kron(coarse, np.ones((6, 6, 6), np.int16)).astype(np.int16))
    env = scene.env_map(64, 32)
    tf = scene.tf_rect_source([(-100.0, 300.0, 0.0, 4000.0, (0.9, 0.6, 0.3, 0.7))])
    pos, d = look_at_centre(vol, [-15, 30, -20])
    hits = _parity(orc, gpu_ctx, vol, env, tf, (128, 96), pos, d, scene.glibc_rand(3))
    assert hits > 3000
    # the same with a rule that reads `gradient` (the literal 7-fetch route at irregular positions)
    tf2 = scene.tf_rect_source([(-100.0, 300.0, -1.0, 3000.0, (0.9, 0.6, 0.3, 0.7))], stats=(-2000.0, 3000.0, 0.0, 4000.0))
    _parity(orc, gpu_ctx, vol, env, tf2, (128, 96), pos, d, scene.glibc_rand(2))


def test_free_form_tf_that_matches_the_border_value(gpu_ctx, orc):
    """the same border case through the hiprtc route: a free-form source (outside the rule grammar) whose event set
    contains value 0; the compiled classifier also evaluates is_event_gen(0, 0) for the border texel."""
    source = ("inline bool is_event_gen(short value, short gradient, int4 *color){\n"
              "  int4 shade = {229, 153, 76, 178};\n"
              "  if(value >= -100 && value <= 300 || value == 31000) { *color = shade; return true; }\n"
              "  return false;\n}\n")
    with pytest.raises(ffi.ClwhError):
        ffi.parse_tf(source)
    rules = orc.parse_tf(scene.tf_rect_source([(-100.0, 300.0, 0.0, 4000.0, (0.9, 0.6, 0.3, 0.7))]))
    assert tuple(rules.rules[0].color) == (229, 153, 76, 178)
    rng = np.random.default_rng(123)
    coarse = rng.choice(np.array([-1000, 700, 150, 1100, 40, -50], np.int16), size=(8, 8, 8))
    vol = np.ascontiguousarray(np.kron(coarse, np.ones((6, 6, 6), np.int16)).astype(np.int16))
    env = scene.env_map(64, 32)
    pos, d = look_at_centre(vol, [-15, 30, -20])
    sdf, _, _ = orc.sdf_build(vol, rules)
    g = GpuScene(gpu_ctx, vol, None, env, source, (128, 96))
    gpu_ctx.sdf_build(g.volume, source, g.sdf)
    assert np.array_equal(g.sdf.pull(), sdf)
    o = orc.Scene(vol, sdf, env, rules, (128, 96))
    for s in scene.glibc_rand(3):
        g.render(pos, d, s)
        o.render(pos, d, s)
        assert np.array_equal(g.hit_index.pull(), o.hit_index)
        assert np.array_equal(g.contrib.pull(), o.contrib)
    assert np.array_equal(g.cache.pull(), o.cache)
    g.release()
