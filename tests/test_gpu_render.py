"""-m gpu: the render pass on the MI355X through the C ABI against the CPU oracle.

Parity targets: hit voxel per pixel, per-sample contribution, the voxel cache after K
passes (below the 256-token cap), the deterministic resolve of the frame, the image-space
accumulation -- all bit-exact; the reference's raw racy frame is not a target."""
import numpy as np
import pytest

from cl_volume_renderer_amd import ffi, scene
from tests.gpu_util import GpuScene, look_at_centre, small_scene

pytestmark = pytest.mark.gpu


def _compare_passes(orc, ctx, vol, sdf, env, tf_source, frame_wh, pos, d, seeds, launch_wh=None):
    g = GpuScene(ctx, vol, sdf, env, tf_source, frame_wh, launch_wh)
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf_source), frame_wh, launch_wh)
    for s in seeds:
        g.render(pos, d, s)
        o.render(pos, d, s)
        assert np.array_equal(g.hit_index.pull(), o.hit_index), "primary hit voxel per pixel"
        assert np.array_equal(g.contrib.pull(), o.contrib), "per-sample contribution"
    assert o.cache.reshape(-1, 4)[:, 3].max() < 256, "test must stay below the token cap"
    assert np.array_equal(g.cache.pull(), o.cache), "voxel cache"
    o.resolve(pos, d)
    assert np.array_equal(g.frame.pull(), o.frame), "resolved frame"
    stats = dict(hits=int((o.hit_index >= 0).sum()), touched=int((o.cache.reshape(-1, 4)[:, 3] > 0).sum()))
    g.release()
    return stats


def test_voxel_cache_parity_phantom64(gpu_ctx, orc):
    vol, sdf, env, tf = small_scene(orc, 64)
    pos, d = look_at_centre(vol, [-25, 50, -25])
    st = _compare_passes(orc, gpu_ctx, vol, sdf, env, tf, (256, 256), pos, d, scene.glibc_rand(6))
    assert st["hits"] > 10000 and st["touched"] > 1000


def test_voxel_cache_parity_default_camera_128(gpu_ctx, orc):
    vol, sdf, env, tf = small_scene(orc, 128, env_wh=(1024, 512))
    pos, d = scene.default_camera(128)
    st = _compare_passes(orc, gpu_ctx, vol, sdf, env, tf, (512, 288), pos, d, scene.glibc_rand(8)[4:])
    assert st["hits"] > 10000


def test_gradient_tf_parity(gpu_ctx, orc):
    """the 7-texel step: TF reads `gradient`"""
    vol, sdf, env, tf = small_scene(orc, 48, tf_source=scene.tf_gradient_source())
    pos, d = look_at_centre(vol, [-20, 40, -30])
    st = _compare_passes(orc, gpu_ctx, vol, sdf, env, tf, (128, 128), pos, d, scene.glibc_rand(3))
    assert st["hits"] > 1000


def test_two_material_tf_with_partial_roughness(gpu_ctx, orc):
    tf = scene.tf_rect_source([(500.0, 1200.0, 0.0, 4000.0, (1.0, 0.7, 0.4, 0.35)),
                               (20.0, 60.0, 0.0, 4000.0, (0.3, 0.9, 0.5, 0.9))])
    vol, sdf, env, tf = small_scene(orc, 48, tf_source=tf)
    pos, d = look_at_centre(vol, [-20, 30, -30])
    _compare_passes(orc, gpu_ctx, vol, sdf, env, tf, (128, 128), pos, d, scene.glibc_rand(3))


def test_reference_sdf_test_volume(gpu_ctx, orc, sdf_golden):
    """the only real data the reference ships: tests/sdf/testdata.nrrd with its test TF"""
    vol, gold = sdf_golden
    sdf = gold.reshape(vol.shape).astype(np.int8)
    env = scene.env_map(128, 64)
    pos, d = look_at_centre(vol, [-30, 60, -40])
    # `return (value > 800);` never assigns *color -> zero energy; geometry and token counts still exercise
    _compare_passes(orc, gpu_ctx, vol, sdf, env, scene.TF_TEST_VALUE_GT_800, (128, 128), pos, d, [7, 8])
    tf = scene.tf_rect_source([(800.5, 3000.0, 0.0, 4000.0, (0.9, 0.8, 0.7, 1.0))])
    sdf2, _, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    st = _compare_passes(orc, gpu_ctx, vol, sdf2, env, tf, (128, 128), pos, d, [7, 8, 9])
    assert st["hits"] > 500


@pytest.mark.parametrize("dims", [(72, 40, 56), (96, 40, 56), (144, 24, 40)])
def test_non_cubic_volume_and_wide_frame(gpu_ctx, orc, dims):
    """rows of 72 voxels take k_repack's voxel-by-voxel staging; rows of 96 / 144 (a multiple of 16, not of the 64 voxels a block
    repacks) its 16-byte staging in the first block(s) of a row and the other one in the last"""
    vol, sdf, env, tf = small_scene(orc, 64, dims=dims)
    pos, d = look_at_centre(vol, [-30, 55, -20])
    _compare_passes(orc, gpu_ctx, vol, sdf, env, tf, (320, 64), pos, d, [11, 12])


def test_camera_inside_volume_and_all_miss(gpu_ctx, orc):
    vol, sdf, env, tf = small_scene(orc, 48)
    inside = np.array([24.0, 24.0, 24.0], np.float32)
    _compare_passes(orc, gpu_ctx, vol, sdf, env, tf, (64, 64), inside, scene.camera_direction(0.3, 0.2), [5, 6])
    pos = np.array([-100, 200, -100], np.float32)
    away = np.array([-0.6, 0.5, -0.6], np.float32)
    away /= np.linalg.norm(away)
    st = _compare_passes(orc, gpu_ctx, vol, sdf, env, tf, (64, 64), pos, away.astype(np.float32), [5])
    assert st["hits"] == 0


def test_frame_image_larger_than_launch(gpu_ctx, orc):
    """the reference allocates a 2048x1024 frame whatever the launch size (renderer.cpp:11,145);
    generate_ray uses the IMAGE dims (ray_marching.cl:162)."""
    vol, sdf, env, tf = small_scene(orc, 32)
    pos, d = look_at_centre(vol, [-15, 30, -15])
    _compare_passes(orc, gpu_ctx, vol, sdf, env, tf, (128, 96), pos, d, [3], launch_wh=(64, 48))


def test_generic_launch_is_the_reference_call(gpu_ctx, orc):
    """render_func.execute({w,h},{8,8}, frame, volume, sdf, env, buffer_volume, pos.xyz, dir.xyz, seed)
    (app/renderer.cpp:145-148) through clwh_launch, with the reference's UNPADDED cache allocation."""
    vol, sdf, env, tf = small_scene(orc, 48)
    pos, d = look_at_centre(vol, [-20, 40, -20])
    g = GpuScene(gpu_ctx, vol, sdf, env, tf, (96, 96), padded_cache=False)
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (96, 96))
    for s in scene.glibc_rand(3):
        g.kernel.launch([96, 96], [8, 8], g.frame, g.volume, g.sdf, g.env, g.cache,
                        float(pos[0]), float(pos[1]), float(pos[2]), float(d[0]), float(d[1]), float(d[2]), int(s))
        o.render(pos, d, s)
    n = 48 * 48 * 48 * 4
    assert np.array_equal(g.cache.pull(), o.cache[:n]) and not o.cache[n:].any()
    o.resolve(pos, d)
    assert np.array_equal(g.frame.pull(), o.frame)
    with pytest.raises(ffi.ClwhError) as e:  # 100 is not a multiple of the 8x8 work-group
        g.kernel.launch([100, 96], [8, 8], g.frame, g.volume, g.sdf, g.env, g.cache,
                        0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1)
    assert e.value.status == 8
    g.release()


def test_token_cap_is_reached_exactly(gpu_ctx, orc):
    """utility.cl:20-31: no voxel ever holds more than 256 samples; lanes never carry."""
    vol, sdf, env, tf = small_scene(orc, 32)
    pos, d = look_at_centre(vol, [-12, 25, -12])
    g = GpuScene(gpu_ctx, vol, sdf, env, tf, (256, 256))
    for s in scene.glibc_rand(40):
        g.render(pos, d, s, debug=False)
    c = g.cache.pull().reshape(-1, 4)
    assert c[:, 3].max() == 256 and (c[:, 3] == 256).sum() > 100
    assert c[:, :3].max() <= 256 * 255
    g.release()


def test_image_space_mode_and_tile_partition(gpu_ctx, orc):
    """float4 accumulation per pixel (north_star's multi-GPU mode): exact, and the union of the
    tile-partitioned ranks equals the single-rank result."""
    vol, sdf, env, tf = small_scene(orc, 48)
    pos, d = look_at_centre(vol, [-20, 40, -20])
    w, h = 192, 128
    seeds = scene.glibc_rand(3)
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h), mode=orc.MODE_IMAGE_SPACE)
    one = GpuScene(gpu_ctx, vol, sdf, env, tf, (w, h), world=1)
    for s in seeds:
        o.render(pos, d, s)
        one.render(pos, d, s, mode=ffi.ACCUM_IMAGE_SPACE)
        assert np.array_equal(one.contrib.pull(), o.contrib)
    o.resolve(pos, d)
    got = one.accum_row_major(0)
    hit = o.hit_index.reshape(h, w) >= 0
    assert np.array_equal(got[hit], o.accum[hit])
    assert not got[~hit].any()  # misses never touch the accumulation buffer
    assert np.array_equal(one.frame.pull(), o.frame)

    for world in (2, 3):
        many = GpuScene(gpu_ctx, vol, sdf, env, tf, (w, h), world=world)
        for r in range(world):
            for s in seeds:
                many.render(pos, d, s, mode=ffi.ACCUM_IMAGE_SPACE, rank=r, write_frame=False, debug=False)
        merged = sum(many.accum_row_major(r) for r in range(world))
        assert np.array_equal(merged, got)
        # what the RCCL all-gather produces: the ranks' buffers back to back -> one resolve
        allr = np.concatenate([many.accum[r].pull(np.float32) for r in range(world)])
        buf = gpu_ctx.buffer_from(allr)
        many.frame.push(np.zeros((h, w, 4), np.uint8))
        gpu_ctx.accum_resolve(buf, world, w, h, many.frame, many.env, pos, d)
        assert np.array_equal(many.frame.pull(), o.frame)
        buf.release()
        # what bench.py exchanges instead: every rank resolves ITS tiles to RGBA8 first (4 B per pixel), then the tiles are placed
        n = ffi.accum_len(w, h, world)
        parts = []
        for r in range(world):
            t = gpu_ctx.buffer(n * 4, np.uint32)
            gpu_ctx.accum_resolve_tiles(many.accum[r], r, world, w, h, t, many.env, pos, d)
            parts.append(t.pull(np.uint32))
            t.release()
        tiles_all = gpu_ctx.buffer_from(np.concatenate(parts))
        many.frame.push(np.zeros((h, w, 4), np.uint8))
        gpu_ctx.frame_from_tiles(tiles_all, world, w, h, many.frame)
        assert np.array_equal(many.frame.pull(), o.frame)
        tiles_all.release()
        many.release()
    one.release()


@pytest.mark.parametrize("mode", ["voxel", "image"])
def test_several_seeds_in_one_launch(gpu_ctx, orc, mode):
    """n_seeds passes fused into one launch of the persistent bounce kernel == the same passes one by one."""
    vol, sdf, env, tf = small_scene(orc, 48)
    pos, d = look_at_centre(vol, [-20, 40, -20])
    w, h = 160, 96
    seeds = scene.glibc_rand(7)
    omode = orc.MODE_VOXEL_CACHE if mode == "voxel" else orc.MODE_IMAGE_SPACE
    gmode = ffi.ACCUM_VOXEL_CACHE if mode == "voxel" else ffi.ACCUM_IMAGE_SPACE
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h), mode=omode)
    for s in seeds:
        o.render(pos, d, s)
    o.resolve(pos, d)
    g = GpuScene(gpu_ctx, vol, sdf, env, tf, (w, h))
    g.render(pos, d, None, mode=gmode, seeds=seeds[:4], debug=False)
    g.render(pos, d, None, mode=gmode, seeds=seeds[4:], debug=False)
    if mode == "voxel":
        assert o.cache.reshape(-1, 4)[:, 3].max() < 256
        assert np.array_equal(g.cache.pull(), o.cache)
    else:
        hit = o.hit_index.reshape(h, w) >= 0
        assert np.array_equal(g.accum_row_major(0)[hit], o.accum[hit])
    assert np.array_equal(g.frame.pull(), o.frame)
    g.release()


@pytest.mark.parametrize("refill,step", [(16, 16), (1, 48), (8, 1), (64, 1), (33, 64)])
def test_scheduling_thresholds_do_not_change_results(orc, refill, step, monkeypatch):
    """k_bounce's lane scheduling (idle lanes refill at `refill` idle lanes, the march phase ends below `step`
    marching lanes; chosen per launch by default) is placement only: every setting gives the oracle's bits.
    The knobs are read when a context is created."""
    monkeypatch.setenv("CLWH_TUNE_REFILL", str(refill))
    monkeypatch.setenv("CLWH_TUNE_STEP", str(step))
    ctx = ffi.Context(0)
    vol, sdf, env, tf = small_scene(orc, 48)
    pos, d = look_at_centre(vol, [-20, 40, -20])
    w, h = 160, 96
    seeds = scene.glibc_rand(9)
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h), mode=orc.MODE_IMAGE_SPACE)
    for s in seeds:
        o.render(pos, d, s)
    o.resolve(pos, d)
    g = GpuScene(ctx, vol, sdf, env, tf, (w, h))
    g.render(pos, d, seeds[0], mode=ffi.ACCUM_IMAGE_SPACE)   # one pass, with the per-pixel contribution output
    first = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h), mode=orc.MODE_IMAGE_SPACE)
    first.render(pos, d, seeds[0])
    assert np.array_equal(g.contrib.pull(), first.contrib)
    g.render(pos, d, None, mode=ffi.ACCUM_IMAGE_SPACE, seeds=seeds[1:], debug=False)  # eight passes in one launch
    hit = o.hit_index.reshape(h, w) >= 0
    assert hit.sum() > 1000
    assert np.array_equal(g.accum_row_major(0)[hit], o.accum[hit])
    assert np.array_equal(g.frame.pull(), o.frame)
    g.release()
    ctx.destroy()


def test_primary_hits_follow_camera_volume_and_tf_changes(gpu_ctx, orc):
    """the per-camera primary hits and the packed records are derived data: every input change must
    rebuild them (camera move, TF flush, SDF rebuild, volume push)."""
    vol, sdf, env, tf = small_scene(orc, 40)
    w, h = 96, 96
    g = GpuScene(gpu_ctx, vol, sdf, env, tf, (w, h))
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h))
    cams = [look_at_centre(vol, [-20, 40, -20]), look_at_centre(vol, [60, 10, -30]), look_at_centre(vol, [-20, 40, -20])]
    for k, (pos, d) in enumerate(cams):
        for s in (100 + k, 200 + k):
            g.render(pos, d, s, debug=False)
            o.render(pos, d, s)
    assert np.array_equal(g.cache.pull(), o.cache)
    # TF flush: new kernel object with another TF, new SDF pushed into the same image, cache reset
    tf2 = scene.tf_rect_source([(20.0, 60.0, 0.0, 4000.0, (0.9, 0.5, 0.2, 0.6))])
    sdf2, _, _ = orc.sdf_build(vol, orc.parse_tf(tf2))
    g.sdf.push(sdf2)
    g.kernel.release()
    g.kernel = gpu_ctx.kernel("ray_marching.cl", "render", tf2)
    gpu_ctx.buffer_reset(g.cache)
    o2 = orc.Scene(vol, sdf2, env, orc.parse_tf(tf2), (w, h))
    pos, d = cams[0]
    for s in (7, 8):
        g.render(pos, d, s, debug=False)
        o2.render(pos, d, s)
    assert np.array_equal(g.cache.pull(), o2.cache)
    # volume push (same shape, other content)
    vol3 = np.ascontiguousarray(vol[::-1])
    sdf3, _, _ = orc.sdf_build(vol3, orc.parse_tf(tf2))
    g.volume.push(vol3)
    g.sdf.push(sdf3)
    gpu_ctx.buffer_reset(g.cache)
    o3 = orc.Scene(vol3, sdf3, env, orc.parse_tf(tf2), (w, h))
    g.render(pos, d, 9, debug=False)
    o3.render(pos, d, 9)
    o3.resolve(pos, d)
    assert np.array_equal(g.cache.pull(), o3.cache)
    assert np.array_equal(g.frame.pull(), o3.frame)
    g.release()


def test_uncertified_env_lookups_are_fixed_up_exactly(gpu_ctx, orc):
    """a wide environment map makes the fast lookup's bracket straddle texel boundaries often (~1 %):
    those samples go through the fix-up records + exact binary64 lookup and must still be bit-exact."""
    vol, sdf, env, tf = small_scene(orc, 48, env_wh=(16384, 8192))
    pos, d = look_at_centre(vol, [-20, 40, -20])
    st = _compare_passes(orc, gpu_ctx, vol, sdf, env, tf, (256, 192), pos, d, scene.glibc_rand(4))
    assert st["hits"] > 5000


def test_gradient_tf_literal_taps_agree_with_baked_class(orc):
    """TFs that read `gradient`: the march normally uses the class baked from the gradient at the voxel's
    integer position and only falls back to the reference's literal 7-fetch step where a tap coordinate
    rounds across an integer.  CLWH_TUNE_LITERAL_GRADIENT=1 forces the literal route everywhere: both
    must give the oracle's bits."""
    import os

    tf = scene.tf_rect_source([(500.0, 1200.0, 150.0, 1500.0, (1.0, 0.8, 0.6, 0.5)),
                               (20.0, 60.0, 30.0, 4000.0, (0.2, 0.9, 0.4, 0.8))])
    vol, sdf, env, tf = small_scene(orc, 48, tf_source=tf)
    pos, d = look_at_centre(vol, [-20, 40, -30])
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (128, 128))
    seeds = scene.glibc_rand(3)
    for s in seeds:
        o.render(pos, d, s)
    assert (o.hit_index >= 0).sum() > 1000
    for literal in ("1", None):
        if literal:
            os.environ["CLWH_TUNE_LITERAL_GRADIENT"] = literal
        else:
            os.environ.pop("CLWH_TUNE_LITERAL_GRADIENT", None)
        ctx = ffi.Context(0)
        g = GpuScene(ctx, vol, sdf, env, tf, (128, 128))
        for s in seeds:
            g.render(pos, d, s)
        assert np.array_equal(g.cache.pull(), o.cache), "literal=%s" % literal
        assert np.array_equal(g.hit_index.pull(), o.hit_index)
        g.release()
        ctx.destroy()
    os.environ.pop("CLWH_TUNE_LITERAL_GRADIENT", None)


def test_released_and_reallocated_images_do_not_alias_derived_data(orc):
    """a new volume allocated where a released one lived (same size -> usually the same device address)
    must not be mistaken for the old content by the packed-record / primary-hit caches."""
    ctx = ffi.Context(0)
    tf = scene.tf_default_source()
    env = scene.env_map(128, 64)
    d_env = ctx.image_from(env, channels=4)
    w, h = 96, 96
    frame = ctx.image([w, h], 4, np.uint8, (h, w, 4))
    k = ctx.kernel("ray_marching.cl", "render", tf)
    addresses = []
    for variant in range(2):
        vol = scene.phantom(40, seed=1234 + variant)
        if variant:
            vol = np.ascontiguousarray(vol[:, ::-1])
        sdf, _, _ = orc.sdf_build(vol, orc.parse_tf(tf))
        d_vol, d_sdf = ctx.image_from(vol), ctx.image_from(sdf)
        addresses.append((d_vol.device_ptr, d_sdf.device_ptr))
        cache = ctx.buffer(ffi.cache_len(40, 40, 40) * 2, np.uint16)
        ctx.buffer_reset(cache)
        pos, d = look_at_centre(vol, [-18, 35, -16])
        o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h))
        for s in (11, 12):
            k.render(frame=frame, volume=d_vol, sdf=d_sdf, env=d_env, buffer_volume=cache, cam_pos=pos, cam_dir=d,
                     seed=s, width=w, height=h)
            o.render(pos, d, s)
        assert np.array_equal(cache.pull(), o.cache), "variant %d" % variant
        for m in (d_vol, d_sdf, cache):
            m.release()
    # informational: on this allocator the second pair normally reuses the first pair's addresses
    print("device addresses", addresses)
    ctx.destroy()


FREE_FORM_TF = ("inline bool is_event_gen(short value, short gradient, int4 *color){\n"
                "  if (value > 800 || value < -1010) {\n    int4 tmp_color = {200,100,50,255};\n    *color = tmp_color;\n"
                "    return true;\n  }\n  if (value >= 30 && value <= 45) { int4 c2 = {20,220,120,128}; *color = c2; return true; }\n"
                "  return false;\n}\n")


def _free_form_tf_as_rules(orc):
    """the same function as first-match rules, for the oracle (its table cannot say `||`)"""
    tf = orc.Tf()
    spec = [(801, 32767, (200, 100, 50, 255)), (-32768, -1011, (200, 100, 50, 255)), (30, 45, (20, 220, 120, 128))]
    for k, (lo, hi, col) in enumerate(spec):
        r = tf.rules[k]
        r.v_lo, r.v_hi, r.g_lo, r.g_hi = lo, hi, -32768, 32767
        r.use_gradient, r.writes_color, r.terminal = 0, 1, 0
        for q in range(4):
            r.color[q] = col[q]
    tf.n = len(spec)
    return tf


def test_free_form_tf_source_goes_through_hiprtc(gpu_ctx, orc):
    """source outside the rule grammar (`||`, a differently named temporary) is compiled with hiprtc into the
    per-voxel classifier ("hiprtc as the general fallback"); SDF, cache and frame stay bit-exact."""
    with pytest.raises(ffi.ClwhError):
        ffi.parse_tf(FREE_FORM_TF)                      # the rule parser refuses it ...
    rules = _free_form_tf_as_rules(orc)
    vol = scene.phantom(48)
    env = scene.env_map(256, 128)
    want_sdf, n_want, _ = orc.sdf_build(vol, rules)
    g = GpuScene(gpu_ctx, vol, None, env, FREE_FORM_TF, (160, 128))   # ... clwh_kernel_get compiles it
    assert gpu_ctx.sdf_build(g.volume, FREE_FORM_TF, g.sdf) == n_want
    assert np.array_equal(g.sdf.pull(), want_sdf)
    pos, d = look_at_centre(vol, [-20, 40, -20])
    o = orc.Scene(vol, want_sdf, env, rules, (160, 128))
    for s in scene.glibc_rand(3):
        g.render(pos, d, s)
        o.render(pos, d, s)
        assert np.array_equal(g.contrib.pull(), o.contrib)
    assert (o.hit_index >= 0).sum() > 3000
    assert np.array_equal(g.cache.pull(), o.cache)
    o.resolve(pos, d)
    assert np.array_equal(g.frame.pull(), o.frame)
    g.release()


def test_uncompilable_tf_source_is_refused(gpu_ctx):
    with pytest.raises(ffi.ClwhError) as e:
        gpu_ctx.kernel("ray_marching.cl", "render", "inline bool is_event_gen(short value, short gradient, int4 *color){ return undefined_symbol(value); }")
    assert e.value.status == 6  # CLWH_ERR_TF_UNSUPPORTED


def test_ambient_occlusion_mode_matches_the_oracle(gpu_ctx, orc):
    """compute_ao (ray_marching.cl:104-149), the reference's alternate shading function: {samples, occluded} per
    voxel in the 2-ushort view of buffer_volume (utility.cl:123-159), cap 100; per-pixel occlusion flag, the cache
    below the cap and the resolved frame are bit-exact; several passes in one launch equal pass-by-pass."""
    vol, sdf, env, tf = small_scene(orc, 64)
    pos, d = look_at_centre(vol, [-25, 50, -25])
    w, h = 192, 128
    g = GpuScene(gpu_ctx, vol, sdf, env, tf, (w, h))
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h), shading=orc.SHADE_AO)
    seeds = scene.glibc_rand(4)
    for s in seeds:
        g.render(pos, d, s, shading=ffi.SHADE_AO)
        o.render(pos, d, s)
        assert np.array_equal(g.hit_index.pull(), o.hit_index)
        assert np.array_equal(g.contrib.pull(), o.contrib)
    pairs = o.cache[: o.cache.size // 2].reshape(-1, 2)
    assert 0 < pairs[:, 0].max() < 100, "below the cap of 100 samples per voxel"
    assert pairs[:, 1].max() > 0, "some AO rays must be occluded"
    assert np.array_equal(g.cache.pull(), o.cache)
    o.resolve(pos, d)
    frame = g.frame.pull()
    assert np.array_equal(frame, o.frame)
    hit = o.hit_index.reshape(h, w) >= 0
    assert (frame[hit][:, 3] == 1).all() and (frame[~hit][:, 3] == 200).all()
    gpu_ctx.buffer_reset(g.cache)
    g.render(pos, d, None, shading=ffi.SHADE_AO, seeds=seeds, debug=False)
    assert np.array_equal(g.cache.pull(), o.cache)
    assert np.array_equal(g.frame.pull(), o.frame)
    g.release()


def test_ambient_occlusion_cap_of_100_samples(gpu_ctx, orc):
    """beyond the cap the reference is order-dependent (racy read-modify-write); what must hold: no voxel exceeds
    100 samples, occluded <= samples, and voxels below the cap equal the oracle's."""
    vol, sdf, env, tf = small_scene(orc, 32)
    pos, d = look_at_centre(vol, [-12, 25, -12])
    w, h = 256, 256
    g = GpuScene(gpu_ctx, vol, sdf, env, tf, (w, h))
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h), shading=orc.SHADE_AO)
    seeds = scene.glibc_rand(6)
    g.render(pos, d, None, shading=ffi.SHADE_AO, seeds=seeds, debug=False)
    for s in seeds:
        o.render(pos, d, s)
    got = g.cache.pull()[: o.cache.size // 2].reshape(-1, 2)
    want = o.cache[: o.cache.size // 2].reshape(-1, 2)
    assert got[:, 0].max() == 100 and want[:, 0].max() == 100
    assert (got[:, 1] <= got[:, 0]).all()
    assert np.array_equal(got[:, 0], want[:, 0])          # sample counts: min(pixel-samples, 100) either way
    below = want[:, 0] < 100
    assert np.array_equal(got[below], want[below])
    g.release()


def test_frame_handoff_through_the_c_abi_without_a_pull(orc):
    """clwh_image_wrap + clwh_ctx_acquire_from / clwh_ctx_release_to: the frame lives in memory another owner
    allocated and reads on its own stream; no clwh_mem_pull of the frame anywhere."""
    import torch

    vol, sdf, env, tf = small_scene(orc, 48)
    pos, d = look_at_centre(vol, [-20, 40, -30])
    w, h = 128, 96
    ctx = ffi.Context(0)                       # its own non-blocking stream
    g = GpuScene(ctx, vol, sdf, env, tf, (w, h))
    own_frame = g.frame
    consumer = torch.cuda.Stream()
    target = torch.full((h, w, 4), 7, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    g.frame = ctx.image_wrap(target.data_ptr(), [w, h], 4, np.uint8, (h, w, 4))
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h))
    for s in scene.glibc_rand(3):
        ctx.acquire_from(consumer.cuda_stream)
        g.render(pos, d, s, debug=False)
        ctx.release_to(consumer.cuda_stream)
        with torch.cuda.stream(consumer):
            seen = target.clone()
        o.render(pos, d, s)
        o.resolve(pos, d)
        consumer.synchronize()
        assert np.array_equal(seen.cpu().numpy(), o.frame)
    g.frame.release()
    g.frame = own_frame
    g.release()
    ctx.destroy()


@pytest.mark.parametrize("fused", [False, True])
def test_token_cap_regime_against_the_oracle(gpu_ctx, orc, fused):
    """the cap regime had no oracle comparison.  Beyond the cap the reference is order-dependent in WHICH
    256 contributions a voxel keeps, but not in how many: every request is granted while the count is below 256, so the
    count of every voxel is min(requests, 256) in any order -- equal to the oracle's; voxels below the cap are equal
    entry for entry; a capped voxel holds 256 contributions of at most 255.  Pass by pass, and all passes fused into
    one launch (clwh_render_desc.n_seeds)."""
    vol, sdf, env, tf = small_scene(orc, 32)
    pos, d = look_at_centre(vol, [-12, 25, -12])
    g = GpuScene(gpu_ctx, vol, sdf, env, tf, (256, 256))
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (256, 256))
    seeds = scene.glibc_rand(40)
    if fused:
        g.render(pos, d, None, seeds=seeds, debug=False)
    else:
        for s in seeds:
            g.render(pos, d, s, debug=False)
    for s in seeds:
        o.render(pos, d, s)
    got, want = g.cache.pull().reshape(-1, 4), o.cache.reshape(-1, 4)
    assert want[:, 3].max() == 256 and (want[:, 3] == 256).sum() > 100 and ((want[:, 3] > 0) & (want[:, 3] < 256)).sum() > 100
    assert np.array_equal(got[:, 3], want[:, 3])
    below = want[:, 3] < 256
    assert np.array_equal(got[below], want[below])
    assert got[~below, :3].max() <= 256 * 255
    g.release()
