"""-m gpu: SDF builder on the MI355X through the C ABI, against the reference's golden vector and the oracle."""
import numpy as np
import pytest

from cl_volume_renderer_amd import ffi, scene

pytestmark = pytest.mark.gpu


def _upload(ctx, vol):
    Z, Y, X = vol.shape
    v = ctx.image_from(vol.astype(np.int16))
    s = ctx.image([X, Y, Z], 1, np.int8, (Z, Y, X))
    return v, s


def test_fused_build_matches_reference_golden_vector(gpu_ctx, sdf_golden):
    """reference tests/sdf/sdf_test.cpp:6-33 re-run on the HIP path: exact."""
    vol, gold = sdf_golden
    v, s = _upload(gpu_ctx, vol)
    n = gpu_ctx.sdf_build(v, scene.TF_TEST_VALUE_GT_800, s)
    out = s.pull()
    assert np.array_equal(out.reshape(-1).astype(np.int32), gold)
    assert n == 13
    v.release(); s.release()


def test_reference_host_loop_over_the_generic_launch(gpu_ctx, sdf_golden):
    """app/signed_distance_field.cpp:7-35 re-enacted call by call over clwh_kernel_get / clwh_launch,
    i.e. what the reference's unmodified host code would do through the clw_* wrappers."""
    vol, gold = sdf_golden
    Z, Y, X = vol.shape
    ctx = gpu_ctx
    v = ctx.image_from(vol.astype(np.int16))
    sdf = ctx.image([X, Y, Z], 1, np.int8, (Z, Y, X)); sdf.push(np.zeros((Z, Y, X), np.int8))
    pong = ctx.image([X, Y, Z], 1, np.int8, (Z, Y, X))
    code = scene.TF_TEST_VALUE_GT_800
    max_it = min(max(X, Y, Z) // 2, 127)
    ev = lambda g, l: (g + l - 1) // l * l  # noqa: E731  evenness(), app/common.hpp:59-66
    gsz = [ev(X, 8), ev(Y, 8), ev(Z, 8)]
    ctx.kernel("signed_distance_field.cl", "create_base_image", code).launch(gsz, [4, 4, 4], v, sdf, pong, np.uint32(max_it))
    layer = ctx.kernel("signed_distance_field.cl", "create_signed_distance_field", code)
    counter = ctx.buffer(4, np.int32)
    ping_, pong_ = sdf, pong
    launches = 0
    for i in range(1, max_it + (max_it % 2) + 1 + 1):
        counter.push(np.zeros(1, np.int32))
        layer.launch(gsz, [4, 4, 4], ping_, pong_, np.uint32(i), counter, np.uint32(max_it))
        ping_, pong_ = pong_, ping_
        launches += 1
        if counter.pull()[0] == 0 and i % 2 == 1:
            break
    assert launches == 13
    assert np.array_equal(sdf.pull().reshape(-1).astype(np.int32), gold)
    for m in (v, sdf, pong, counter):
        m.release()


@pytest.mark.parametrize("dims,tf", [
    ((64, 64, 64), "default"), ((64, 64, 64), "gradient"), ((70, 33, 45), "default"), ((40, 96, 24), "gradient"),
])
def test_fused_build_matches_oracle(gpu_ctx, orc, dims, tf):
    src = scene.tf_default_source() if tf == "default" else scene.tf_gradient_source()
    vol = scene.phantom(max(dims), dims=dims)
    want, n_want, _ = orc.sdf_build(vol, orc.parse_tf(src))
    v, s = _upload(gpu_ctx, vol)
    n = gpu_ctx.sdf_build(v, src, s)
    got = s.pull()
    assert np.array_equal(got, want)
    assert n == n_want
    v.release(); s.release()


def test_degenerate_volumes(gpu_ctx, orc):
    """all-event and no-event volumes: nothing to propagate, the loop stops at the first odd layer."""
    for fill in (-1000, 900):
        vol = np.full((16, 16, 16), fill, np.int16)
        want, n_want, _ = orc.sdf_build(vol, orc.parse_tf(scene.tf_default_source()))
        v, s = _upload(gpu_ctx, vol)
        n = gpu_ctx.sdf_build(v, scene.tf_default_source(), s)
        assert np.array_equal(s.pull(), want) and n == n_want == 1
        v.release(); s.release()


def test_unknown_kernel_and_bad_ndrange_are_reported(gpu_ctx):
    with pytest.raises(ffi.ClwhError) as e:
        gpu_ctx.kernel("2d_image_filter.cl", "bilateral_filter")
    assert e.value.status == 5
    k = gpu_ctx.kernel("empty.cl", "empty")
    with pytest.raises(ffi.ClwhError) as e:
        k.launch([10, 8, 1], [4, 4, 1])  # clw_function.hpp:235 asserts global % local == 0
    assert e.value.status == 8
    k.release()


def test_full_sdf_at_256_equals_the_oracle(gpu_ctx, orc):
    """every one of the 16.8 M values and the reference's launch count, against the oracle run here (about 15 s of CPU)"""
    vol = scene.phantom(256)
    tf = scene.tf_default_source()
    v, s = _upload(gpu_ctx, vol)
    n = gpu_ctx.sdf_build(v, tf, s)
    want, launches, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    assert np.array_equal(s.pull(), want)
    assert n == launches
    v.release(); s.release()


@pytest.mark.parametrize("tf_name", ["default", "gradient"])
def test_sdf_at_512_equals_the_committed_oracle_checksum(gpu_ctx, golden_dir, tf_name):
    """the SDF the headline benchmark marches through (512^3; the 7-tap gradient TF of config 3 too): SHA-256, sums
    and launch count of the ORACLE's field, computed once in the build container (5 min of CPU each,
    tests/golden/make_sdf_checksums.py) and committed -- byte-for-byte equality without re-running the oracle."""
    import hashlib
    import json
    import os

    want = json.load(open(os.path.join(golden_dir, "sdf_phantom_checksums.json")))["phantom(512) %s tf" % tf_name]
    vol = scene.phantom(512)
    tf = scene.tf_default_source() if tf_name == "default" else scene.tf_gradient_source()
    v, s = _upload(gpu_ctx, vol)
    n = gpu_ctx.sdf_build(v, tf, s)
    sdf = s.pull()
    assert n == want["launches"]
    assert int(sdf.astype(np.int64).sum()) == want["sum"]
    assert hashlib.sha256(sdf.tobytes()).hexdigest() == want["sha256"]
    v.release(); s.release()


def _blobs(dims, seed):
    """random smooth blobs: fronts in every direction, seeds in all four parity classes, surfaces that meet the volume's faces"""
    rng = np.random.default_rng(seed)
    Z, Y, X = dims[2], dims[1], dims[0]
    z, y, x = np.mgrid[0:Z, 0:Y, 0:X].astype(np.float32)
    vol = np.full((Z, Y, X), -900.0, np.float32)
    for _ in range(5):
        c = rng.uniform(0, 1, 3) * np.array([X, Y, Z])
        r = rng.uniform(0.08, 0.3) * max(dims)
        vol = np.maximum(vol, 1000.0 - 60.0 * (np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) - r))
    return np.ascontiguousarray(np.clip(vol, -1000, 1100).astype(np.int16))


@pytest.mark.parametrize("variant", ["bits", "bits16", "bits_rec_lds", "front"])
@pytest.mark.parametrize("dims", [(200, 170, 150), (65, 49, 17), (2, 40, 40), (130, 3, 7), (31, 97, 129)])
def test_every_build_variant_on_awkward_shapes(gpu_ctx, gpu_ctx_sdf_waves16, gpu_ctx_sdf_rec_lds, gpu_ctx_sdf_front, orc, variant, dims):
    """The bit-parallel build works on regions of 64 x 48 x 16 (or 48) voxels with 8-voxel halos, rows of 32-bit words and a
    clamped neighbourhood at the faces: sizes that are no multiple of any of these, one-voxel-thick volumes, and surfaces that
    run into the faces -- every value and the reference's launch count against the oracle, for all build paths (the 8-wave
    blocks also with their layer records in LDS)."""
    ctx = {"bits": gpu_ctx, "bits16": gpu_ctx_sdf_waves16, "bits_rec_lds": gpu_ctx_sdf_rec_lds, "front": gpu_ctx_sdf_front}[variant]
    vol = _blobs(dims, seed=sum(dims))
    tf = scene.tf_default_source()
    want, n_want, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    v, s = _upload(ctx, vol)
    n = ctx.sdf_build(v, tf, s)
    got = s.pull()
    assert np.array_equal(got, want)
    assert n == n_want
    assert (np.abs(want.astype(np.int32)) > 1).any() or min(dims) < 4  # something was propagated
    v.release(); s.release()


def test_rebuild_after_a_transfer_function_change(gpu_ctx, orc):
    """the build's scratch (bit sets, block states, lists) is reused across builds: a second build with another table must not see the first"""
    vol = scene.phantom(96)
    v, s = _upload(gpu_ctx, vol)
    for src in (scene.tf_default_source(), scene.tf_gradient_source(), scene.tf_default_source()):
        want, n_want, _ = orc.sdf_build(vol, orc.parse_tf(src))
        n = gpu_ctx.sdf_build(v, src, s)
        assert np.array_equal(s.pull(), want) and n == n_want
    v.release(); s.release()
