"""-m gpu: BASELINE.json's config 2 at full size (512^3 volume, 1920x1080 frame, default camera):
the HIP path against the oracle on whole passes, plus size-independent properties of the job."""
import os

import numpy as np
import pytest

from cl_volume_renderer_amd import ffi, scene
from tests.gpu_util import GpuScene

pytestmark = pytest.mark.gpu

N, W, H = 512, 1920, 1080


@pytest.fixture(scope="module")
def full(gpu_ctx):
    vol = scene.phantom(N)
    env = scene.env_map(4096, 2048)
    tf = scene.tf_default_source()
    g = GpuScene(gpu_ctx, vol, None, env, tf, (W, H))
    n_layers = gpu_ctx.sdf_build(g.volume, tf, g.sdf)
    pos, d = scene.default_camera(N)
    yield dict(g=g, vol=vol, env=env, tf=tf, pos=pos, d=d, n_layers=n_layers, sdf=g.sdf.pull())
    g.release()


def test_sdf_properties_at_512(full):
    """properties of the corner-neighbour distance field that hold at any size: sign == event class,
    |v| in [1, 127], a settled voxel of value k > 1 has a corner neighbour holding k - 1, and the loop
    ends on an odd layer."""
    sdf, vol = full["sdf"], full["vol"]
    event = (vol >= 500) & (vol <= 1200)
    assert np.array_equal(sdf < 0, event)
    a = np.abs(sdf.astype(np.int16))
    assert a.min() == 1 and a.max() == 127
    assert full["n_layers"] % 2 == 1
    # check the recurrence on a z-slab (clamped corner neighbours)
    z0, z1 = 200, 232
    s = a[z0 - 1:z1 + 1]
    core = s[1:-1, 1:-1, 1:-1]
    m = np.full(core.shape, 127, np.int16)
    for dz in (0, 2):
        for dy in (0, 2):
            for dx in (0, 2):
                m = np.minimum(m, s[dz:dz + core.shape[0], dy:dy + core.shape[1], dx:dx + core.shape[2]])
    settled = (core > 1) & (core < 127)
    assert np.array_equal(core[settled], m[settled] + 1)


def test_full_size_passes_match_the_oracle(full, orc):
    g, pos, d = full["g"], full["pos"], full["d"]
    threads = min(16, len(os.sched_getaffinity(0)))
    o = orc.Scene(full["vol"], full["sdf"], full["env"], orc.parse_tf(full["tf"]), (W, H), threads=threads)
    gpu = g.ctx
    gpu.buffer_reset(g.cache)
    seeds = scene.glibc_rand(2)
    for s in seeds:
        g.render(pos, d, s)
        o.render(pos, d, s)
    assert np.array_equal(g.hit_index.pull(), o.hit_index)
    assert np.array_equal(g.contrib.pull(), o.contrib)       # last pass, every pixel
    cache = g.cache.pull()
    assert cache.reshape(-1, 4)[:, 3].max() < 256
    assert np.array_equal(cache, o.cache)
    o.resolve(pos, d)
    assert np.array_equal(g.frame.pull(), o.frame)
    full["hits"] = int((o.hit_index >= 0).sum())
    assert full["hits"] > 300000


def test_job_level_properties_without_the_oracle(full):
    """64 spp in image-space mode as bench.py runs it (4 launches of 16 seeds) vs 64 single-seed launches
    of 1 seed in voxel-cache mode: counts, checksums and fused == unfused."""
    g, pos, d = full["g"], full["pos"], full["d"]
    ctx = g.ctx
    seeds = scene.glibc_rand(64)
    for a in g.accum:
        ctx.buffer_reset(a)
    for k in range(0, 64, 16):
        g.render(pos, d, None, mode=ffi.ACCUM_IMAGE_SPACE, seeds=seeds[k:k + 16], debug=False, write_frame=False)
    fused = g.accum[0].pull(np.float32).reshape(-1, 4).copy()
    hit = fused[:, 3] > 0
    assert set(np.unique(fused[:, 3])) == {0.0, 64.0}        # every hit pixel got exactly one sample per pass
    assert not fused[~hit].any()
    assert fused[hit, :3].max() <= 64 * 255
    ctx.buffer_reset(g.accum[0])
    for s in seeds[:8]:
        g.render(pos, d, s, mode=ffi.ACCUM_IMAGE_SPACE, debug=False, write_frame=False)
    eight = g.accum[0].pull(np.float32).reshape(-1, 4).copy()
    ctx.buffer_reset(g.accum[0])
    g.render(pos, d, None, mode=ffi.ACCUM_IMAGE_SPACE, seeds=seeds[:8], debug=False, write_frame=False)
    assert np.array_equal(g.accum[0].pull(np.float32).reshape(-1, 4), eight)   # fused == one by one
    # voxel-cache mode: the token cap holds, and below the cap the cache is the scatter-add of the image-space sums
    ctx.buffer_reset(g.cache)
    for s in seeds[:8]:
        g.render(pos, d, s, debug=False, write_frame=False)
    cache = g.cache.pull().reshape(-1, 4).astype(np.int64)
    assert cache[:, 3].max() <= 256
    assert cache[:, 3].sum() <= 8 * int(hit.sum())
    if cache[:, 3].max() < 256:
        assert cache[:, 3].sum() == 8 * int(hit.sum())
        assert np.array_equal(cache[:, :3].sum(axis=0), eight[:, :3].astype(np.int64).sum(axis=0))


def test_large_volume_64bit_indexing(gpu_ctx, orc):
    """1024^3: the voxel cache (8.6 GB) and the packed records (4 GiB + 1 GiB) need 64-bit byte offsets;
    the reference itself cannot run here (its int index overflows above ~812^3, utility.cl:21, SURVEY
    fact 9).  One pass against the oracle on the GPU-built SDF, plus the SDF's own properties."""
    n = 1024
    vol = scene.phantom(n)
    env = scene.env_map(1024, 512)
    tf = scene.tf_default_source()
    w, h = 640, 360
    g = GpuScene(gpu_ctx, vol, None, env, tf, (w, h))
    n_layers = gpu_ctx.sdf_build(g.volume, tf, g.sdf)
    sdf = g.sdf.pull()
    assert n_layers % 2 == 1 and np.abs(sdf.astype(np.int16)).max() == 127
    assert np.array_equal(sdf < 0, (vol >= 500) & (vol <= 1200))
    pos, d = scene.default_camera(n)
    o = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h), threads=min(16, len(os.sched_getaffinity(0))))
    for s in scene.glibc_rand(2):
        g.render(pos, d, s)
        o.render(pos, d, s)
    hits = o.hit_index[o.hit_index >= 0]
    assert hits.size > 20000
    assert (hits * 8 > 2 ** 32).any()          # byte offsets beyond 4 GiB are really exercised
    assert np.array_equal(g.hit_index.pull(), o.hit_index)
    assert np.array_equal(g.contrib.pull(), o.contrib)
    assert np.array_equal(g.cache.pull(), o.cache)
    g.release()


def test_4k_frame_pass_matches_the_oracle(full, orc):
    """config 5's frame size (3840x2160) on one GPU: one voxel-cache pass and one image-space pass of the
    same seed against the oracle; the two ranks of a 2-way tile split together give the same accumulation."""
    w, h = 3840, 2160
    vol, env, tf, pos, d = full["vol"], full["env"], full["tf"], full["pos"], full["d"]
    ctx = full["g"].ctx
    g = GpuScene(ctx, vol, full["sdf"], env, tf, (w, h), world=1)
    threads = min(16, len(os.sched_getaffinity(0)))
    o = orc.Scene(vol, full["sdf"], env, orc.parse_tf(tf), (w, h), threads=threads)
    seed = scene.glibc_rand(1)[0]
    g.render(pos, d, seed, debug=True, write_frame=False)
    o.render(pos, d, seed)
    assert np.array_equal(g.hit_index.pull(), o.hit_index)
    assert np.array_equal(g.contrib.pull(), o.contrib)
    assert np.array_equal(g.cache.pull(), o.cache)
    g.release()
    g = GpuScene(ctx, vol, full["sdf"], env, tf, (w, h), world=2)
    merged = np.zeros((h, w, 4), np.float32)
    for r in range(2):
        g.render(pos, d, seed, mode=ffi.ACCUM_IMAGE_SPACE, rank=r, debug=False, write_frame=False)
        merged += g.accum_row_major(r)
    hit = o.hit_index.reshape(h, w) >= 0
    assert np.array_equal(merged[hit][:, :3], o.contrib.reshape(h, w, 4)[hit][:, :3].astype(np.float32))
    assert np.all(merged[hit][:, 3] == 1) and not merged[~hit].any()
    g.release()
