"""-m gpu: clwh_cache_exchange_plan / clwh_cache_apply_contributions -- the reference's world-space accumulation
(utility.cl:20-54) with its 256-token rule (ray_marching.cl:28,39) applied to the GLOBAL count, behind the C ABI
(SURVEY 8e "reference-exact voxel-cache mode")."""
import numpy as np
import pytest

from cl_volume_renderer_amd import ffi, scene, tiles

pytestmark = pytest.mark.gpu


def _ordered_capped_add(n_entries, entries, passes):
    """the rule, restated serially: contributions in list order while the voxel's count is below 256"""
    cache = np.zeros((n_entries, 4), np.int64)
    for rgb in passes:
        for e, c in zip(entries.tolist(), rgb.tolist()):
            if 0 <= e < n_entries and cache[e, 3] < 256:
                cache[e, :3] += c[:3]
                cache[e, 3] += 1
    return cache


@pytest.mark.parametrize("stride", [3, 4])
def test_apply_contributions_equals_the_serial_rule(gpu_ctx, stride):
    """random lists with heavy sharing (some voxels get > 256 requests within a few passes) and entries outside the cache;
    bit-exact against the serial restatement, whatever the kernel's parallel order"""
    rng = np.random.default_rng(5 + stride)
    n_entries = 5000
    n = 40000
    hot = rng.integers(0, n_entries, 40)
    entries = np.where(rng.random(n) < 0.5, hot[rng.integers(0, 40, n)], rng.integers(0, n_entries, n)).astype(np.int64)
    entries[rng.integers(0, n, 50)] = -2          # hits outside the allocation (k_primary reports them so)
    entries[rng.integers(0, n, 50)] = n_entries + 7
    cache = gpu_ctx.buffer(n_entries * 8, np.uint16, (n_entries, 4))
    gpu_ctx.buffer_reset(cache)
    m_entries = gpu_ctx.buffer_from(entries)
    plan = gpu_ctx.exchange_plan(m_entries, n)
    passes = []
    for _ in range(6):
        rgb = rng.integers(0, 256, (n, stride)).astype(np.int32)
        passes.append(rgb)
        m_rgb = gpu_ctx.buffer_from(rgb)
        plan.apply(cache, m_rgb, stride)
        gpu_ctx.finish()
        m_rgb.release()
    want = _ordered_capped_add(n_entries, entries, passes)
    got = cache.pull().astype(np.int64)
    assert want[:, 3].max() == 256 and (want[:, 3] == 256).sum() >= 30 and ((want[:, 3] > 0) & (want[:, 3] < 256)).sum() > 1000
    assert np.array_equal(got, want)
    plan.release()
    # an empty list is a valid plan
    empty = gpu_ctx.exchange_plan(m_entries, 0)
    empty.release()
    m_entries.release()
    cache.release()


def test_exchange_of_two_emulated_ranks_against_the_oracle(gpu_ctx, orc):
    """The whole exchange as a C caller would drive it, both ranks of a 2-way tile split emulated on one GPU: each rank renders
    its tiles in image-space scratch with per-pixel contribution output, the lists are concatenated in (rank, pixel) order, and
    the library applies them to the cache.  On a scene that reaches the cap: counts equal the single-rank oracle cache's
    everywhere, entries below the cap are equal bit for bit, and the torch restatement used by the CPU (gloo) tests agrees
    with the HIP kernel on every byte -- which ties tests/test_multi_rank_cpu.py to the product path."""
    import torch

    from tests.gpu_util import GpuScene

    vol = scene.phantom(24)
    tf = scene.tf_default_source()
    sdf, _, _ = orc.sdf_build(vol, orc.parse_tf(tf))
    env = scene.env_map(128, 64)
    pos = np.array([-9.0, 20.0, -9.0], np.float32)
    d = np.array([12.0, 12.0, 12.0], np.float32) - pos
    d = (d / np.linalg.norm(d)).astype(np.float32)
    w = h = 384
    world = 2
    g = GpuScene(gpu_ctx, vol, sdf, env, tf, (w, h), world=world)
    n_entries = ffi.cache_len(24, 24, 24) // 4
    seeds = scene.glibc_rand(8)
    words = torch.zeros(n_entries * 2, dtype=torch.int32, device="cuda")
    m_cache = gpu_ctx.wrap(words.data_ptr(), n_entries * 8)
    words_cpu = torch.zeros(n_entries * 2, dtype=torch.int32)
    x_cpu = tiles.VoxelExchange(words_cpu, 1)
    plan = None
    idx = [None, None]
    for i, s in enumerate(seeds):
        rgb = []
        ent = []
        for r in range(world):
            g.hit_index.push(np.full(w * h, -1, np.int64))   # a rank writes only the pixels of its own tiles
            g.render(pos, d, s, mode=ffi.ACCUM_IMAGE_SPACE, rank=r, write_frame=False)
            if i == 0:
                hit = g.hit_index.pull()
                idx[r] = np.nonzero((hit >= 0) & (hit < n_entries))[0]
                ent.append(hit[idx[r]])
            rgb.append(g.contrib.pull()[idx[r], :3].astype(np.int32))
        if i == 0:
            entries = np.concatenate(ent)
            m_entries = gpu_ctx.buffer_from(entries)
            plan = gpu_ctx.exchange_plan(m_entries, entries.size)
            x_cpu.set_camera(torch.from_numpy(entries))
        allrgb = np.ascontiguousarray(np.concatenate(rgb))
        m_rgb = gpu_ctx.buffer_from(allrgb)
        plan.apply(m_cache, m_rgb, 3)
        gpu_ctx.finish()
        m_rgb.release()
        x_cpu.add_pass(torch.from_numpy(allrgb))
    got = words.cpu().numpy().view(np.uint16).reshape(-1, 4)
    assert np.array_equal(got, words_cpu.numpy().view(np.uint16).reshape(-1, 4))
    one = orc.Scene(vol, sdf, env, orc.parse_tf(tf), (w, h))
    for s in seeds:
        one.render(pos, d, s)
    want = one.cache.reshape(-1, 4)[:n_entries]
    assert want[:, 3].max() == 256 and (want[:, 3] == 256).sum() > 20 and (want[:, 3] < 256).sum() > 20
    assert np.array_equal(got[:, 3], want[:, 3])
    below = want[:, 3] < 256
    assert np.array_equal(got[below], want[below])
    assert (got[~below, :3].astype(np.int64) <= 255 * 256).all()
    # the frame of the replica: the library's resolve_only path
    g.kernel.render(frame=g.frame, volume=g.volume, sdf=g.sdf, env=g.env, buffer_volume=m_cache, cam_pos=pos, cam_dir=d, seed=0,
                    width=w, height=h, mode=ffi.ACCUM_VOXEL_CACHE, resolve_only=True)
    frame = g.frame.pull()
    assert frame[..., 3].max() == 200 and (frame[..., 3] == 1).sum() > 1000
    plan.release()
    m_entries.release()
    g.release()


def test_voxel_exchange_class_on_device_needs_the_library():
    import torch

    words = torch.zeros(64, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError):
        tiles.VoxelExchange(words, 1)
