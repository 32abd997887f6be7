"""N>1 path on CPU: two gloo ranks shard one frame into interleaved 8x8 tiles, each renders its
tiles (with the oracle standing in for the device kernels), the tile-major float4 buffers are
all-gathered by the same function bench.py uses, and the unpacked frame equals the single-rank one."""
import os
import socket

import numpy as np
import pytest

from cl_volume_renderer_amd import scene, tiles

W, H, WORLD = 96, 64, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(orc_ffi):
    vol = scene.phantom(32)
    tf = orc_ffi.parse_tf(scene.tf_default_source())
    sdf, _, _ = orc_ffi.sdf_build(vol, tf)
    env = scene.env_map(128, 64)
    pos = np.array([-14, 28, -14], np.float32)
    d = np.array([16, 16, 16], np.float32) - pos
    return vol, sdf, env, tf, pos, (d / np.linalg.norm(d)).astype(np.float32)


def _worker(rank, port, out_path):
    import torch
    import torch.distributed as dist

    from oracle import orc_ffi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    vol, sdf, env, tf, pos, d = _scene(orc_ffi)
    sc = orc_ffi.Scene(vol, sdf, env, tf, (W, H), mode=orc_ffi.MODE_IMAGE_SPACE, tile_rank=rank, tile_world=WORLD)
    for s in scene.glibc_rand(3):
        sc.render(pos, d, s)
    mine = torch.from_numpy(tiles.pack_tile_major(sc.accum, rank, WORLD).reshape(-1).copy())
    assert mine.numel() == tiles.accum_len(W, H, WORLD) * 4
    everyone = torch.zeros(mine.numel() * WORLD, dtype=torch.float32)
    tiles.gather_accum(mine, everyone, WORLD)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the max-over-ranks reduction bench.py applies to its timing
    assert t.item() == WORLD
    if rank == 0:
        np.save(out_path, tiles.unpack_all_ranks(everyone.numpy(), WORLD, W, H))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_partition_and_gather(orc, tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
    gathered = np.load(out)

    vol, sdf, env, tf, pos, d = _scene(orc)
    one = orc.Scene(vol, sdf, env, tf, (W, H), mode=orc.MODE_IMAGE_SPACE)
    for s in scene.glibc_rand(3):
        one.render(pos, d, s)
    assert one.accum[..., 3].max() == 3
    assert np.array_equal(gathered, one.accum)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_tile_major_round_trip(world):
    rng = np.random.default_rng(world)
    w, h = 128, 72
    full = rng.random((h, w, 4), dtype=np.float32)
    parts = [tiles.pack_tile_major(full, r, world) for r in range(world)]
    assert all(p.shape[0] == tiles.accum_len(w, h, world) for p in parts)
    assert np.array_equal(tiles.unpack_all_ranks(np.concatenate(parts), world, w, h), full)


def test_accum_len_matches_the_c_abi():
    from cl_volume_renderer_amd import ffi

    for (w, h, world) in [(1920, 1080, 1), (1920, 1080, 8), (3840, 2160, 8), (64, 64, 3), (96, 64, 2)]:
        assert tiles.accum_len(w, h, world) == ffi.accum_len(w, h, world)


def _voxel_worker(rank, port, out_path):
    import torch
    import torch.distributed as dist

    from oracle import orc_ffi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    vol, sdf, env, tf, pos, d = _scene(orc_ffi)
    sc = orc_ffi.Scene(vol, sdf, env, tf, (W, H), mode=orc_ffi.MODE_VOXEL_CACHE, tile_rank=rank, tile_world=WORLD)
    for s in scene.glibc_rand(3):
        sc.render(pos, d, s)
    words = torch.from_numpy(sc.cache.view(np.int32).copy())
    tiles.reduce_voxel_caches(words, WORLD)
    if rank == 0:
        np.save(out_path, words.numpy().view(np.uint16))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_voxel_cache_sum_equals_single_rank(orc, tmp_path):
    """the reference-exact world-space mode across ranks: private caches, one all-reduce(SUM) of the packed
    words; equal to the single-rank cache while no voxel reaches the 256-token cap"""
    import torch.multiprocessing as mp

    out = str(tmp_path / "cache.npy")
    mp.spawn(_voxel_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
    summed = np.load(out)
    vol, sdf, env, tf, pos, d = _scene(orc)
    one = orc.Scene(vol, sdf, env, tf, (W, H))
    for s in scene.glibc_rand(3):
        one.render(pos, d, s)
    assert 0 < one.cache.reshape(-1, 4)[:, 3].max() < 256
    assert np.array_equal(summed, one.cache)


def _over_cap_worker(rank, port, out_path):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    # four entries {r, g, b, count}; entry 1 exceeds the cap only globally, entry 2 sits exactly on it
    e = np.zeros((4, 4), np.uint16)
    e[0] = (100, 200, 300, 3) if rank == 0 else (7, 8, 9, 1)
    e[1] = (200 * 255, 200 * 254, 200 * 1, 200) if rank == 0 else (256 * 255, 256 * 3, 256 * 250, 256)
    e[2] = (128 * 10, 128 * 20, 128 * 30, 128)
    e[3] = (0, 0, 0, 0)
    words = torch.from_numpy(e.reshape(-1).view(np.int32).copy())
    tiles.reduce_voxel_caches(words, WORLD, chunk_words=4)  # two chunks: the chunking is exercised too
    if rank == 0:
        np.save(out_path, words.numpy().view(np.uint16).reshape(4, 4))
    dist.barrier()
    dist.destroy_process_group()


def test_voxel_cache_reduce_never_carries_between_lanes(tmp_path):
    """a raw int32 all-reduce of the packed {r,g,b,count} words lets r carry into g and b into count
    once the GLOBAL count of a voxel passes 256.  The lanes are widened before the sum; entries over the cap are
    rescaled to 256 samples."""
    import torch.multiprocessing as mp

    out = str(tmp_path / "over.npy")
    mp.spawn(_over_cap_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
    got = np.load(out).astype(np.int64)
    assert got[0].tolist() == [107, 208, 309, 4]                      # below the cap: the exact sum
    total = np.array([200 * 255 + 256 * 255, 200 * 254 + 256 * 3, 200 * 1 + 256 * 250, 456], np.int64)
    assert got[1].tolist() == [total[0] * 256 // 456, total[1] * 256 // 456, total[2] * 256 // 456, 256]
    assert got[1, :3].max() <= 255 * 256                              # fits the u16 lanes again
    assert got[2].tolist() == [256 * 10, 256 * 20, 256 * 30, 256]     # exactly on the cap: still the exact sum
    assert got[3].tolist() == [0, 0, 0, 0]


CAP_W, CAP_H, CAP_PASSES = 384, 384, 8


def _cap_scene(orc_ffi):
    vol = scene.phantom(24)
    tf = orc_ffi.parse_tf(scene.tf_default_source())
    sdf, _, _ = orc_ffi.sdf_build(vol, tf)
    env = scene.env_map(128, 64)
    pos = np.array([-9.0, 20.0, -9.0], np.float32)
    d = np.array([12.0, 12.0, 12.0], np.float32) - pos
    return vol, sdf, env, tf, pos, (d / np.linalg.norm(d)).astype(np.float32)


def _exchange_worker(rank, port, out_dir):
    import torch
    import torch.distributed as dist

    from oracle import orc_ffi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    vol, sdf, env, tf, pos, d = _cap_scene(orc_ffi)
    sc = orc_ffi.Scene(vol, sdf, env, tf, (CAP_W, CAP_H), mode=orc_ffi.MODE_IMAGE_SPACE, tile_rank=rank, tile_world=WORLD)
    n_entries = orc_ffi.cache_len(24, 24, 24) // 4
    words = torch.zeros(n_entries * 2, dtype=torch.int32)
    x = tiles.VoxelExchange(words, WORLD)
    idx = None
    for i, s in enumerate(scene.glibc_rand(CAP_PASSES)):
        sc.render(pos, d, s)
        if i == 0:
            hit = torch.from_numpy(sc.hit_index)
            idx = torch.nonzero((hit >= 0) & (hit < n_entries)).flatten()
            x.set_camera(hit[idx])
        x.add_pass(torch.from_numpy(sc.contrib.astype(np.int32))[idx, :3])
    np.save(os.path.join(out_dir, "replica%d.npy" % rank), words.numpy().view(np.uint16))
    dist.barrier()
    dist.destroy_process_group()


def test_global_token_rule_across_two_ranks(orc, tmp_path):
    """the reference's 256-token cap applied to the GLOBAL count of a voxel, pass by pass
    (tiles.VoxelExchange).  On a scene that reaches the cap: the replicas of both ranks are identical; no voxel
    exceeds 256 samples; the sample count of every voxel equals the single-rank cache's; every voxel below the cap
    equals the single-rank cache entry bit for bit; capped voxels hold 256 contributions of <= 255."""
    import torch.multiprocessing as mp

    mp.spawn(_exchange_worker, args=(_free_port(), str(tmp_path)), nprocs=WORLD, join=True)
    r0 = np.load(str(tmp_path / "replica0.npy")).reshape(-1, 4)
    r1 = np.load(str(tmp_path / "replica1.npy")).reshape(-1, 4)
    assert np.array_equal(r0, r1)
    vol, sdf, env, tf, pos, d = _cap_scene(orc)
    one = orc.Scene(vol, sdf, env, tf, (CAP_W, CAP_H))
    for s in scene.glibc_rand(CAP_PASSES):
        one.render(pos, d, s)
    want = one.cache.reshape(-1, 4)[: r0.shape[0]]
    assert want[:, 3].max() == 256 and (want[:, 3] == 256).sum() > 20, "the scene must reach the cap"
    assert (want[:, 3] < 256).sum() > 20
    assert r0[:, 3].max() == 256
    assert np.array_equal(r0[:, 3], want[:, 3])
    below = want[:, 3] < 256
    assert np.array_equal(r0[below], want[below])
    assert (r0[~below, :3].astype(np.int64) <= 255 * 256).all()
