"""Image-tile partition of a frame over ranks and the one collective of the path.

8x8 pixel tiles, owner = (tx + ty) % world; a rank numbers its tiles
``slot = ty * tiles_per_row + tx // world`` and stores 64 float4 per slot (tile-major), which is the
layout the HIP kernels write (csrc/render_device.hpp ``tile_from_slot``) and ``clwh_accum_resolve`` reads.
The gather works on any torch device/backend (RCCL on the GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def tiles_per_row(width: int, world: int) -> int:
    return (width // 8 + world - 1) // world


def accum_len(width: int, height: int, world: int) -> int:
    """float4 elements of one rank's buffer (== clwh_accum_len)."""
    return (height // 8) * tiles_per_row(width, world) * 64


def owner_of_tile(tx: int, ty: int, world: int) -> int:
    return (tx + ty) % world


def pack_tile_major(row_major: np.ndarray, rank: int, world: int) -> np.ndarray:
    """[h][w][4] -> the rank's tile-major buffer [slots*64][4] (tiles of other ranks are left out)."""
    h, w, _ = row_major.shape
    per_row = tiles_per_row(w, world)
    out = np.zeros(((h // 8) * per_row * 64, 4), dtype=row_major.dtype)
    for ty in range(h // 8):
        for tx in range(w // 8):
            if owner_of_tile(tx, ty, world) != rank:
                continue
            slot = ty * per_row + tx // world
            out[slot * 64:(slot + 1) * 64] = row_major[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8].reshape(64, 4)
    return out


def unpack_all_ranks(accum_all: np.ndarray, world: int, width: int, height: int) -> np.ndarray:
    """all ranks' buffers back to back (what the all-gather produces) -> [h][w][4]."""
    per_row = tiles_per_row(width, world)
    per_rank = (height // 8) * per_row * 64
    a = accum_all.reshape(world, per_rank, 4)
    out = np.zeros((height, width, 4), dtype=accum_all.dtype)
    for ty in range(height // 8):
        for tx in range(width // 8):
            r = owner_of_tile(tx, ty, world)
            slot = ty * per_row + tx // world
            out[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8] = a[r, slot * 64:(slot + 1) * 64].reshape(8, 8, 4)
    return out


def gather_accum(accum, accum_all, world: int):
    """The path's only collective: every rank contributes its tile-major float4 buffer, every rank
    receives all of them back to back.  `accum`, `accum_all`: 1-D torch tensors on the same device."""
    if world == 1:
        accum_all.copy_(accum)
        return accum_all
    import torch.distributed as dist

    if dist.get_backend() == "nccl" or not accum.is_cuda:
        dist.all_gather_into_tensor(accum_all, accum)
    else:
        # rehearsal without RCCL (gloo): stage through the host
        host_all = accum_all.cpu()
        dist.all_gather_into_tensor(host_all, accum.cpu())
        accum_all.copy_(host_all)
    return accum_all


def reduce_voxel_caches(cache_words, world: int):
    """Reference-exact voxel-cache mode across ranks (second row): every rank keeps a private
    world-space cache for its image tiles; the caches are summed once per job.  A cache entry is four u16
    lanes {r, g, b, count} seen as two 32-bit words, so an all-reduce(SUM) of the words adds lane-wise as long
    as no lane carries -- which holds exactly while the GLOBAL count of a voxel stays <= 256 (lanes <= 255 x
    count <= 65280), the same condition under which the single-GPU cache is order-independent (SURVEY facts
    3, 4).  Beyond the cap every rank has applied the reference's token rule to its own pixels only.
    `cache_words`: 1-D int32 torch tensor viewing the rank's cache; reduced in place on every rank."""
    if world == 1:
        return cache_words
    import torch.distributed as dist

    if dist.get_backend() == "nccl" or not cache_words.is_cuda:
        dist.all_reduce(cache_words, op=dist.ReduceOp.SUM)
    else:  # rehearsal without RCCL (gloo): stage through the host
        host = cache_words.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        cache_words.copy_(host)
    return cache_words
