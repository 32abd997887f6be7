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
    """The path's only collective: every rank contributes its tile-major buffer, every rank receives all of them back
    to back.  `accum`, `accum_all`: 1-D torch tensors on the same device -- the float4 sums (16 B per pixel; then
    `clwh_accum_resolve`), or, as bench.py does it, the rank's tiles already resolved to RGBA8 by
    `clwh_accum_resolve_tiles` (4 B per pixel; then `clwh_frame_from_tiles`)."""
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


def reduce_voxel_caches(cache_words, world: int, chunk_words: int = 1 << 25):
    """End-of-job sum of the ranks' private world-space caches (second row).  A cache entry is four
    u16 lanes {r, g, b, count} in two 32-bit words (utility.cl:39-54).  The lanes are widened to int32 before the
    all-reduce(SUM), so nothing can carry from one lane into its neighbour whatever the global count is, and
    repacked afterwards:
      * global count <= 256 (the reference's token cap, ray_marching.cl:28,39): the entry is the exact lane-wise
        sum -- identical to the single-GPU cache, which is order-independent in that regime;
      * global count > 256 (every rank capped only its own pixels): the sums are rescaled to 256 samples
        (lane * 256 // count, count = 256), i.e. the entry keeps the mean of ALL contributions and fits its u16
        lanes again.  The reference itself is order-dependent beyond the cap; the mode that applies the token rule
        to the global count pass by pass is `exchange_pass_contributions` below.
    `cache_words`: 1-D int32 torch tensor viewing the rank's cache; reduced in place on every rank, chunk by chunk
    (the widened copy of a chunk is 4 x chunk_words bytes x 2)."""
    if world == 1:
        return cache_words
    import torch
    import torch.distributed as dist

    direct = dist.get_backend() == "nccl" or not cache_words.is_cuda
    n = cache_words.numel()
    assert n % 2 == 0
    chunk_words -= chunk_words % 2
    for i in range(0, n, chunk_words):
        w = cache_words[i:i + chunk_words]
        lanes = torch.stack([w & 0xFFFF, (w >> 16) & 0xFFFF])  # [2][words] int32: low lanes (r, b), high lanes (g, count)
        if direct:
            dist.all_reduce(lanes, op=dist.ReduceOp.SUM)
        else:  # rehearsal without RCCL (gloo): stage through the host
            host = lanes.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            lanes.copy_(host)
        lo, hi = lanes[0], lanes[1]
        count = hi[1::2]
        over = count > 256
        if bool(over.any()):
            c = torch.where(over, count, torch.ones_like(count))
            lo[0::2] = torch.where(over, lo[0::2] * 256 // c, lo[0::2])  # r   (sums stay below 2^31: 255 x 256 x world)
            hi[0::2] = torch.where(over, hi[0::2] * 256 // c, hi[0::2])  # g
            lo[1::2] = torch.where(over, lo[1::2] * 256 // c, lo[1::2])  # b
            hi[1::2] = torch.clamp(count, max=256)
        w.copy_(lo | (hi << 16))
    return cache_words


class VoxelExchange:
    """The reference's world-space accumulation across ranks WITH its 256-token rule applied to the GLOBAL count
    (utility.cl:20-31, ray_marching.cl:28,39), pass by pass.

    Every rank keeps an identical replica of the packed voxel cache.  Per camera, the ranks all-gather the cache
    entry of each of their hit pixels once (`set_camera`); per pass they all-gather only the pixels' contributions
    (3 x int32 per hit pixel, `add_pass`) and every rank applies ALL of them to its replica with the same
    deterministic rule: the contributions to one voxel are taken in (rank, pixel) order while the voxel's count is
    below 256 and dropped afterwards.  That is one legal outcome of the reference's race (which of the competing
    work-items gets the last tokens is unspecified there), identical on every rank, identical to the single-GPU
    cache while the global count stays below the cap, and never above the cap.

    On the GPU the grouping and the capped add are the library's (`clwh_cache_exchange_plan`, one stable sort per camera;
    `clwh_cache_apply_contributions`, ONE kernel per pass -- the entry points a C/C++ caller uses as well); this class
    only adds the two all-gathers.  `ctx`: the ffi.Context whose stream the collectives' results are consumed on; a
    device cache without one is refused.  The torch restatement of the same rule below is for CPU tensors over gloo
    (tests/test_multi_rank_cpu.py), where no GPU exists."""

    CAP = 256

    def __init__(self, cache_words, world: int, ctx=None):
        import torch

        self.torch = torch
        self.words = cache_words        # 1-D int32 tensor, 2 words per entry: r | g << 16, b | count << 16
        self.world = world
        self.dev = cache_words.device
        self.ctx = ctx
        self.plan = None
        if cache_words.is_cuda:
            if ctx is None:
                raise RuntimeError("VoxelExchange on device memory needs the ffi.Context of the HIP library (no torch fallback on the GPU)")
            self.m_cache = ctx.wrap(cache_words.data_ptr(), cache_words.numel() * 4)

    def _all_gather_padded(self, t, n_max):
        """ranks contribute tensors whose first dimension differs: pad to n_max rows, gather, return the list"""
        torch = self.torch
        if self.world == 1:
            return [t]
        import torch.distributed as dist

        pad = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        out = torch.empty((self.world * n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if dist.get_backend() == "nccl" or not t.is_cuda:
            dist.all_gather_into_tensor(out, pad)
        else:  # rehearsal without RCCL (gloo): stage through the host
            host = out.cpu()
            dist.all_gather_into_tensor(host, pad.cpu())
            out.copy_(host)
        return [out[r * n_max:(r + 1) * n_max] for r in range(self.world)]

    def set_camera(self, own_entries):
        """own_entries: int64 tensor, the cache entry of each of this rank's hit pixels in pixel order (entries
        outside the cache already removed).  One all-gather per camera."""
        torch = self.torch
        n_own = torch.tensor([own_entries.numel()], dtype=torch.int64, device=self.dev)
        if self.world > 1:
            import torch.distributed as dist

            sizes = [torch.zeros_like(n_own) for _ in range(self.world)]
            if dist.get_backend() == "nccl" or not n_own.is_cuda:
                dist.all_gather(sizes, n_own)
            else:
                host = [s.cpu() for s in sizes]
                dist.all_gather(host, n_own.cpu())
                sizes = [h.to(self.dev) for h in host]
            self.sizes = [int(s.item()) for s in sizes]
        else:
            self.sizes = [int(n_own.item())]
        self.n_max = max(max(self.sizes), 1)
        parts = self._all_gather_padded(own_entries, self.n_max)
        entries = torch.cat([p[:n] for p, n in zip(parts, self.sizes)])     # (rank, pixel) order
        if self.ctx is not None:
            if self.plan is not None:
                self.plan.release()
            self.plan = None
            self.entries = entries.contiguous()
            if self.entries.numel() == 0:   # no rank hit anything: nothing to exchange for this camera
                return
            torch.cuda.current_stream().synchronize()  # the plan is built on the context's stream
            m_entries = self.ctx.wrap(self.entries.data_ptr(), self.entries.numel() * 8)
            self.plan = self.ctx.exchange_plan(m_entries, self.entries.numel())   # blocking: the list is consumed here
            m_entries.release()
            return
        # group the contributions by voxel, keeping (rank, pixel) order inside a group
        order = torch.argsort(entries, stable=True)
        sorted_e = entries[order]
        self.unique, self.inverse_sorted, counts = torch.unique_consecutive(sorted_e, return_inverse=True, return_counts=True)
        starts = torch.cumsum(counts, 0) - counts
        self.pos_in_group = torch.arange(sorted_e.numel(), device=self.dev) - starts[self.inverse_sorted]
        self.order = order

    def add_pass(self, own_rgb):
        """own_rgb: int32 [n_own][3], this pass's contribution of each of this rank's hit pixels (same order as the
        entries given to set_camera).  One all-gather, then the capped scatter-add into the replica."""
        torch = self.torch
        parts = self._all_gather_padded(own_rgb, self.n_max)
        if self.ctx is not None:
            if self.plan is None:
                return
            rgb = parts[0][: self.sizes[0]] if self.world == 1 else torch.cat([p[:n] for p, n in zip(parts, self.sizes)])
            self.rgb = rgb.to(torch.int32).contiguous()   # kept alive until the next pass: the kernel reads it asynchronously
            m_rgb = self.ctx.wrap(self.rgb.data_ptr(), self.rgb.numel() * 4)
            self.plan.apply(self.m_cache, m_rgb, int(self.rgb.shape[1]))
            m_rgb.release()
            return
        rgb = torch.cat([p[:n] for p, n in zip(parts, self.sizes)])[self.order].to(torch.int64)
        u = self.unique
        w0 = self.words[2 * u].to(torch.int64) & 0xFFFFFFFF
        w1 = self.words[2 * u + 1].to(torch.int64) & 0xFFFFFFFF
        count = w1 >> 16
        remaining = torch.clamp(self.CAP - count, min=0)
        keep = (self.pos_in_group < remaining[self.inverse_sorted]).to(torch.int64)
        add = torch.zeros((u.numel(), 4), dtype=torch.int64, device=self.dev)
        add.index_add_(0, self.inverse_sorted, torch.cat([rgb * keep[:, None], keep[:, None]], dim=1))
        r = (w0 & 0xFFFF) + add[:, 0]
        g = (w0 >> 16) + add[:, 1]
        b = (w1 & 0xFFFF) + add[:, 2]
        c = count + add[:, 3]
        # <= 256 contributions of <= 255 each: every lane stays below 2^16
        n0 = (r | (g << 16))
        n1 = (b | (c << 16))
        self.words[2 * u] = torch.where(n0 >= 2 ** 31, n0 - 2 ** 32, n0).to(torch.int32)
        self.words[2 * u + 1] = torch.where(n1 >= 2 ** 31, n1 - 2 ** 32, n1).to(torch.int32)
