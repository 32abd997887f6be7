"""numpy restatement of the volume pre-processing kernels next to the hot path.
TEST INFRASTRUCTURE ONLY (see oracle/orc.h).  PARITY UNPINNED: the reference holds no fixture for them.

fetch_stats  opencl_kernels/reference_volume_figures.cl:10-26 (+ utility_filter.cl:2-35)
apply_clip   opencl_kernels/reference_volume_clip.cl:4-15
"""
import numpy as np


def _shift(vol, axis, by):
    """vol sampled at index + by along `axis`, border texel 0."""
    out = np.zeros_like(vol)
    src = [slice(None)] * 3
    dst = [slice(None)] * 3
    n = vol.shape[axis]
    if by > 0:
        src[axis], dst[axis] = slice(by, n), slice(0, n - by)
    else:
        src[axis], dst[axis] = slice(0, n + by), slice(-by, n)
    out[tuple(dst)] = vol[tuple(src)]
    return out


def gradient_length_int(vol: np.ndarray) -> np.ndarray:
    """(int) length(gradient_prewitt_nn) per voxel: central differences without the 1/2 factor, float32
    arithmetic in the kernel's order, truncated like the implicit float -> int of atomic_min/max."""
    v = vol.astype(np.int32)
    gx = (_shift(v, 2, 1) - _shift(v, 2, -1)).astype(np.float32)
    gy = (_shift(v, 1, 1) - _shift(v, 1, -1)).astype(np.float32)
    gz = (_shift(v, 0, 1) - _shift(v, 0, -1)).astype(np.float32)
    s = (gx * gx + gy * gy) + gz * gz
    return np.sqrt(s, dtype=np.float32).astype(np.int32)


def fetch_stats(vol: np.ndarray, init=(2**31 - 1, -2**31, 2**31 - 1, -2**31, -2**31)) -> np.ndarray:
    """stats[5] after the kernel, starting from the host's initial values (app/reference_volume.cpp:23-28)."""
    g = gradient_length_int(vol)
    out = np.array(init, dtype=np.int64)
    out[0] = min(out[0], int(vol.min()))
    out[1] = max(out[1], int(vol.max()))
    out[2] = min(out[2], int(g.min()))
    out[3] = max(out[3], int(g.max()))
    return out.astype(np.int32)


def apply_clip(vol: np.ndarray, start, length) -> np.ndarray:
    """clipped[z][y][x] = original[start + (x,y,z)], border 0 outside the original."""
    Z, Y, X = vol.shape
    out = np.zeros((length[2], length[1], length[0]), dtype=vol.dtype)
    x1, y1, z1 = [min(s + l, d) for s, l, d in zip(start, length, (X, Y, Z))]
    if x1 > start[0] and y1 > start[1] and z1 > start[2]:
        out[: z1 - start[2], : y1 - start[1], : x1 - start[0]] = vol[start[2]:z1, start[1]:y1, start[0]:x1]
    return out
