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


def gradient_length_f32(vol: np.ndarray) -> np.ndarray:
    v = vol.astype(np.int32)
    gx = (_shift(v, 2, 1) - _shift(v, 2, -1)).astype(np.float32)
    gy = (_shift(v, 1, 1) - _shift(v, 1, -1)).astype(np.float32)
    gz = (_shift(v, 0, 1) - _shift(v, 0, -1)).astype(np.float32)
    return np.sqrt((gx * gx + gy * gy) + gz * gz, dtype=np.float32)


def _round_half_away(x: np.ndarray) -> np.ndarray:
    return np.where(x >= 0, np.floor(x + np.float32(0.5)), np.ceil(x - np.float32(0.5))).astype(np.float32)


def tf_sort_values(vol, width, height, min_v, max_v, min_g, max_g) -> np.ndarray:
    """opencl_kernels/histogram.cl:4-32; bins outside the frame are dropped (the reference writes out of
    bounds for them).  Returns frame[width*height] indexed x*height + y."""
    f32 = np.float32
    g = gradient_length_f32(vol)
    v = vol.astype(np.float32)
    keep = ~(g > f32(max_g)) & ~(v > f32(max_v))
    px = _round_half_away(((v - f32(min_v)) / (f32(max_v) - f32(min_v))) * f32(width)).astype(np.int64)
    py = _round_half_away(((g - f32(min_g)) / (f32(max_g) - f32(min_g))) * f32(height)).astype(np.int64)
    keep &= (px >= 0) & (px < width) & (py >= 0) & (py < height)
    frame = np.zeros(width * height, dtype=np.uint32)
    np.add.at(frame, (px[keep] * height + py[keep]).ravel(), 1)
    return frame


def render_tf(vol, width, height, min_v, max_v, min_g, max_g) -> np.ndarray:
    """app/renderer.cpp:45-124 end to end: RGBA8 [height][width][4]."""
    import math

    frame = tf_sort_values(vol, width, height, min_v, max_v, min_g, max_g).astype(np.int64)
    distinct = set()
    for i in np.nonzero(frame)[0]:
        value = int(frame[i])
        unit = max(int(math.pow(10, math.floor(math.log10(value)) - 1)), 1)
        corrected = int(math.floor(value // unit) * unit)
        frame[i] = corrected
        distinct.add(corrected)
    lookup = sorted(distinct)
    rank = {v: i for i, v in enumerate(lookup)}
    out = np.zeros((height, width, 4), dtype=np.uint8)
    out[..., 3] = 255
    if not lookup:
        return out * 0  # the kernel is not launched: the image keeps its zeros
    fr = frame.reshape(width, height)
    for y in range(height):
        col = fr[:, height - y - 1]
        for x in np.nonzero(col)[0]:
            r = np.float32(20.0) + (np.float32(rank[int(col[x])]) / np.float32(len(lookup))) * np.float32(235.0)
            out[y, x, :3] = min(max(int(r), 0), 255)
    return out
