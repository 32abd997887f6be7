"""Where do the march steps of the distribution rays (the bounce phase = what k_bounce executes) fetch their step bytes?
Instrumented CPU oracle on the benchmark scene: python tools/step_locality.py [N] [--sdf-cache FILE.npy] [--passes P]

Answers three design questions with exact counts:
  * a per-lane line cache: how many fetches stay in the 64-byte line (4^3 sub-brick) / 8^3 brick of the ray's previous fetch;
  * LDS staging of the neighbourhood of a wave's surface patch (north_star): how many fetches fall within R voxels of the
    sample's primary hit;
  * an L2-resident table of 'uniform' 4^3 sub-bricks (all 64 step bytes equal): how many fetches such a table would serve.
Test / analysis tooling: uses the oracle, never the product library."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cl_volume_renderer_amd import scene  # noqa: E402
from oracle import orc_ffi  # noqa: E402

args = sys.argv[1:]
n = int(args[0]) if args and args[0].isdigit() else 256
cache = args[args.index("--sdf-cache") + 1] if "--sdf-cache" in args else None
passes = int(args[args.index("--passes") + 1]) if "--passes" in args else 4
w, h = (1920, 1080) if n >= 512 else (960, 544)
vol = scene.phantom(n)
tf = orc_ffi.parse_tf(scene.tf_default_source())
if cache and os.path.exists(cache):
    sdf = np.load(cache)
else:
    sdf, _, _ = orc_ffi.sdf_build(vol, tf)
    if cache:
        np.save(cache, sdf)
env = scene.env_map(4096, 2048)
pos, d = scene.default_camera(n)
# the march's step byte: (event class != 0) << 7 | max(sdf, 0)  (csrc/packed_volume.hpp)
stepb = (np.maximum(sdf, 0).astype(np.uint8) | (((vol >= 500) & (vol <= 1200)).astype(np.uint8) << 7))
s4 = stepb.reshape(n // 4, 4, n // 4, 4, n // 4, 4).transpose(0, 2, 4, 1, 3, 5).reshape(n // 4, n // 4, n // 4, 64)
uniform4 = np.ascontiguousarray((s4.min(axis=3) == s4.max(axis=3)).astype(np.uint8))
sc = orc_ffi.Scene(vol, sdf, env, tf, (w, h), mode=orc_ffi.MODE_IMAGE_SPACE, threads=len(os.sched_getaffinity(0)))
sc.locality = np.zeros(orc_ffi.LOCALITY_TOTAL, np.uint64)
sc.uniform4 = uniform4
for s in scene.glibc_rand(passes):
    sc.render(pos, d, s)
L = dict(zip(orc_ffi.LOCALITY_NAMES, (int(v) for v in sc.locality)))
c = sc.counter_dict()
items = c["n_hit"]
f = float(L["fetches"])
print("scene: phantom(%d), %dx%d, default camera, default TF, %d passes; %d (hit, seed) items" % (n, w, h, passes, items))
print("bounce-phase march steps per item: %.2f; step-byte fetches inside the volume per item: %.2f" % (L["steps"] / items, f / items))
print("step length (voxels):  <=1 %.1f %%   <=2 %.1f %%   <=8 %.1f %%   <=32 %.1f %%   >32 %.1f %%" % tuple(
    100.0 * x / L["steps"] for x in (L["step_le_1"], L["step_le_2"], L["step_le_8"], L["step_le_32"], L["steps"] - L["step_le_32"])))
print("fetch in the same 4^3 sub-brick (64-B line) as the ray's previous fetch: %.1f %%;  same 8^3 brick: %.1f %%" % (
    100 * L["same_sub4"] / f, 100 * L["same_brick8"] / f))
print("fetch within R voxels (Chebyshev) of the sample's primary hit:  R=8 %.1f %%   R=16 %.1f %%   R=32 %.1f %%   R=64 %.1f %%" % tuple(
    100 * L[k] / f for k in ("near_8", "near_16", "near_32", "near_64")))
print("fetch in a 4^3 sub-brick whose 64 step bytes are all equal: %.1f %%   (%.1f %% of the volume's sub-bricks are uniform)" % (
    100 * L["uniform4"] / f, 100.0 * uniform4.mean()))


def percentiles(hist, qs):
    cum = np.cumsum(hist)
    return [int(np.searchsorted(cum, q * cum[-1])) for q in qs]


ray = sc.locality[orc_ffi.LOCALITY_HIST_RAY:orc_ffi.LOCALITY_HIST_RAY + 256].astype(np.int64)
item = sc.locality[orc_ffi.LOCALITY_HIST_ITEM:orc_ffi.LOCALITY_HIST_ITEM + 512].astype(np.int64)
qs = (0.5, 0.9, 0.99, 0.999, 0.9999)
print("march steps per distribution ray: mean %.1f, percentiles 50 / 90 / 99 / 99.9 / 99.99: %s, max %d" % (
    (ray * np.arange(256)).sum() / ray.sum(), " / ".join(map(str, percentiles(ray, qs))), int(np.nonzero(ray)[0].max())))
print("march steps per sample (two rays, one lane walks them back to back): mean %.1f, percentiles: %s, max %d" % (
    (item * np.arange(512)).sum() / item.sum(), " / ".join(map(str, percentiles(item, qs))), int(np.nonzero(item)[0].max())))
# a wave walks 64 samples in lock-step phases: its step phases last as long as its longest sample
rng = np.random.default_rng(0)
p_item = item / item.sum()
draw = rng.choice(512, size=(20000, 64), p=p_item)
p_ray = ray / ray.sum()
draw_ray = rng.choice(256, size=(20000, 64), p=p_ray)
print("expected longest of 64 independent samples: %.0f steps; of 64 independent single rays: %.0f steps" % (
    draw.max(axis=1).mean(), draw_ray.max(axis=1).mean()))
