"""How many of the distribution rays' step fetches could an 'exit certificate' avoid?  (candidate optimisation, measured on
the instrumented CPU oracle before anything is built):  python tools/exit_certificate.py [N] [--sdf-cache FILE.npy]

A march that ends in Exit_volume contributes through its DIRECTION only (ray_marching.cl:54-62 samples the environment with
current_ray.direction); where it leaves the volume does not matter.  If, from a coarse table of macro cells (event-free? smallest
SDF value?), a ray can be proven to leave the volume without a Hit within its remaining steps, its remaining step fetches -- far
field, one 128-byte line each -- need not be made at all, and the result stays bit-identical."""
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
variants = "--variants" in args  # compare the cheaper proofs (one table lookup instead of a walk over the cells)
w, h = (1920, 1080) if n >= 512 else (960, 544)
vol = scene.phantom(n)
tf = orc_ffi.parse_tf(scene.tf_default_source())
sdf = np.load(cache) if cache and os.path.exists(cache) else orc_ffi.sdf_build(vol, tf)[0]
env = scene.env_map(4096, 2048)
pos, d = scene.default_camera(n)
event = (vol >= 500) & (vol <= 1200)


def macro_table(m, margin=2):
    """0 where the cell dilated by `margin` voxels holds an event voxel or a non-positive SDF value, else its smallest SDF value"""
    bad = event | (sdf <= 0)
    val = np.where(bad, 0, sdf).astype(np.uint8)
    c = (n + m - 1) // m
    out = np.zeros((c, c, c), np.uint8)
    for cz in range(c):
        z0, z1 = max(cz * m - margin, 0), min((cz + 1) * m + margin, n)
        for cy in range(c):
            y0, y1 = max(cy * m - margin, 0), min((cy + 1) * m + margin, n)
            for cx in range(c):
                x0, x1 = max(cx * m - margin, 0), min((cx + 1) * m + margin, n)
                out[cz, cy, cx] = val[z0:z1, y0:y1, x0:x1].min()
    return out


def report(sc, label):
    L = dict(zip(orc_ffi.LOCALITY_NAMES, (int(v) for v in sc.locality)))
    items = sc.counter_dict()["n_hit"]
    print("%s: %.2f tries, %.2f granted per item; %.2f of the %.2f step fetches per item avoided (%.1f %%); wrong certificates: %d" % (
        label, L["cert_tried"] / items, L["cert_granted"] / items, L["cert_saved"] / items, L["steps"] / items,
        100.0 * L["cert_saved"] / L["steps"], int(sc.locality[orc_ffi.LOCALITY_CERT_WRONG])), flush=True)


if variants:
    # what k_bounce builds: cells of 16^3 voxels dilated by a brick; mode 0 walks the cells the ray crosses (a 3-D DDA), mode 2
    # asks for the whole box between the cell and the volume corner of the direction's octant to be free and bounds the steps by
    # t_exit / min_free, mode 3 bounds them by the box's diagonal / its smallest SDF value (no t_exit: ONE table lookup)
    print("scene: phantom(%d), %dx%d, default camera, default TF, 1 pass; cells of 16^3 voxels dilated by 8" % (n, w, h))
    margin = int(args[args.index("--margin") + 1]) if "--margin" in args else 8
    if margin != 8:
        print("(cells dilated by %d voxels instead of a brick)" % margin)
    cell = int(args[args.index("--cell") + 1]) if "--cell" in args else 16
    if cell != 16:
        print("(cells of %d^3 voxels)" % cell)
    table = macro_table(cell, margin=margin)
    finer = "--finer" in args  # round 3: direction bins finer than an octant (dominant axis x minor-slope bins), same one-lookup form
    modes = ((0, 1, "walk"), (2, 4, "octant box, t_exit / 4"), (3, 2, "octant box, diagonal / box minimum"))
    if finer:
        modes = ((3, 2, "octant box, diagonal / min (ships)"), (4, 2, "octant box, t_exit / min"), (5, 2, "dominant-axis pyramid (24 bins)"),
                 (6, 2, "pyramid, 2 x 2 slope bins (96)"), (8, 2, "pyramid, 4 x 4 slope bins (384)"), (0, 1, "walk"))
    for mode, min_free, name in modes:
        for t in ((8, 16) if finer else (8, 16, 32)):
            sc = orc_ffi.Scene(vol, sdf, env, tf, (w, h), mode=orc_ffi.MODE_IMAGE_SPACE, threads=len(os.sched_getaffinity(0)))
            sc.locality = np.zeros(orc_ffi.LOCALITY_TOTAL, np.uint64)
            sc.macro_free_min, sc.macro_m, sc.cert_t, sc.cert_mode, sc.cert_min_free = table, cell, t, mode, min_free
            sc.render(pos, d, scene.glibc_rand(1)[0])
            report(sc, "%-36s tried at every step >= %2d" % (name, t))
    sys.exit(0)

print("scene: phantom(%d), %dx%d, default camera, default TF, 2 passes" % (n, w, h))
for m in (64, 32, 16):
    table = macro_table(m)
    for t in (8, 16, 32):
        sc = orc_ffi.Scene(vol, sdf, env, tf, (w, h), mode=orc_ffi.MODE_IMAGE_SPACE, threads=len(os.sched_getaffinity(0)))
        sc.locality = np.zeros(orc_ffi.LOCALITY_TOTAL, np.uint64)
        sc.macro_free_min, sc.macro_m, sc.cert_t = table, m, t
        for s in scene.glibc_rand(2):
            sc.render(pos, d, s)
        L = dict(zip(orc_ffi.LOCALITY_NAMES, (int(v) for v in sc.locality)))
        wrong = int(sc.locality[orc_ffi.LOCALITY_CERT_WRONG])
        items = sc.counter_dict()["n_hit"]
        print("cells of %2d^3 voxels (%5.1f %% event-free, table %7d B), certificate tried when the next step is >= %2d: "
              "%.2f tries, %.2f granted per item; %.2f of the %.2f step fetches per item avoided (%.1f %%); wrong certificates: %d" % (
                  m, 100.0 * (table > 0).mean(), table.size, t, L["cert_tried"] / items, L["cert_granted"] / items,
                  L["cert_saved"] / items, L["steps"] / items, 100.0 * L["cert_saved"] / L["steps"], wrong))
