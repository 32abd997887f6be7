"""Generates tests/golden/sdf_phantom_checksums.json: SHA-256, sums and per-launch write counts of the ORACLE's
signed-distance field (oracle/orc_sdf.c, pinned on the reference's own golden vector) for the synthetic phantoms the
full-size GPU tests build -- 512^3 takes minutes on the CPU, so the GPU test compares against these instead of
re-running the oracle.  Run from the repository root:  python tests/golden/make_sdf_checksums.py"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cl_volume_renderer_amd import scene  # noqa: E402
from oracle import orc_ffi  # noqa: E402

out = {}
for n, tf_name in ((256, "default"), (512, "default"), (512, "gradient")):
    t = time.time()
    vol = scene.phantom(n)
    src = scene.tf_default_source() if tf_name == "default" else scene.tf_gradient_source()
    sdf, launches, counts = orc_ffi.sdf_build(vol, orc_ffi.parse_tf(src))
    out["phantom(%d) %s tf" % (n, tf_name)] = {
        "sha256": hashlib.sha256(sdf.tobytes()).hexdigest(), "launches": int(launches),
        "sum": int(sdf.astype(np.int64).sum()), "abs_sum": int(np.abs(sdf.astype(np.int64)).sum()),
        "layer_write_counts": [int(c) for c in counts]}
    print(n, tf_name, "%.0fs" % (time.time() - t), flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "sdf_phantom_checksums.json"), "w") as f:
        json.dump(out, f, indent=1)
