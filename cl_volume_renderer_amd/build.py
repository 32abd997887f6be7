"""Build recipe of the product library ``libclwhip.so`` (HIP for gfx950, cross-compiles without a GPU).

Float semantics are part of the contract (DESIGN.md "Semantics"): no FMA contraction, correctly
rounded fp32 division / sqrt, denormals preserved.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libclwhip.so")

SOURCES = ["clwh_runtime.hip", "render_kernels.hip", "sdf_kernels.hip", "tf_parse.cpp"]
HEADERS = ["clwh_internal.hpp", "device_math.hpp", "render_device.hpp", "packed_volume.hpp"]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "clwh.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    cmd = [HIPCC] + FLAGS + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
