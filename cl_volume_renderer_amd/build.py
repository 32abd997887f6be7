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

SOURCES = ["clwh_runtime.hip", "render_kernels.hip", "sdf_kernels.hip", "volume_kernels.hip", "exchange_kernels.hip", "tf_parse.cpp", "tf_jit.cpp"]
HEADERS = ["clwh_internal.hpp", "device_math.hpp", "render_device.hpp", "packed_volume.hpp", "env_fast.hpp"]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
]


HOST_LIB = os.path.join(HERE, "libclvr_host.so")
APP = os.path.join(HERE, "app")
HOST_SOURCES = ["renderer.cpp", "reference_volume.cpp", "signed_distance_field.cpp", "nrrd_loader.cpp", "hdre_loader.cpp", "png_reader.cpp", "jpeg_reader.cpp", "tf_part.cpp", "host_c_api.cpp"]
HOST_PROGRAMS = {"sdf_test": "sdf_test_main.cpp", "sdf_benchmark": "sdf_benchmark_main.cpp", "clvr_headless": "headless_main.cpp"}
CXX = os.environ.get("CXX", "g++")


def build_host(force: bool = False, verbose: bool = False) -> str:
    """The C++ host mirror of the reference's renderer / reference_volume / signed_distance_field,
    compiled with plain g++ against include/clw_*.hpp and linked to libclwhip.so."""
    deps = [os.path.join(APP, f) for f in os.listdir(APP)]
    deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))] + [LIB]
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in deps):
        return HOST_LIB
    cmd = [CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", APP]
    cmd += [os.path.join(APP, s) for s in HOST_SOURCES]
    cmd += ["-o", HOST_LIB, "-L", HERE, "-lclwhip", "-lz", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    # the reference's two headless programs over the same classes (tests/sdf/sdf_test.cpp, app/sdf_benchmark.cpp)
    for name, src in HOST_PROGRAMS.items():
        cmd = [CXX, "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", APP, os.path.join(APP, src),
               "-o", os.path.join(HERE, name), "-L", HERE, "-lclvr_host", "-lclwhip", "-lz", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HOST_LIB


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "clwh.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    extra = os.environ.get("CLVR_EXTRA_HIPCC_FLAGS", "").split()
    cmd = [HIPCC] + FLAGS + extra + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB, "-lhiprtc"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
