"""CPU-side checks of bench.py: it imports, its byte-accounting helpers are what DESIGN.md says, and without a GPU it
refuses to run (no CPU fallback) -- with one process or with self-spawned ranks."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_byte_accounting_splits_the_path_by_kernel():
    b = _bench()
    c = dict(n_sdf=9.42, n_vol=10.50, n_env=1.127, n_add=0.16, n_sdf_primary=4.61, n_vol_primary=5.10, n_env_primary=0.84, samples=1.0)
    primary, bounce, contract = b.bytes_primary(c), b.bytes_bounce(c), b.bytes_contract_image_space(c)
    assert abs(bounce - (4.81 + 2 * 5.40 + 4 * 0.287 + 8 * 0.16)) < 1e-6          # what k_bounce executes: 18.0 B per sample
    assert abs(primary - (4.61 + 2 * 5.10 + 4 * 0.84 + 4)) < 1e-6                  # what k_primary executes, once per camera
    # the SURVEY 8d contract counts every texel once: primary + bounce texels, the atomic, the frame write
    assert abs(contract - (primary + bounce)) < 1e-6
    # what k_bounce fetches at texel granularity: one step byte per march step, one 8-byte hit record per secondary Hit, the env
    # texels of the bounce phase, one 8-byte atomic per sample -- for a gradient TF the contract's seven texels per step drop out
    c["n_hit_bounce"] = 0.15
    assert abs(b.bytes_bounce_executed(c) - (4.81 + 8 * 0.15 + 4 * 0.287 + 8 * 0.16)) < 1e-6
    cg = dict(c, n_vol=10.50 + 6 * 4.0)   # a TF that reads `gradient`: six more texels per classified step in the contract ...
    assert b.bytes_bounce(cg) > b.bytes_bounce(c) + 40 and b.bytes_bounce_executed(cg) == b.bytes_bounce_executed(c)   # ... none executed
    assert set(b.PRESETS) == {2, 3, 4, 5} and b.PRESETS[2]["spp"] == 64 and b.PRESETS[4]["volume"] == 2048


def test_bench_refuses_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                         text=True, env=env, timeout=300)
    assert one.returncode != 0 and "no GPU visible" in (one.stderr + one.stdout)
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert two.returncode != 0                       # the self-spawned ranks fail loudly, and so does the parent
    assert "REHEARSAL" in two.stderr                 # fewer GPUs than ranks: it says so before starting them
