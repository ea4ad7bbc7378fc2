"""GPU: the driver's bench contract -- `python bench.py` prints ONE JSON line with the agreed keys, a roofline
object whose numbers are consistent with each other, and (default run) cpu_baseline + the `also` legs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_main_line_and_roofline_are_consistent():
    d = _run("--steps", "3", "--warmup", "1", "--no-cpu", "--also-only", "k2_funk")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["scaling"] == "n/a"                            # one GPU: neither weak nor strong
    # the epoch is a chain of hand-offs: the floor that chain sets is printed beside the byte roofline
    cf = d["roofline"]
    assert cf["chain_levels"] == d["config"]["levels"] and cf["min_hop_us"] > 0
    assert abs(cf["chain_floor_ms"] - cf["chain_levels"] * cf["min_hop_us"] * 1e-3) < 1e-12 and 0 < cf["frac_of_chain_floor"] < 1
    assert abs(cf["us_per_chain_level"] * cf["chain_levels"] - cf["avg_launch_us"]) < 1e-6 * cf["avg_launch_us"]
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "cfg2" in d["config"]["workload"]
    assert abs(d["value"] - 10000000 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-9
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / average launch duration
    assert abs(r["achieved"] * 1e9 - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6)) / (r["achieved"] * 1e9) < 1e-9
    # cfg 2 is served by the static resident-side epoch (ONE launch per epoch); the PMC traffic figure comes from
    # profiles/ and is only attached when it was taken on this very kernel
    assert r["kernel"].startswith("k1_res<") and "static" in r["kernel"] and r["launches_per_epoch"] == 1
    assert d["config"]["epoch_kernel"] == "static"
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"]
    assert 2.5e9 < d["value"] < 5.0e9                       # ~3.1 G updates/s (the level schedule: ~1.95 G)
    k2 = d["also"]["k2_funk"]
    # shuffled order at cfg 2: levels of ~16 K ratings -> the resident-side Funk pass (lane = item, values inside the granules)
    assert k2["unit"] == "scalar-updates/s" and k2["roofline"]["kernel"].startswith("k2_res<")
    assert k2["value"] > 4.0e9                              # round 1's launch path: 4.8 G


def test_multi_path_rehearsal_prints_one_line_with_the_exact_leg(monkeypatch):
    """The N > 1 code path of bench.py with ONE rank (forced 1-rank RCCL communicator): cfg 3, user-sharded plan,
    real all-reduce, the DSGD leg -- the line must carry `strong` scaling, the cfg-3 workload and also.dsgd_exact."""
    env = dict(os.environ, MFX_BENCH_FORCE_MULTI="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--users", "200000", "--items", "40000", "--nnz", "8000000"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and "RCCL all-reduce" in d["config"]["workload"]
    assert "cpu_baseline" not in d
    assert d["also"]["dsgd_exact"]["value"] > 0 and "error" not in d["also"]["dsgd_exact"]
    # the averaging `value` carries its RMSE gap to the sequential loop and names the parity-holding mode
    assert "error" not in d["one_gpu_same_workload"] and d["speedup_vs_one_gpu"] > 0
    g = d["rmse_gap_vs_sequential"]
    assert "error" not in g and g["budget"] == 1e-5 and abs(g["value"]) < 0.5 and "dsgd_exact" in d["parity_mode"]
