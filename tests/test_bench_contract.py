"""The driver's contract with bench.py: one JSON line on stdout with the fields it reads, the
roofline and cpu_baseline objects, on a reduced workload (the full C2 run is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--npoints", "200000", "--trees", "8", "--nq", "1000", "--no-cpu-baseline",
                          "--other-configs", "none"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
                "cpu_baseline", "knn"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert j["unit"] == "vectors/s" and j["value"] > 0 and j["ms_per_step"] > 0
    assert abs(j["value"] - 200000 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert j["knn"]["value"] > 0 and j["knn"]["unit"] == "queries/s"
    ex = j["knn"]["exchange_one_rank"]          # record -> ncclAllGather -> merge really ran
    assert "error" not in ex and ex["trees"] == 1 and ex["knn_ms_with_forced_exchange"] > 0
    assert j["other_configs"] is None


@pytest.mark.gpu
def test_other_configs_child_reports_c3_with_its_roofline():
    """The other_configs leg (a child process of bench.py): BASELINE configs[2] (C3, 1 M x 784 CSR)
    with build / kNN times and a roofline object for the CSR projection kernel."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--_other-child",
                          "--other-configs", "c3", "--steps", "1", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])["c3"]
    assert "error" not in j, j
    assert j["build_ms"] > 0 and j["knn_queries_per_s"] > 0
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1.0 and "proj_csr" in r["kernel"]


def test_bench_gpus_n_needs_no_launcher():
    """`python bench.py --gpus 2` is a complete command: ONE process drives the GPUs through
    rpt_comm_init (librccl).  Without a launcher it must get as far as asking the HIP runtime for
    its devices — here, on a box with fewer than 2 of them, that is where it stops."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=300)
    msg = out.stderr + out.stdout
    if out.returncode != 0:          # (a node with >= 2 GPUs simply runs the bench)
        assert "torch.distributed.run" not in msg
        assert "HIP device" in msg, msg[-2000:]


def test_bench_refuses_a_launcher_mismatch():
    """--gpus N under a launcher that started a different number of ranks must not run."""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "launcher started 4 ranks" in (out.stderr + out.stdout)
