"""GPU: bench.py's multi-rank path with the REAL front end, rehearsed on the one GPU of the test box
(WFE_BENCH_ONE_GPU=1: both ranks on GPU 0, gloo between them -- RCCL does not put two ranks on one device).
What it covers that the CPU plumbing test (tests/test_bench_cpu.py, stand-in transform) cannot: two processes
that each own a HIP front end, the sharded workload of BASELINE config 4 giving the one-rank checksum, the
weak-scaling default workload's barriers and max-time reduction, ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=280):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _line(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]           # ONE line, from rank 0
    return json.loads(lines[0])


def test_two_ranks_shard_config4_like_one(tmp_path):
    common = ["--shard-clips", "1200", "--batch", "128", "--n-mel", "128"]
    one = _run(["--gpus", "1"] + common)
    assert one.returncode == 0, one.stderr[-2000:]
    report = tmp_path / "maps.json"
    two = _run(["--gpus", "2"] + common, {"WFE_BENCH_ONE_GPU": "1", "WFE_BENCH_PARENT_MAPS": str(report)})
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = _line(one), _line(two)
    assert (a["n_gpus"], b["n_gpus"]) == (1, 2) and a["items"] == b["items"] == 1200
    assert a["checksum"] == b["checksum"]                   # clip i depends on i alone, whatever the sharding
    assert "REHEARSAL" in b["config"]["workload"] and b["scaling"] == "strong"
    assert json.loads(report.read_text()) == {"loaded": [], "torch_imported": False}     # the GPU-free parent, on a GPU box


def test_two_ranks_weak_scaling_line():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", "64", "--no-cpu-baseline"], {"WFE_BENCH_ONE_GPU": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    ln = _line(r)
    assert ln["n_gpus"] == 2 and ln["steps"] == 5 and ln["scaling"] == "weak"
    assert ln["clips_per_sec"] == pytest.approx(2 * 64 * 5 / (ln["ms_per_step"] * 5 / 1e3), rel=1e-6)
    assert "REHEARSAL" in ln["config"]["parallelism"] and "also" not in ln and "cpu_baseline" not in ln
    assert 0.0 < ln["roofline"]["frac"] <= 1.0
