"""bench.py's multi-rank plumbing on CPU: `--gpus N` outside a torchrun launch starts N ranks
itself (the driver's own command line), relays rank 0's line and fails loudly when ranks or
devices are missing.  The 2-rank run uses gloo and bench.py's stand-in transform
(--stub-transform): the launch, the sharding and the reductions are what is tested here --
the HIP transform is covered by the gpu tests."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_launch_command_is_the_drivers():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "7"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [BENCH, "--gpus", "4", "--steps", "7"]


def test_two_gloo_ranks_through_the_self_launch_equal_one():
    common = ["--shard-clips", "301", "--batch", "32", "--stub-transform"]
    one = _run(["--gpus", "1"] + common)
    assert one.returncode == 0, one.stderr[-2000:]
    two = _run(["--gpus", "2"] + common)
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    b = json.loads(two.stdout.strip().splitlines()[-1])
    assert len([ln for ln in two.stdout.splitlines() if ln.startswith("{")]) == 1     # ONE line, from rank 0
    assert (a["n_gpus"], b["n_gpus"]) == (1, 2) and a["items"] == b["items"] == 301
    assert a["checksum"] == b["checksum"]                   # clip i depends on i alone, whatever the sharding
    assert b["data"] == "stub" and "not a measurement" in b["config"]["workload"]
    assert b["steps"] == 5 and a["steps"] == 10             # ceil(301 / (32 * world))


def test_missing_devices_and_mismatched_world_fail_loudly():
    r = _run(["--gpus", "2"])                               # no GPU in the build container
    if r.returncode == 0:
        pytest.skip("two GPUs are visible here")
    assert "GPU(s) visible" in (r.stderr + r.stdout)
    r = _run(["--gpus", "2", "--shard-clips", "64", "--stub-transform"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "they must agree" in (r.stderr + r.stdout)
    r = _run(["--gpus", "0"])
    assert r.returncode != 0


def test_self_launch_parent_never_loads_torch_or_hip(tmp_path):
    """The parent of a self-launched multi-rank run must not be able to reach hipInit before it spawns
    the ranks: bench.py's top level imports the standard library only, devices are
    counted from sysfs.  The parent reports what it had mapped when it spawned."""
    report = tmp_path / "maps.json"
    r = _run(["--gpus", "2", "--shard-clips", "64", "--batch", "32", "--stub-transform"],
             {"WFE_BENCH_PARENT_MAPS": str(report)})
    assert r.returncode == 0, r.stderr[-2000:]
    seen = json.loads(report.read_text())
    assert seen == {"loaded": [], "torch_imported": False}
    # and importing the module alone pulls in neither torch nor numpy
    code = "import sys; sys.path.insert(0, %r); import bench; print('torch' in sys.modules, 'numpy' in sys.modules)" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.stdout.split() == ["False", "False"], out.stdout + out.stderr


def test_visible_gpu_count_reads_env_lists_without_hip(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    base = bench.visible_gpu_count()                         # None in a container without /sys/class/kfd
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    n = bench.visible_gpu_count()
    assert n == (3 if base is None else min(base, 3))
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
