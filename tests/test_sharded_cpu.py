"""The N>1 path on CPU: two gloo ranks shard the utterance index space with no
data-path collective; whole-job item count and checksum-of-checksums equal the
single-process run.  The per-batch transform injected here is the CPU oracle
(the GPU transform is exercised by the gpu tests and bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from wfe_amd.sharded import shard_range
    for n in (0, 1, 7, 100000, 100003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(100000, 3, 8) == (37500, 50000)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _clip(i, n=1600):
    return (0.1 * np.random.default_rng(7000 + i).standard_normal(n)).astype(np.float32)


def _make_batch(begin, count):
    return ([_clip(i) for i in range(begin, begin + count)],)


def _transform(waves):
    from oracle import logmel as ol          # checker injected as the transform, CPU test only
    return torch.from_numpy(ol.logmel_f32(waves, n_mel=80, n_samples=1600))


def _worker(rank, world, port, n_items, q):
    for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as td
    from wfe_amd.sharded import run_sharded
    torch.set_num_threads(1)
    td.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = run_sharded(n_items, 4, _make_batch, _transform)
    q.put((rank, res))
    td.barrier()
    td.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    from wfe_amd.sharded import run_sharded
    n_items = 19
    single = run_sharded(n_items, 4, _make_batch, _transform, rank=0, world=1)
    assert single["items"] == n_items and single["shard"] == (0, n_items)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0]["shard"] == (0, 10) and results[1]["shard"] == (10, 19)
    for r in (0, 1):
        assert results[r]["items"] == n_items and results[r]["world"] == 2
        assert results[r]["checksum"] == pytest.approx(single["checksum"], rel=1e-12)
        assert results[r]["seconds"] == results[0]["seconds"]        # max over ranks, same on every rank
