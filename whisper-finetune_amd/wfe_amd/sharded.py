"""Data-parallel sharding of the front end over the GPUs of one node.

Utterances are independent (the widest dependency is the per-utterance max), so
the path shards with NO data-path collective: every rank takes a contiguous
slice of the utterance index space (the split torch's DistributedSampler /
SURVEY 8e prescribe for BASELINE config 4), runs its own front end on its own
GPU and keeps its features.  torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used only for the barrier around the timed region, the max-over-ranks
time and the reduction of per-rank checksums.
"""
from __future__ import annotations

import time
from typing import Callable, Optional, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end): the first n_items % world ranks get one more."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def _dist():
    import torch.distributed as td
    return td if td.is_available() and td.is_initialized() else None


def run_sharded(n_items: int, batch: int, make_batch: Callable[[int, int], tuple],
                transform: Callable[..., torch.Tensor], sync: Optional[Callable[[], None]] = None,
                rank: Optional[int] = None, world: Optional[int] = None, checksum_stride: int = 1) -> dict:
    """Runs ``transform(*make_batch(begin, count))`` over this rank's shard in
    batches of ``batch`` and returns whole-job figures (identical on every rank):
    items, seconds (max over ranks, bracketed by barriers), and a checksum of
    checksums (sum over all features, float64) that is independent of the world size.

    make_batch(begin, count) must derive utterance i from i alone (counter-based
    seeds), so that any sharding sees the same data.  checksum_stride > 1 sums every
    stride-th element of each item only (a throughput run should not spend a full extra
    read of the output per batch on its own checksum); still independent of the world size."""
    td = _dist()
    rank = (td.get_rank() if td else 0) if rank is None else rank
    world = (td.get_world_size() if td else 1) if world is None else world
    begin, end = shard_range(n_items, rank, world)

    def fence():
        if td:
            td.barrier()
        if sync:
            sync()

    fence()
    t0 = time.perf_counter()
    acc = torch.zeros((), dtype=torch.float64)
    done = 0
    for b0 in range(begin, end, batch):
        count = min(batch, end - b0)
        out = transform(*make_batch(b0, count))
        # sampled PER ITEM (every stride-th element of each utterance's features), so that the sum
        # does not depend on where the batch boundaries of a sharding fall
        part = out if checksum_stride == 1 else out.reshape(out.shape[0], -1)[:, ::checksum_stride]
        acc = acc.to(out.device) + part.sum(dtype=torch.float64)  # stays on the device; no host sync per batch
        done += count
    fence()
    elapsed = time.perf_counter() - t0
    stats = torch.tensor([elapsed, float(done), float(acc.item())], dtype=torch.float64)
    if td:
        dev = acc.device if td.get_backend() == "nccl" else torch.device("cpu")
        tmax = stats[:1].clone().to(dev)
        tsum = stats[1:].clone().to(dev)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        td.all_reduce(tsum, op=td.ReduceOp.SUM)
        stats = torch.cat([tmax.cpu(), tsum.cpu()])
    return {"items": int(stats[1].item()), "seconds": float(stats[0].item()), "checksum": float(stats[2].item()),
            "world": world, "shard": (begin, end)}
