#!/usr/bin/env python3
"""bench.py -- log-mel audio-hours/sec of the HIP front end on N MI355X.

One "step" = one pass of the hot path (pad/trim -> STFT -> mel -> log -> clamp)
over one batch of 256 synthetic 30 s clips that are already resident in HBM
(BASELINE.json configs[1]: whisper-small 80-mel front end, batch 256, no
augmentation).  With N > 1 (launched by torch.distributed.run, one rank per
GPU) every rank runs its own front end on its own batch -- utterances are
independent, there is no data-path collective -- so scaling is weak
and `value` is the sum over ranks.

Prints ONE JSON line on rank 0.  Besides the driver's contract it carries
  roofline      achieved = the f32 MFMA FLOPs the folded DFT + mel kernel has to issue
                (0.243 GF per 30 s clip at 80 mel: the folds remove 4/5 of the
                1.061 GF a plain 400x402 DFT-GEMM is priced at in SURVEY 8d; that
                figure is reported as priced_tflops) x clips per launch / mean device
                time of the kernel, measured with HIP events on the launch stream
                inside the timed region; peak = 157.3 TFLOP/s (f32 MFMA,
                MI355X_MICROARCH.md); frac = achieved / peak <= 1.  hbm_floor_frac =
                algorithmic bytes / 8 TB/s / kernel time.
  cpu_baseline  the reference's own CPU path -- the installed transformers
                WhisperFeatureExtractor called per utterance as utils/reader.py:195 does
                (kind "reference"; the oracle port oracle/logmel.py only where that class
                is not importable, kind "port") -- timed on this box's host cores on a
                bounded sample, rank 0 at N=1 only, BEFORE this process touches the GPU;
                sub-keys threads1 (one torch thread) and procs8 (8 single-threaded
                processes, the reference's DataLoader(num_workers=8) structure).
  also          (default N=1 run only) short timed passes of the other BASELINE legs that
                fit one GPU -- config 3 (augmentation chain fused), 128 mel + the config-4
                1-GPU leg (sharded clips), and the PCIe-inclusive rates -- each with the same
                HIP-event kernel timing.  The headline `value` is untouched by them.

The parent of a self-launched multi-rank run (--gpus N without WORLD_SIZE) imports nothing
that loads the HIP runtime: this module's top level is standard library only, torch and
numpy are imported by the ranks.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

np = None        # numpy and torch: imported by _heavy(), never by the parent of a self-launch
torch = None

SR = 16000
N_SAMPLES = 480000
CLIP_HOURS = 30.0 / 3600.0
ALGO_GFLOP = {80: 1.0613, 128: 1.1192}       # 2*3000*400*402 + 2*3000*201*M
ALGO_BYTES = {80: 2.88e6, 128: 3.456e6}      # waveform read + log-mel write
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
DTYPE_NAME = {"float32": "f32", "float16": "f16", "bfloat16": "bf16"}


def _heavy() -> None:
    global np, torch
    if torch is None:
        import numpy as _np
        import torch as _torch
        np, torch = _np, _torch


def executed_gflop_per_clip(n_mel: int) -> float:
    """The algorithmic work of the FOLDED formulation: 47 tiles x 4 waves x
    (25 k-steps x (14 odd-class + 8 even-class) + 8 = 558 DFT + 72 | 84 mel) v_mfma_f32_16x16x4_f32 of
    2048 FLOP each (round 3: the even class is folded once more; 700 DFT MFMAs before).
    SQ_INSTS_MFMA per launch equals this count exactly (profiles/)."""
    return 47 * 4 * (558 + (72 if n_mel <= 80 else 84)) * 2048 / 1e9


def measured_traffic(batch: int, n_mel: int, augment: bool, lengths: str):
    """HBM bytes per launch of kernel 1 (and of the clamp kernel) from profiles/traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command line, one entry per
    (batch, n_mel, augment, lengths).  None when that configuration was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        for e in tj["entries"]:
            if (e["batch"], e["n_mel"], bool(e["augment"]), e.get("lengths", "full")) == (batch, n_mel, augment, lengths):
                return e
    except Exception:  # noqa: BLE001
        pass
    return None


# ------------------------------------------------------------------ multi-rank launch (GPU-free parent)
def launch_command(n_gpus: int, argv, port: int):
    """The command the driver itself uses for N > 1 (one rank per GPU over RCCL); bench.py
    starts it as a CHILD when asked for --gpus N outside a torchrun launch."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def visible_gpu_count():
    """GPUs this process tree can see, counted WITHOUT loading the HIP runtime: the kfd topology
    in sysfs (nodes with SIMDs are GPUs), narrowed by the *_VISIBLE_DEVICES lists.  None when
    sysfs says nothing (no kfd driver in this container): then the ranks decide -- each one
    fails loudly when its local rank has no device."""
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    n = None
    try:
        n = 0
        for d in os.listdir(nodes):
            try:
                with open(os.path.join(nodes, d, "properties")) as f:
                    props = dict(ln.split(None, 1) for ln in f.read().splitlines() if " " in ln)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        n = None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([x for x in v.split(",") if x.strip() != ""])
            n = listed if n is None else min(n, listed)
    return n


def loaded_gpu_runtime():
    """Names of GPU-runtime / torch libraries mapped into THIS process (for the parent's self-check)."""
    hits = set()
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                path = ln.rsplit(None, 1)[-1] if "/" in ln else ""
                base = os.path.basename(path)
                if base.startswith(("libamdhip64", "libhsa-runtime64", "libtorch", "libc10", "librccl", "libwfe_hip")):
                    hits.add(base)
    except OSError:
        pass
    return sorted(hits)


def one_gpu_rehearsal() -> bool:
    """WFE_BENCH_ONE_GPU=1: every rank of a multi-rank run uses GPU 0 and the ranks talk over gloo -- the
    whole multi-rank path (launch, rendezvous, shards, barriers, reductions, rank 0's line) with the real
    front end on a box that has ONE GPU (tests/test_bench_gpu.py).  RCCL does not put two ranks on one
    device; the line such a run prints says so and is never a measurement."""
    return os.environ.get("WFE_BENCH_ONE_GPU", "0") == "1"


def self_launch(args, argv) -> int:
    """--gpus N > 1 without WORLD_SIZE: start the N ranks from a process that has loaded neither
    torch nor the HIP runtime (nothing between process start and the spawn can reach hipInit:
    the module's top level is standard library only, devices are counted from sysfs), relay
    rank 0's JSON line, exit non-zero if any rank fails.  Never retried, never re-exec'd."""
    import socket
    import subprocess
    if not args.stub_transform and not one_gpu_rehearsal():
        n_dev = visible_gpu_count()
        if n_dev is not None and n_dev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    loaded = loaded_gpu_runtime()
    report = os.environ.get("WFE_BENCH_PARENT_MAPS")       # tests: what the parent had mapped when it spawned
    if report:
        with open(report, "w") as f:
            json.dump({"loaded": loaded, "torch_imported": "torch" in sys.modules}, f)
    if loaded or "torch" in sys.modules:
        raise SystemExit(f"bench.py: the launching parent has GPU runtime libraries mapped ({loaded}); refusing to spawn ranks")
    proc = subprocess.run(launch_command(args.gpus, argv, port), env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        raise SystemExit(f"bench.py --gpus {args.gpus}: a rank failed (exit {proc.returncode})")
    if not lines:
        raise SystemExit(f"bench.py --gpus {args.gpus}: no JSON line from rank 0")
    print(lines[-1], flush=True)
    return 0


# ------------------------------------------------------------------ inputs
def device_clips(batch: int, device, seed0: int):
    """SURVEY 8d config 2 recipe (0.1*N(0,1), amplitude-modulated at 3 Hz),
    generated on the device from per-clip seeds."""
    t = torch.arange(N_SAMPLES, device=device, dtype=torch.float32) / SR
    env = 0.5 * (1.0 + torch.sin(2 * np.pi * 3.0 * t))
    out = torch.empty((batch, N_SAMPLES), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    for b in range(batch):
        g.manual_seed(seed0 + b)
        out[b] = 0.1 * torch.randn(N_SAMPLES, generator=g, device=device, dtype=torch.float32) * env
    return out


# ------------------------------------------------------------------ CPU baseline (before the GPU is touched)
def _cpu_clips(n: int):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    return [cases.synth_clip(i) for i in range(n)]


def _reference_extractor(n_mel: int):
    """The class the reference calls (utils/reader.py:192-198 through WhisperProcessor): third-party,
    present in this image; None where it is not importable."""
    try:
        from transformers import WhisperFeatureExtractor
        return WhisperFeatureExtractor(feature_size=n_mel)
    except Exception:  # noqa: BLE001
        return None


def _time_calls(fn, clips, budget_s: float):
    fn(clips[0])                                                 # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        fn(clips[n % len(clips)])
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            return n, el


def cpu_worker(n_mel: int, seconds: float) -> None:
    """One of cpu_baseline's 8 worker processes (the reference's DataLoader(num_workers=8) structure:
    one single-threaded extractor per process).  Prints `calls seconds`."""
    _heavy()
    torch.set_num_threads(1)
    fe = _reference_extractor(n_mel)
    clips = _cpu_clips(4)
    if fe is None:
        from oracle import logmel as ol   # baseline only
        n, el = _time_calls(lambda c: ol.logmel_f32([c], n_mel=n_mel), clips, seconds)
    else:
        n, el = _time_calls(lambda c: fe(c, sampling_rate=SR), clips, seconds)
    print(f"{n} {el:.6f}", flush=True)


def cpu_baseline(n_mel: int, budget_s: float = 24.0) -> dict:
    """Reference CPU path timed beside the GPU path: per-utterance calls, as
    CustomDataset.__getitem__ makes them (utils/reader.py:195).  Bounded to ~budget_s seconds
    of wall time in all: all host threads torch gives us (`value`), one thread, 8 processes."""
    import subprocess
    _heavy()
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    clips = _cpu_clips(8)
    fe = _reference_extractor(n_mel)
    from oracle import logmel as ol   # checker / baseline only
    port = lambda c: ol.logmel_f32([c], n_mel=n_mel)           # noqa: E731
    ref = (lambda c: fe(c, sampling_rate=SR)) if fe is not None else None   # noqa: E731
    main_fn = ref or port
    # 8 single-threaded worker processes FIRST (they start while nothing else loads the cores)
    per = budget_s / 4.0
    workers = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(per), "--n-mel", str(n_mel)],
                                stdout=subprocess.PIPE, text=True, env=dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1"))
               for _ in range(8)]
    procs8 = None
    try:
        outs = [w.communicate(timeout=per + 240)[0] for w in workers]
        calls = [(int(o.split()[0]), float(o.split()[1])) for o in outs if o.strip()]
        if len(calls) == 8:
            procs8 = sum(n for n, _ in calls) * CLIP_HOURS / max(el for _, el in calls)
    except Exception:  # noqa: BLE001
        for w in workers:
            if w.poll() is None:
                w.kill()
    torch.set_num_threads(cores)
    n, el = _time_calls(main_fn, clips, per)
    res = {"value": n * CLIP_HOURS / el, "unit": "audio-hours/sec", "cores": cores,
           "kind": "reference" if ref else "port",
           "sample": (f"{n} per-utterance calls of transformers.WhisperFeatureExtractor(feature_size={n_mel}) "
                      if ref else f"{n} per-utterance calls of oracle.logmel_f32 (torch.stft fp32 restatement) ")
                     + f"on 30 s synthetic clips, {el:.1f} s, torch threads={cores}"}
    t1 = time.perf_counter()
    (fe(clips, sampling_rate=SR) if ref else ol.logmel_f32(clips, n_mel=n_mel))
    res["batched8_audio_hours_per_sec"] = len(clips) * CLIP_HOURS / (time.perf_counter() - t1)
    torch.set_num_threads(1)
    n1, el1 = _time_calls(main_fn, clips, per / 2)
    res["threads1"] = {"value": n1 * CLIP_HOURS / el1, "cores": 1, "calls": n1}
    res["procs8"] = {"value": procs8, "cores": 8,
                     "note": "8 processes x 1 thread, each calling the extractor per utterance (DataLoader num_workers=8, finetune.py:26,129)"}
    torch.set_num_threads(cores)
    if ref:                                                       # the oracle port beside it, for continuity with rounds 1-2
        np_, elp = _time_calls(port, clips, per / 2)
        res["oracle_port_audio_hours_per_sec"] = np_ * CLIP_HOURS / elp
    return res


# ------------------------------------------------------------------ measurement helpers
def timed_steps(fe, step, steps: int, warmup: int, prewarm_ms: float, barrier):
    """W warm-up steps, then exactly K timed steps between two barriers; kernel times from the
    HIP events the library records around its two kernels inside the timed region."""
    # clock ramp: after an idle period the first ~40 launches run ~10 % slower.  Untimed steps for
    # prewarm_ms before the W warm-up steps, so that a short run (small K, W) measures the same
    # sustained rate as a long one; the timed region is still exactly K steps.
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < prewarm_ms:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    barrier()
    fe.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    k1_ms, k2_ms, launches = fe.kernel_times()
    fe.kernel_timing(False)
    return elapsed, k1_ms, k2_ms, launches


def roofline_block(n_mel: int, batch: int, lens, k1_ms, k2_ms, launches, augment: bool, lengths: str,
                   exact_input: bool) -> dict:
    k1_avg_s = k1_ms / max(launches, 1) / 1e3
    k2_avg_s = k2_ms / max(launches, 1) / 1e3
    # tiles whose 64 frames lie in the zero padding behind a short utterance do no DFT (--lengths speech)
    live = np.minimum(47, (lens.cpu().numpy().astype(np.int64) + 199) // 10240 + 1)
    live_frac = float(live.sum()) / (47.0 * batch)
    achieved = executed_gflop_per_clip(n_mel) * live_frac * batch / k1_avg_s / 1e3   # TFLOP/s the MFMA pipe ran
    priced = ALGO_GFLOP[n_mel] * batch / k1_avg_s / 1e3                              # SURVEY 8d, unfolded
    algo_bytes = ALGO_BYTES[n_mel] * batch
    traffic, traffic_src, clamp_gbs, clamp_traffic, extra = None, None, None, None, {}
    te = measured_traffic(batch, n_mel, augment, lengths)
    if te is not None and exact_input:
        traffic = (te["frontend_kernel"]["fetch_bytes"] + te["frontend_kernel"]["write_bytes"]) / 1e9
        ck = te.get("clamp_kernel")
        if ck and k2_ms > 0:                                                          # data-dependent: same clips only
            clamp_traffic = (ck["fetch_bytes"] + ck["write_bytes"]) / 1e9
            clamp_gbs = clamp_traffic / k2_avg_s
        for name in ("resolve_partial_kernel", "noise_meansq_kernel"):
            if name in te:
                extra[name + "_traffic"] = (te[name]["fetch_bytes"] + te[name]["write_bytes"]) / 1e9
        traffic_src = "profiles/traffic.json entry %r (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, GB per " \
                      "launch; algorithmic %.3f GB)" % (te.get("source", "?"), algo_bytes / 1e9)
    blk = {
        # executed f32 MFMA rate of the folded DFT + mel kernel against the matrix-pipe peak
        "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
        "kernel": "wfe::frontend_kernel", "kernel_ms_avg": k1_avg_s * 1e3, "launches": launches,
        "gflop_per_clip": executed_gflop_per_clip(n_mel), "live_tile_fraction": live_frac,
        # the same launch priced with SURVEY 8d's unfolded 400x402 DFT-GEMM (can exceed the peak:
        # the fold removes 3/4 of those FLOPs) -- not a roofline fraction
        "priced_gflop_per_clip": ALGO_GFLOP[n_mel], "priced_tflops": priced,
        # against the other roof: algorithmic bytes (waveform read + log-mel write) at 8 TB/s
        "hbm_algorithmic_gb": algo_bytes / 1e9, "hbm_algorithmic_gbs": algo_bytes / k1_avg_s / 1e9,
        "hbm_floor_ms": algo_bytes / (PEAK_HBM_GBS * 1e9) * 1e3,
        "hbm_floor_frac": algo_bytes / (PEAK_HBM_GBS * 1e9) / k1_avg_s,
        "clamp_kernel_ms_avg": k2_avg_s * 1e3, "clamp_kernel_traffic": clamp_traffic,
        # the elementwise stage against HBM: measured bytes (profiles/traffic.json) / measured time
        "clamp_kernel_hbm_gbs": clamp_gbs, "clamp_kernel_hbm_frac": clamp_gbs / PEAK_HBM_GBS if clamp_gbs else None,
    }
    blk.update(extra)
    return blk


def augment_step(fe, wav, lens, out, device):
    """BASELINE config 3: ranges of configs/augmentation.json with the probabilities of
    noise / speed / shift / volume forced to 1.0, decisions drawn on the host with random.seed(0) in
    the reference's order, noise bank = 16 synthetic 10 s clips."""
    import random
    from wfe_amd.augment import AugmentPlanner
    from wfe_amd.frontend import NoiseBank
    bank = [(0.05 * np.random.default_rng(9000 + j).standard_normal(10 * SR)).astype(np.float32)
            for j in range(16)]
    planner = AugmentPlanner([
        {"type": "noise", "prob": 1.0, "params": {"min_snr_dB": 10, "max_snr_dB": 50}},
        {"type": "speed", "prob": 1.0, "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}},
        {"type": "shift", "prob": 1.0, "params": {"min_shift_ms": -5, "max_shift_ms": 5}},
        {"type": "volume", "prob": 1.0, "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}},
    ], noise_lengths=[len(c) for c in bank])
    rnd = random.Random(0)
    host_lens = lens.cpu().tolist()
    t_plan = time.perf_counter()
    plans = [planner.draw(n, SR, rnd)[0] for n in host_lens]
    packed = fe.pack_plans(plans)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t_plan
    noise = NoiseBank(bank, device)
    step = lambda: fe.augment_logmel(wav, lens, packed, noise, out=out)  # noqa: E731
    note = ("augmentation fused: noise(10..50 dB SNR) -> speed{0.9,1.0,1.1} -> shift[-5,5] ms -> "
            "volume[-15,15] dB, probabilities forced to 1.0, op lists resident in HBM "
            f"(host draw + pack of one batch: {t_plan * 1e3:.2f} ms, outside the timed region)")
    return step, note


def reverb_chain_step(fe, wav, lens, out, device):
    """Config 3's chain with the shipped file's reverb entry firing too (configs/augmentation.json order:
    noise -> reverb -> speed -> shift -> volume, probabilities forced to 1.0): the staged execution of the
    collator (wfe_amd/data.py _staged) -- [noise -> waveform], wfe_reverb, [speed -> shift -> volume + log-mel].
    Impulse responses: 8 synthetic rooms of 0.5 - 1.0 s (8 000 - 16 000 taps) at 16 kHz."""
    import random
    from wfe_amd.augment import AugmentPlanner, split_stages
    from wfe_amd.frontend import NoiseBank
    bank = [(0.05 * np.random.default_rng(9000 + j).standard_normal(10 * SR)).astype(np.float32) for j in range(16)]
    rirs = []
    for j in range(8):
        n = 8000 + 1000 * j + (1000 if j == 7 else 0)
        h = np.random.default_rng(9100 + j).standard_normal(n) * np.exp(-np.arange(n) / (0.2 * n))
        h[0] = 1.0
        rirs.append(h.astype(np.float32))
    planner = AugmentPlanner([
        {"type": "noise", "prob": 1.0, "params": {"min_snr_dB": 10, "max_snr_dB": 50}},
        {"type": "reverb", "prob": 1.0, "params": {}},
        {"type": "speed", "prob": 1.0, "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}},
        {"type": "shift", "prob": 1.0, "params": {"min_shift_ms": -5, "max_shift_ms": 5}},
        {"type": "volume", "prob": 1.0, "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}},
    ], noise_lengths=[len(c) for c in bank], n_rir_files=len(rirs))
    rnd = random.Random(0)
    host_lens = [int(n) for n in lens.cpu().tolist()]
    stages = [split_stages(planner.draw(n, SR, rnd)[0], n, SR, SR) for n in host_lens]
    assert all(len(st) == 2 and st[0]["post"][0] == "reverb" for st in stages)
    noise, rir_bank = NoiseBank(bank, device), NoiseBank(rirs, device)
    first = fe.pack_plans([st[0]["ops"] for st in stages])
    last = fe.pack_plans([st[1]["ops"] for st in stages])
    rir_index = [st[0]["post"][1] for st in stages]
    width = (max(host_lens) + 3) // 4 * 4

    def step():
        w, ln = fe.augment_waveform(wav, lens, first, noise, n_samples=width)
        w = fe.reverb(w, ln, rir_bank, rir_index, out_stride=width)
        fe.augment_logmel(w, ln, last, noise, out=out)
    note = ("noise(10..50 dB SNR) -> reverb(8 rooms, 0.5 - 1.0 s responses, FFT convolution) -> speed{0.9,1.0,1.1} -> "
            "shift[-5,5] ms -> volume[-15,15] dB, probabilities forced to 1.0; three launches per batch (chain -> "
            "waveform, wfe_reverb, chain + log-mel), op lists and clip banks resident in HBM")
    return step, note


def pcie_leg(fe, wav, lens, out, steps: int, warmup: int, barrier) -> dict:
    """Inputs start in pinned host memory: H2D copies on a second stream, double buffered, overlap the
    kernels (the PCIe-inclusive rate; never the headline value)."""
    host = [wav.cpu().pin_memory(), wav.cpu().pin_memory()]
    dev = [torch.empty_like(wav), torch.empty_like(wav)]
    copy_stream, ready, consumed = torch.cuda.Stream(), [torch.cuda.Event(), torch.cuda.Event()], \
        [torch.cuda.Event(), torch.cuda.Event()]

    def pipelined(k):
        main = torch.cuda.current_stream()
        for i in range(k + 1):
            if i < k:
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(consumed[i & 1])          # buffer free again
                    dev[i & 1].copy_(host[i & 1], non_blocking=True)
                    ready[i & 1].record(copy_stream)
            if i > 0:
                j = (i - 1) & 1
                main.wait_event(ready[j])
                fe.logmel(dev[j], lens, out=out)
                consumed[j].record(main)
    for e in consumed:
        e.record(torch.cuda.current_stream())
    pipelined(warmup)
    barrier()
    t0 = time.perf_counter()
    pipelined(steps)
    barrier()
    el = time.perf_counter() - t0
    batch = wav.shape[0]
    return {"value": batch * steps * CLIP_HOURS / el, "unit": "audio-hours/sec", "steps": steps,
            "ms_per_step": el / steps * 1e3,
            "h2d_GBps": batch * N_SAMPLES * wav.element_size() * steps / el / 1e9,
            "input": "pinned host int16 PCM" if wav.dtype == torch.int16 else "pinned host fp32"}


def to_pcm16(wav):
    return torch.clamp(torch.round(wav * 32768.0), -32768, 32767).to(torch.int16)


def shard_workload(args, fe, device, rank, world, dist, clips=None, quiet=False):
    """BASELINE config 4: N clips sharded data-parallel, no data-path collective."""
    from wfe_amd.sharded import run_sharded
    frames = N_SAMPLES // 160
    n_clips = clips or args.shard_clips
    if args.stub_transform:
        # plumbing test only (CPU, gloo): no front end, a stand-in transform; never a measurement
        base = torch.arange(args.batch, dtype=torch.float32).reshape(-1, 1).repeat(1, 320)
        pool = torch.cat([base, base])
        lens = torch.full((args.batch,), 320, dtype=torch.int32)

        def transform(wav, ln):
            return wav[:, :2].reshape(wav.shape[0], 1, 2) + 1.0
        sync = None
    else:
        # clip i = base clip i % batch, whatever the shard or batch start: the pool holds the base
        # clips twice, so that any batch [begin, begin + count) is a zero-copy view of it
        base = device_clips(args.batch, device, seed0=0)
        pool = torch.cat([base, base])
        del base
        lens = torch.full((args.batch,), N_SAMPLES, dtype=torch.int32, device=device)
        fe.set_output_dtype(getattr(torch, args.out_dtype))
        out = torch.empty((args.batch, fe.n_mel, frames), dtype=fe.out_dtype, device=device)

        def transform(wav, ln):
            return fe.logmel(wav, ln, out=out[:wav.shape[0]])
        sync = torch.cuda.synchronize

    def make_batch(begin, count):
        o = begin % args.batch
        return pool[o:o + count], lens[:count]

    warm = min(n_clips, 4 * args.batch * world)
    run_sharded(warm, args.batch, make_batch, transform, sync=sync, checksum_stride=1021)     # warm-up
    res = run_sharded(n_clips, args.batch, make_batch, transform, sync=sync, checksum_stride=1021)
    steps = max(1, -(-n_clips // (args.batch * world)))
    n_mel = fe.n_mel if fe is not None else args.n_mel
    line = {
        "metric": "log-mel audio-hours/sec", "value": res["items"] * CLIP_HOURS / res["seconds"],
        "unit": "audio-hours/sec", "n_gpus": world, "steps": steps,
        "warmup": max(1, -(-warm // (args.batch * world))), "ms_per_step": res["seconds"] / steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": DTYPE_NAME[args.out_dtype], "data": "stub" if args.stub_transform else "synthetic",
        "config": {"workload": f"whisper-large-v3 {n_mel}-mel front end, {n_clips} synthetic 30 s "
                               f"clips sharded contiguously over {world} GPU(s), batches of {args.batch}"
                               + (" [STUB TRANSFORM: plumbing test, not a measurement]" if args.stub_transform else "")
                               + (" [REHEARSAL: all ranks on GPU 0 over gloo, not a measurement]"
                                  if world > 1 and one_gpu_rehearsal() and not args.stub_transform else ""),
                   "parallelism": f"dp{world} (utterances sharded, no data-path collective)"},
        "clips_per_sec": res["items"] / res["seconds"], "items": res["items"], "checksum": res["checksum"]}
    if quiet:
        return line
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()
    return line


def also_legs(args, fe80, wav, lens, out, device, barrier) -> dict:
    """The other BASELINE legs that fit one GPU, as short timed passes (default N=1 run only)."""
    from wfe_amd.frontend import HipFrontEnd
    res = {}
    K, W, PRE = args.also_steps, 10, 50.0

    def leg(fe, step, n_mel, augment, note):
        el, k1, k2, n = timed_steps(fe, step, K, W, PRE, barrier)
        r = roofline_block(n_mel, args.batch, lens, k1, k2, n, augment, "full", True)
        return {"workload": note, "steps": K, "ms_per_step": el / K * 1e3,
                "value": args.batch * K * CLIP_HOURS / el, "unit": "audio-hours/sec",
                "kernel_ms_avg": r["kernel_ms_avg"], "clamp_kernel_ms_avg": r["clamp_kernel_ms_avg"],
                "frac": r["frac"], "traffic": r["traffic"], "hbm_floor_frac": r["hbm_floor_frac"],
                **{k: v for k, v in r.items() if k.endswith("_kernel_traffic")}}

    # BASELINE config 3: the augmentation chain fused into the same launch
    step, note = augment_step(fe80, wav, lens, out, device)
    res["config3"] = leg(fe80, step, 80, True, "BASELINE config 3: 80 mel, batch %d, %s" % (args.batch, note))
    # the same with the shipped file's reverb entry firing: wall time of the three launches only
    step, note = reverb_chain_step(fe80, wav, lens, out, device)
    for _ in range(3):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(max(10, K // 4)):
        step()
    barrier()
    el = time.perf_counter() - t0
    res["config3_reverb"] = {"workload": "config 3 + reverb: 80 mel, batch %d, %s" % (args.batch, note),
                             "steps": max(10, K // 4), "ms_per_step": el / max(10, K // 4) * 1e3,
                             "value": args.batch * max(10, K // 4) * CLIP_HOURS / el, "unit": "audio-hours/sec"}
    # PCIe-inclusive rates; fp32 is 8.7 ms per step: fewer steps
    res["pcie_f32"] = pcie_leg(fe80, wav, lens, out, max(10, K // 5), 3, barrier)
    pcm = to_pcm16(wav)
    res["pcie_pcm16"] = pcie_leg(fe80, pcm, lens, out, max(10, K // 4), 3, barrier)
    del pcm
    # 128 mel (whisper-large-v3): the batch step, then BASELINE config 4's 1-GPU leg (sharded clips)
    fe128 = HipFrontEnd(n_mel=128, device=device.index)
    out128 = torch.empty((args.batch, 128, N_SAMPLES // 160), dtype=torch.float32, device=device)
    res["mel128"] = leg(fe128, lambda: fe128.logmel(wav, lens, out=out128), 128, False,
                        "whisper-large-v3 128-mel front end, batch %d x 30 s, no augmentation, input resident in HBM" % args.batch)
    del out128
    sh = shard_workload(args, fe128, device, 0, 1, False, clips=args.also_shard_clips, quiet=True)
    res["config4_1gpu"] = {k: sh[k] for k in ("value", "unit", "steps", "ms_per_step", "items", "checksum")}
    res["config4_1gpu"]["workload"] = sh["config"]["workload"]
    fe128.close()
    return res


def main(argv=None) -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the first ~40 launches after an idle period run 10 % slower (clock ramp): the defaults time the
    # sustained rate; 220 steps of 0.7 ms are still well under a second
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-ms", type=float, default=100.0,
                    help="untimed steps for this long before the warm-up steps (GPU clock ramp); 0 disables")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--n-mel", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the extra BASELINE legs (config 3, 128 mel / config 4, PCIe) of the default N=1 run")
    ap.add_argument("--also-steps", type=int, default=100)
    ap.add_argument("--also-shard-clips", type=int, default=8192)
    ap.add_argument("--lengths", choices=["full", "speech"], default="full",
                    help="'speech': utterance lengths uniform in 1..15 s (seeded) instead of 30 s, still padded to "
                         "30 s -- what a real corpus looks like; throughput is still counted in 30 s windows")
    ap.add_argument("--from-host", action="store_true",
                    help="inputs start in pinned host memory: H2D copies on a second stream, double buffered, "
                         "overlap the kernels (the PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--pcm16", action="store_true",
                    help="input is 16-bit PCM (wfe_logmel_pcm16: converted on the GPU, half the bytes); the same "
                         "clips quantised to int16")
    ap.add_argument("--augment", action="store_true",
                    help="BASELINE config 3: the augmentation chain fused into the same launch.  Ranges of "
                         "configs/augmentation.json with the probabilities of noise / speed / shift / volume forced "
                         "to 1.0 (shipped: 0.5 / 0 / 0 / 0 and a placeholder noise dir), decisions drawn on the host "
                         "with random.seed(0) in the reference's order, noise bank = 16 synthetic 10 s clips")
    ap.add_argument("--out-dtype", choices=["float32", "float16", "bfloat16"], default="float32",
                    help="element type of the log-mel output (float32 = the reference's; the 16-bit types are "
                         "that result rounded to nearest even, SURVEY 8f N4).  Not the headline configuration")
    ap.add_argument("--shard-clips", type=int, default=0,
                    help="BASELINE config 4 instead of the default workload: this many clips in all, contiguous "
                         "shards per rank (wfe_amd.sharded), batches of --batch; use with --n-mel 128")
    ap.add_argument("--stub-transform", action="store_true",
                    help="plumbing test of the multi-rank launch on CPU (gloo) with a stand-in transform; only with "
                         "--shard-clips; the line it prints says so and is not a measurement")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)   # one of cpu_baseline's processes
    args = ap.parse_args(argv)

    if args.cpu_worker > 0:
        cpu_worker(args.n_mel, args.cpu_worker)
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, sys.argv[1:] if argv is None else argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: they must agree")
    dist = world > 1
    _heavy()
    if args.stub_transform:
        if not args.shard_clips:
            raise SystemExit("--stub-transform is a plumbing test of --shard-clips only")
        if dist:
            import torch.distributed as td
            td.init_process_group("gloo")
        shard_workload(args, None, torch.device("cpu"), rank, world, dist)
        return

    plain = not (args.shard_clips or args.from_host or args.augment or args.pcm16 or args.lengths != "full"
                 or args.out_dtype != "float32" or args.n_mel != 80)
    # the CPU baseline runs BEFORE this process initialises the GPU: its 8 worker processes are then
    # children of a GPU-free process
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not args.shard_clips and not args.from_host:
        if visible_gpu_count() == 0:                 # (sysfs, not HIP; None = unknown: the check below decides)
            raise SystemExit("bench.py needs an MI355X: the front end has no CPU path (0 GPU(s) visible)")
        cpu = cpu_baseline(args.n_mel)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the front end has no CPU path (0 GPU(s) visible)")
    rehearsal = dist and one_gpu_rehearsal()
    if rehearsal:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU (only {torch.cuda.device_count()} GPU(s) visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if dist:
        import torch.distributed as td
        if rehearsal:
            td.init_process_group("gloo")
        else:
            td.init_process_group("nccl", device_id=device)

    from wfe_amd.frontend import HipFrontEnd
    fe = HipFrontEnd(n_mel=args.n_mel, device=local_rank)
    if args.shard_clips:
        shard_workload(args, fe, device, rank, world, dist)
        return
    wav = device_clips(args.batch, device, seed0=rank * args.batch)
    lens = torch.full((args.batch,), N_SAMPLES, dtype=torch.int32, device=device)
    if args.lengths == "speech":
        rs = np.random.default_rng(99 + rank)
        lens = torch.from_numpy(rs.integers(SR, 15 * SR, size=args.batch).astype(np.int32)).to(device)
    fe.set_output_dtype(getattr(torch, args.out_dtype))
    out = torch.empty((args.batch, args.n_mel, N_SAMPLES // 160), dtype=fe.out_dtype, device=device)
    if args.pcm16:
        wav = to_pcm16(wav)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    if args.from_host:
        r = pcie_leg(fe, wav, lens, out, args.steps, args.warmup, barrier)
        if rank == 0:
            print(json.dumps({"metric": "log-mel audio-hours/sec (PCIe-inclusive, pinned %s host input)"
                                        % ("int16 PCM" if args.pcm16 else "fp32"),
                              "value": r["value"] * world, "unit": "audio-hours/sec",
                              "n_gpus": world, "steps": args.steps, "ms_per_step": r["ms_per_step"],
                              "h2d_GBps": r["h2d_GBps"]}), flush=True)
        return

    step = lambda: fe.logmel(wav, lens, out=out)                          # noqa: E731
    aug_note = "no augmentation"
    if args.augment:
        step, aug_note = augment_step(fe, wav, lens, out, device)

    elapsed, k1_ms, k2_ms, launches = timed_steps(fe, step, args.steps, args.warmup, args.prewarm_ms, barrier)

    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())

    checksum = float(out.double().sum().item())
    if rank == 0:
        clips = args.batch * args.steps * world
        line = {
            "metric": "log-mel audio-hours/sec",
            "value": clips * CLIP_HOURS / elapsed,
            "unit": "audio-hours/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE_NAME[args.out_dtype],
            "data": "synthetic",
            "config": {"workload": f"whisper-small {args.n_mel}-mel front end, batch {args.batch} synthetic 30 s "
                                   f"16 kHz clips per GPU, {aug_note}, input resident in HBM"
                                   + ("" if args.lengths == "full" else "; utterances 1..15 s padded to 30 s")
                                   + ("; 16-bit PCM input converted on the GPU" if args.pcm16 else "")
                                   + ("" if args.out_dtype == "float32" else f"; output rounded to {args.out_dtype}"),
                       "batch_per_gpu": args.batch, "n_mel": args.n_mel, "clip_seconds": 30,
                       "prewarm_ms": args.prewarm_ms,
                       "parallelism": f"dp{world} (utterances sharded, no data-path collective)"
                                      + (" [REHEARSAL: all ranks on GPU 0 over gloo, not a measurement]" if rehearsal else "")},
            "clips_per_sec": clips / elapsed,
            "roofline": roofline_block(args.n_mel, args.batch, lens, k1_ms, k2_ms, launches, args.augment, args.lengths,
                                       not args.pcm16 and args.out_dtype == "float32"),
            "checksum": checksum,
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if world == 1 and plain and not args.no_also:
            line["also"] = also_legs(args, fe, wav, lens, out, device, barrier)
        print(json.dumps(line), flush=True)
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
