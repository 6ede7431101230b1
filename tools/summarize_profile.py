#!/usr/bin/env python3
"""Condenses a tools/profile_round.sh output directory into the small text summary
committed under profiles/ (kernel stats + per-launch counters of the wfe kernels)."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
out = []
try:
    line = [l for l in open(f"{d}/bench.json") if l.startswith("{")][-1]
    b = json.loads(line)
    out.append("== bench.py (un-profiled run) ==")
    out.append(json.dumps(b, indent=1))
except Exception as e:  # noqa: BLE001
    out.append(f"bench.json unreadable: {e}")
out.append("\n== rocprofv3 --kernel-trace --stats (bench.py --no-cpu-baseline) ==")
for f in glob.glob(f"{d}/trace/*/*kernel_stats.csv"):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 6 or "wfe::" in r["Name"]:
            out.append(f'{r["Name"][:70]:70s} calls={r["Calls"]:>5s} avg_ns={float(r["AverageNs"]):12.0f} '
                       f'min_ns={r["MinNs"]:>9s} max_ns={r["MaxNs"]:>9s} pct={r["Percentage"]}')
for name in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    fs = glob.glob(f"{d}/{name}/*/*counter_collection.csv")
    if not fs:
        continue
    out.append(f"\n== rocprofv3 --pmc pass {name}: mean per launch ==")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "wfe::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.append(k)
        for c, x in sorted(v.items()):
            out.append(f"    {c:32s} n={len(x):3d} mean={sum(x) / len(x):.6g}")
        if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            g = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"]) / 8
            m = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
            out.append(f"    -> chip cycles/launch {g:.4g}; MFMA pipe utilisation {m / (1024 * g):.3f} "
                       "(SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8))")
        if "FETCH_SIZE" in v:
            x = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
            out.append(f"    -> HBM read  {x * 1024 * 2 / 1e6:.1f} MB/launch (FETCH_SIZE KiB x 2: gfx950 wide-read correction)")
        if "WRITE_SIZE" in v:
            x = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
            out.append(f"    -> HBM write {x * 1024 / 1e6:.1f} MB/launch (WRITE_SIZE KiB)")
print("\n".join(out))
