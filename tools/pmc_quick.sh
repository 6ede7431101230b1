#!/bin/bash
# Quick counter pass over bench.py for one WFE_TUNE value: matrix-pipe busy, chip cycles, LDS conflicts.
# Usage (GPU box, repo root): bash tools/pmc_quick.sh <tune> <outdir> [bench args...]
set -u
export TMPDIR=/tmp
T=$1; D=$2; shift 2
mkdir -p "$D"
WFE_TUNE=$T timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$D/pmc" -- python3 bench.py --steps 5 --warmup 1 --prewarm-ms 50 --no-cpu-baseline "$@" > "$D/pmc.log" 2>&1 || exit 1
python3 - "$D" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wfe::frontend" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    g = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k, "launches", len(next(iter(v.values()))))
    for c in sorted(m):
        print(f"   {c:28s} {m[c]:.5g}")
    if g:
        print(f"   chip cycles/launch {g:.4g}  MFMA busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * g):.3f}")
PY
