#!/bin/bash
# One counter pass over bench.py with a caller-chosen counter list (<= 8 SQ_, <= 2 GRBM_ per pass),
# means per kernel of wfe::.  Usage (GPU box, repo root):
#   bash tools/pmc_counters.sh <outdir> "SQ_INSTS_VALU SQ_INSTS_SALU ..." [bench args...]
set -u
export TMPDIR=/tmp
D=$1; C=$2; shift 2
mkdir -p "$D"
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$D/pmc" -- python3 bench.py --steps 5 --warmup 1 --prewarm-ms 50 --no-cpu-baseline --no-also "$@" > "$D/pmc.log" 2>&1 || { tail -n 5 "$D/pmc.log"; exit 1; }
python3 - "$D" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wfe::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, "launches", len(next(iter(v.values()))))
    for c in sorted(m):
        print(f"   {c:28s} {m[c]:.5g}")
PY
rm -rf "$D/pmc"
