#!/usr/bin/env python3
"""profiles/round*_rocprof_other_workloads.txt from the kernel-trace runs of
bench.py --augment / --n-mel 128 / --lengths speech (directory layout as written by
the commands in profiles' header)."""
import csv
import glob
import json
import sys

root = sys.argv[1]
out = ["Additional workloads: rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline <flags>` "
       "(MI355X, batch 256 x 30 s)", ""]
for w, flags in (("aug", "--augment   (BASELINE config 3)"), ("mel128", "--n-mel 128"),
                 ("speech", "--lengths speech   (utterances 1..15 s padded to 30 s)")):
    f = glob.glob(f"{root}/{w}/*/*kernel_stats.csv")[0]
    out.append(f"== bench.py {flags}")
    j = json.loads([l for l in open(f"{root}/{w}.log") if l.startswith("{")][-1])
    out.append(f'   bench line: value {j["value"]:.1f} {j["unit"]}, ms_per_step {j["ms_per_step"]:.4f}, kernel 1 '
               f'{j["roofline"]["kernel_ms_avg"]:.4f} ms, kernel 2 {j["roofline"]["clamp_kernel_ms_avg"]:.4f} ms (HIP events)')
    for r in csv.DictReader(open(f)):
        if "wfe::" in r["Name"]:
            out.append(f'   {r["Name"][:100]:100s} calls={r["Calls"]:>5s} avg_ns={float(r["AverageNs"]):>12.0f} '
                       f'min_ns={r["MinNs"]:>10s} max_ns={r["MaxNs"]:>10s}')
    out.append("")
print("\n".join(out))
