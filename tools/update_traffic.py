#!/usr/bin/env python3
"""Folds the FETCH_SIZE / WRITE_SIZE passes of a tools/profile_round.sh directory into
profiles/traffic.json (one entry per bench configuration; bench.py looks its own up).
    python tools/update_traffic.py gpurun_out/prof_128 --n-mel 128 --source "round 2, profiles/round2_128mel.txt"
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies the 128-byte requests of wide coalesced
reads at 64 bytes); both counters are in KiB."""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--n-mel", type=int, default=80)
ap.add_argument("--augment", action="store_true")
ap.add_argument("--lengths", default="full")
ap.add_argument("--source", default="")
args = ap.parse_args()


def mean_per_kernel(name, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{args.dir}/{name}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "wfe::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = mean_per_kernel("pmc_fetch", "FETCH_SIZE"), mean_per_kernel("pmc_write", "WRITE_SIZE")
entry = {"batch": args.batch, "n_mel": args.n_mel, "augment": bool(args.augment), "lengths": args.lengths,
         "source": args.source or args.dir}
for k in sorted(set(fetch) | set(write)):
    entry[k] = {"fetch_bytes": int(fetch.get(k, 0) * 1024 * 2), "write_bytes": int(write.get(k, 0) * 1024)}
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path))
key = (entry["batch"], entry["n_mel"], entry["augment"], entry["lengths"])
tj["entries"] = [e for e in tj["entries"] if (e["batch"], e["n_mel"], bool(e["augment"]), e.get("lengths", "full")) != key]
tj["entries"].append(entry)
json.dump(tj, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
