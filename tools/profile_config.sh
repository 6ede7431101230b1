#!/bin/bash
# One configuration: tools/profile_round.sh, then the small summaries only (the raw traces are tens of
# MiB and stay on the box).  Usage: bash tools/profile_config.sh <label> [bench.py arguments]
# Leaves gpurun_out/summary_<label>.txt and gpurun_out/traffic_<label>.json.
set -u
L=$1; shift
D=gpurun_out/prof_$L
bash tools/profile_round.sh "$D" "$@" > "gpurun_out/prof_$L.log" 2>&1 || { tail -n 5 "gpurun_out/prof_$L.log" "$D"/*.log; exit 1; }
python3 tools/summarize_profile.py "$D" > "gpurun_out/summary_$L.txt" 2>&1
NM=80; AUG=""; LEN=full
prev=""
for a in "$@"; do
  [ "$prev" = "--n-mel" ] && NM=$a
  [ "$prev" = "--lengths" ] && LEN=$a
  [ "$a" = "--augment" ] && AUG="--augment"
  prev=$a
done
cp profiles/traffic.json /tmp/traffic_before.json
python3 tools/update_traffic.py "$D" --n-mel "$NM" $AUG --lengths "$LEN" --source "round 3, profiles/round3_$L.txt" > "gpurun_out/traffic_$L.json" 2>&1
cp /tmp/traffic_before.json profiles/traffic.json
rm -rf "$D"
echo "profiled $L"
