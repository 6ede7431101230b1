#!/bin/bash
# A/B of WFE_TUNE values through bench.py, one process per value.
# Usage (GPU box): bash tools/ab_env.sh "0 2 6" "--augment" ["--n-mel 128" ...]
T=$1; shift
for t in $T; do
  for args in "$@"; do
    printf "tune %s [%s] " "$t" "$args"
    WFE_TUNE=$t timeout -k 10 120 python bench.py $args --steps 200 --warmup 50 --no-cpu-baseline --no-also 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|\"kernel_ms_avg\": [0-9.]*\|\"clamp_kernel_ms_avg\": [0-9.]*" | tr '\n' ' ' || exit 1
    echo
  done
done
