#!/bin/bash
# The workloads of DESIGN.md's table, one bench.py process each (GPU box, repo root).
for args in "" "--n-mel 128" "--lengths speech" "--augment" "--augment --n-mel 128" "--shard-clips 100000 --n-mel 128" "--pcm16" "--out-dtype float16" "--from-host" "--from-host --pcm16"; do
  printf "[%s] " "$args"
  timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-also 2>&1 | grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*\|\"kernel_ms_avg\": [0-9.]*\|\"frac\": [0-9.]*" | tr '\n' ' ' || exit 1
  echo
done
