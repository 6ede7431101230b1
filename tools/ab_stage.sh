#!/bin/bash
# A/B of experiment builds of the library through tools/stage_bench.py (resampler, or "reverb <taps ...>"), one process per build.
# Usage (GPU box): bash tools/ab_stage.sh "default rs1 rs2" 44100 22050
L=$1; shift
for l in $L; do
  if [ "$l" = default ]; then unset WFE_LIB_PATH; else export WFE_LIB_PATH=$PWD/tools/build/libwfe_hip_$l.so; fi
  timeout -k 10 120 python tools/stage_bench.py "$@" 2>&1 | grep -o "\"from\": [0-9]*\|\"taps\": [0-9]*\|\"ms\": [0-9.]*" | tr '\n' ' ' | sed "s/^/lib $l: /" || exit 1
  echo
done
