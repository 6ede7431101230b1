#!/bin/bash
# A/B of experiment builds of the library (make -C whisper-finetune_amd/csrc variant NAME=.. DEFS=..) through
# bench.py, one process per build.  Usage (GPU box): bash tools/ab_lib.sh "default k1 k2" "--augment" ""
L=$1; shift
for l in $L; do
  for args in "$@"; do
    printf "lib %-8s [%s] " "$l" "$args"
    if [ "$l" = default ]; then unset WFE_LIB_PATH; else export WFE_LIB_PATH=$PWD/tools/build/libwfe_hip_$l.so; fi
    timeout -k 10 120 python bench.py $args --steps 200 --warmup 50 --no-cpu-baseline --no-also 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|\"kernel_ms_avg\": [0-9.]*\|\"clamp_kernel_ms_avg\": [0-9.]*" | tr '\n' ' ' || exit 1
    echo
  done
done
