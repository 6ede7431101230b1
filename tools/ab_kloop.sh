for t in 0 8; do
  for l in default ab64 ab128 ab256 ab448; do
    printf "tune %s lib %-8s " "$t" "$l"
    if [ "$l" = default ]; then unset WFE_LIB_PATH; else export WFE_LIB_PATH=$PWD/tools/build/libwfe_hip_$l.so; fi
    WFE_TUNE=$t timeout -k 10 120 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | grep -o "\"kernel_ms_avg\": [0-9.]*" | head -n 1 || exit 1
  done
done
