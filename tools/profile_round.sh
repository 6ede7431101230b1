#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline line is judged against.
# Usage (on the GPU box, from the repo root):  bash tools/profile_round.sh gpurun_out/prof_rN [bench.py arguments]
# e.g. ... gpurun_out/prof_aug --augment   |   ... gpurun_out/prof_128 --n-mel 128
# Each counter pass is its own run (rocprofv3 does not split counters over passes).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/${1:-gpurun_out/prof}
shift || true
mkdir -p "$D"
cd "$R" || exit 1
timeout -k 10 300 python3 -u bench.py "$@" > "$D/bench.json" 2> "$D/bench.err" || exit 1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 bench.py --no-cpu-baseline --no-also "$@" > "$D/trace.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$D/pmc_sq" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also "$@" > "$D/pmc_sq.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_fetch" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also "$@" > "$D/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/pmc_write" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also "$@" > "$D/pmc_write.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$D/pmc_lds" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also "$@" > "$D/pmc_lds.log" 2>&1 || exit 1
echo done
