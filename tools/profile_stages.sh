#!/bin/bash
# Counter evidence for the stage kernels (wfe_resample, wfe_reverb): tools/stage_bench.py under rocprofv3, one run per
# counter pass, condensed by tools/summarize_profile.py.  Usage (GPU box, repo root): bash tools/profile_stages.sh <label> [stage_bench.py arguments]
# Leaves gpurun_out/summary_<label>.txt.
set -u
export TMPDIR=/tmp
L=${1:-stages}
shift || true
D=gpurun_out/prof_$L
mkdir -p "$D"
timeout -k 10 200 python3 -u tools/stage_bench.py "$@" > "$D/stage_bench.jsonl" 2> "$D/stage_bench.err" || exit 1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 tools/stage_bench.py "$@" > "$D/trace.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$D/pmc_sq" -- python3 tools/stage_bench.py "$@" > "$D/pmc_sq.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_fetch" -- python3 tools/stage_bench.py "$@" > "$D/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/pmc_write" -- python3 tools/stage_bench.py "$@" > "$D/pmc_write.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$D/pmc_lds" -- python3 tools/stage_bench.py "$@" > "$D/pmc_lds.log" 2>&1 || exit 1
{ echo "== tools/stage_bench.py (un-profiled run) =="; cat "$D/stage_bench.jsonl"; python3 tools/summarize_profile.py "$D" | sed -n '/rocprofv3 --kernel-trace/,$p'; } > "gpurun_out/summary_$L.txt" 2>&1
rm -rf "$D"
echo "profiled $L"
