#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench command      -> gpurun_out/prof_<tag>_stats/
#   2. VALU issue / lane-utilisation counters (own pass)      -> gpurun_out/prof_<tag>_pmc_valu/
#   3. HBM write traffic (own pass; WRITE_SIZE is exact for 16-B stores)  -> gpurun_out/prof_<tag>_pmc_hbm/
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $out/prof_${tag}_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $out/prof_${tag}_pmc_valu -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/prof_${tag}_pmc_valu.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/prof_${tag}_pmc_hbm -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/prof_${tag}_pmc_hbm.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/prof_${tag}_pmc_mix -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/prof_${tag}_pmc_mix.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization --output-format csv -d $out/prof_${tag}_pmc_busy -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/prof_${tag}_pmc_busy.log 2>&1 || echo "derived VALUBusy pass failed (non-fatal)"
echo done
