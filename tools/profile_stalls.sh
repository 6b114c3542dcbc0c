#!/bin/bash
# Stall-class counters of the path tracer kernel (one --pmc pass): parked (s_waitcnt/barrier), issue stalls, scalar / LDS /
# branch / SMEM activity.  Usage: tools/profile_stalls.sh <tag> [env assignments are inherited]
set -o pipefail
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $out/prof_${tag}_stall1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/prof_${tag}_stall1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VSKIPPED SQ_INST_CYCLES_SALU --output-format csv -d $out/prof_${tag}_stall2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $out/prof_${tag}_stall2.log 2>&1 || exit 1
echo done
