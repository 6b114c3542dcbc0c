#!/bin/bash
# rocprofv3 counters for one K2 launch of the regroup kernel vs the rounds kernel (tools/rg_time.py), one pass per counter set.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
out=gpurun_out; tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_stats -- python3 tools/rg_time.py "$@" > $out/prof_${tag}_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $out/prof_${tag}_pmc_valu -- python3 tools/rg_time.py "$@" > $out/prof_${tag}_pmc_valu.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $out/prof_${tag}_pmc_mix -- python3 tools/rg_time.py "$@" > $out/prof_${tag}_pmc_mix.log 2>&1 || exit 1
echo done
