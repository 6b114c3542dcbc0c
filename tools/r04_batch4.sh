#!/bin/bash
# round-4 GPU batch 4: cycles per VALU instruction with and without transcendentals in the stream (PMC), the widened rows again
# (8 spheres fixed), the whole GPU test-suite.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
L=vulkan-compute-tests_amd/lib
bash tools/pmc_libs.sh r04_notrans $L/libmc_compute.so $L/libmc_compute_exp_notrans.so > $out/r04_pmc_notrans.log 2>&1 || { tail -20 $out/r04_pmc_notrans.log; exit 1; }
cat $out/pmc_r04_notrans.txt
python tools/time_libs.py $L/libmc_compute.so $L/libmc_compute_exp_notrans.so > $out/r04_time_notrans.txt 2>&1 || exit 1
cat $out/r04_time_notrans.txt
python tools/bench_widened.py --only f3,f4box > $out/r04_bench_widened_a.jsonl 2> $out/r04_bench_widened_a.err || { tail -20 $out/r04_bench_widened_a.err; exit 1; }
grep f4box $out/r04_bench_widened_a.jsonl | cut -c1-330
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r04_gputest4.log 2>&1; echo "pytest rc $?"; tail -12 $out/r04_gputest4.log; grep "spheres /\|enclosed\|unguarded" $out/r04_gputest4.log
