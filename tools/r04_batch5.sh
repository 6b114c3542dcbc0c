#!/bin/bash
# round-4 GPU batch 5: compiler-flag variants of the fast translation unit (A/B timing), then the rocprofv3 evidence of the final library.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
L=vulkan-compute-tests_amd/lib
python tools/time_libs.py $L/libmc_compute.so $L/libmc_compute_exp_cf1.so $L/libmc_compute_exp_cf2.so $L/libmc_compute_exp_cf3.so $L/libmc_compute_exp_cf4.so $L/libmc_compute_exp_cf5.so $L/libmc_compute_exp_cf6.so $L/libmc_compute_exp_cf7.so $L/libmc_compute_exp_cf8.so $L/libmc_compute_exp_cf9.so $L/libmc_compute_exp_cf10.so $L/libmc_compute.so > $out/r04_compiler_flags.txt 2>&1 || exit 1
cat $out/r04_compiler_flags.txt
bash tools/profile_gpu.sh r04_pt_fast > $out/r04_profile_fast.log 2>&1 || { tail $out/r04_profile_fast.log; exit 1; }
bash tools/profile_gpu.sh r04_pt_strict --math strict > $out/r04_profile_strict.log 2>&1 || { tail $out/r04_profile_strict.log; exit 1; }
python tools/summarize_prof.py r04_pt_fast $out/r04_pt_fast > /dev/null && python tools/summarize_prof.py r04_pt_strict $out/r04_pt_strict > /dev/null
cat $out/r04_pt_fast_kernel_stats.csv; cat $out/r04_pt_strict_kernel_stats.csv; grep -A12 derived $out/r04_pt_fast_pmc_summary.json
