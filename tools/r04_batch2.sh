#!/bin/bash
# round-4 GPU batch 2: the refactored (1..8 spheres) kernels — timing against the occupancy variants, the test-suite, the recalibrated
# guard's sweep and a guarded fuzz campaign.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
L=vulkan-compute-tests_amd/lib
python tools/time_libs.py $L/libmc_compute.so $L/libmc_compute_exp_w6hot.so $L/libmc_compute_exp_w7hot.so $L/libmc_compute_exp_w8.so $L/libmc_compute.so > $out/r04_fast_occupancy.txt 2>&1 || exit 1
cat $out/r04_fast_occupancy.txt
MC_TIME_MATH=strict python tools/time_libs.py $L/libmc_compute.so > $out/r04_strict_time.txt 2>&1 || exit 1
cat $out/r04_strict_time.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r04_gputest2.log 2>&1; echo "pytest rc $?"; tail -15 $out/r04_gputest2.log
python tools/enclosed_light_sweep.py > $out/r04_enclosed_light_sweep.txt 2>&1 || exit 1
python tools/fuzz_fast.py --seconds 120 --seed 12 --enclose > $out/r04_fuzz_fast_enclose.log 2>&1; echo "fuzz rc $?"; tail -3 $out/r04_fuzz_fast_enclose.log
