#!/bin/bash
# round-4 GPU batch 1: strict occupancy variants, the guard's calibration sweep, the clock the path tracer really holds under K3,
# a guarded fuzz campaign, then the GPU test-suite.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
L=vulkan-compute-tests_amd/lib
MC_TIME_MATH=strict python tools/time_libs.py $L/libmc_compute.so $L/libmc_compute_exp_sw5r2.so $L/libmc_compute_exp_sw6r2.so > $out/r04_strict_occupancy.txt 2>&1 || exit 1
cat $out/r04_strict_occupancy.txt
python tools/enclosed_light_sweep.py > $out/r04_enclosed_light_sweep.txt 2>&1 || exit 1
tail -40 $out/r04_enclosed_light_sweep.txt
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/prof_r04_k3_clock -- python3 bench.py --config K3 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $out/prof_r04_k3_clock.log 2>&1 || exit 1
tail -3 $out/prof_r04_k3_clock.log
python tools/fuzz_fast.py --seconds 100 --seed 11 --enclose > $out/r04_fuzz_fast_enclose.log 2>&1; echo "fuzz rc $?"; tail -4 $out/r04_fuzz_fast_enclose.log
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $out/r04_gputest1.log 2>&1; echo "pytest rc $?"; tail -15 $out/r04_gputest1.log
