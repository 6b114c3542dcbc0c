#!/bin/bash
# Every GPU measurement of round 4, as it was run through gpurun (one section per call; records under profiles/r04_*).
#   bash tools/r04_measurements.sh <section>      sections: probe occupancy guard k3clock k4order widened notrans flags profile bench fuzz rehearsal tests
# Experiment libraries are built first in the build container:  make -C vulkan-compute-tests_amd exp [EXP_TU=pathtrace_strict] EXP_NAME=<n> EXP_FLAGS="<flags>"
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
L=vulkan-compute-tests_amd/lib
case "$1" in
probe)      # where K2 loses against K3: time over spp and over image height            -> profiles/r04_pool_scaling_probe.txt
  python tools/pool_scaling_probe.py > $out/r04_pool_scaling_probe.txt 2>&1 && MC_TIME_MATH=strict python tools/pool_scaling_probe.py > $out/r04_pool_scaling_probe_strict.txt 2>&1 ;;
occupancy)  # exp libs w6hot (-DMC_PT_POOL_WAVES=6 -DMC_PT_POOL_HOT_VGPR=true), w7hot, w8; strict sw5r2 / sw6r2 (-DMC_PT_POOL_STRICT_WAVES=6 -DMC_PT_POOL_RESULT_BATCHES=2)
  python tools/time_libs.py $L/libmc_compute.so $L/libmc_compute_exp_w6hot.so $L/libmc_compute_exp_w7hot.so $L/libmc_compute_exp_w8.so > $out/r04_fast_occupancy.txt 2>&1
  MC_TIME_MATH=strict python tools/time_libs.py $L/libmc_compute.so $L/libmc_compute_exp_sw5r2.so $L/libmc_compute_exp_sw6r2.so > $out/r04_strict_occupancy.txt 2>&1 ;;
guard)      # calibration of the fast-math guard + a campaign aimed at its boundary     -> profiles/r04_enclosed_light_sweep.txt, r04_fuzz_fast_enclose.log
  python tools/enclosed_light_sweep.py > $out/r04_enclosed_light_sweep.txt 2>&1 && python tools/fuzz_fast.py --seconds 120 --seed 12 --enclose > $out/r04_fuzz_fast_enclose.log 2>&1 ;;
k3clock)    # the clock the path tracer itself holds under K3                           -> profiles/r04_k3_clock.txt
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/prof_r04_k3_clock -- python3 bench.py --config K3 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $out/prof_r04_k3_clock.log 2>&1 ;;
k4order)    # K4 at N = 8: natural / perfect / predicted dispatch orders                 -> profiles/r04_k4_order_probe.txt
  python tools/k4_order_probe.py 8 > $out/r04_k4_order_probe.txt 2>&1 ;;
widened)    # progressive ranges, boxes with 1..8 spheres              -> profiles/r04_bench_widened.jsonl
  python tools/bench_widened.py --only f3,f4box > $out/r04_bench_widened_a.jsonl 2> $out/r04_bench_widened_a.err ;;
notrans)    # exp lib notrans (-DMC_EXPERIMENT_NO_TRANS): cycles per VALU instruction with no transcendental in the stream -> profiles/r04_no_trans_pmc.txt
  bash tools/pmc_libs.sh r04_notrans $L/libmc_compute.so $L/libmc_compute_exp_notrans.so > $out/r04_pmc_notrans.log 2>&1 && python tools/time_libs.py $L/libmc_compute.so $L/libmc_compute_exp_notrans.so > $out/r04_time_notrans.txt 2>&1 ;;
flags)      # exp libs cf1..cf10 (profiles/r04_compiler_flags.txt names the options)
  python tools/time_libs.py $L/libmc_compute.so $L/libmc_compute_exp_cf*.so $L/libmc_compute.so > $out/r04_compiler_flags.txt 2>&1 ;;
profile)    # the rocprofv3 evidence behind bench.py's line                              -> profiles/r04_pt_{fast,strict}_{kernel_stats.csv,pmc_summary.json}
  bash tools/profile_gpu.sh r04_pt_fast > $out/r04_profile_fast.log 2>&1 && bash tools/profile_gpu.sh r04_pt_strict --math strict > $out/r04_profile_strict.log 2>&1 &&
  python tools/summarize_prof.py r04_pt_fast $out/r04_pt_fast > /dev/null && python tools/summarize_prof.py r04_pt_strict $out/r04_pt_strict > /dev/null ;;
bench)      # bench.py lines of every configuration                                      -> profiles/r04_bench_pt_fast.json, r04_bench_others.jsonl
  python bench.py > $out/r04_bench_k2.json 2> $out/r04_bench_k2.err && for c in K1 K1ds K3 K4; do python bench.py --config $c >> $out/r04_bench_others.jsonl 2>> $out/r04_bench_others.err || exit 1; done &&
  python bench.py --math strict --no-secondary >> $out/r04_bench_others.jsonl 2>> $out/r04_bench_others.err ;;
fuzz)       # randomised campaigns on the final build                                    -> profiles/r04_fuzz_parity.log, r04_fuzz_fast.log
  timeout -k 10 420 python tools/fuzz_parity.py --seconds 300 --seed 41 > $out/r04_fuzz_parity.log 2>&1; timeout -k 10 330 python tools/fuzz_fast.py --seconds 240 --seed 42 --enclose > $out/r04_fuzz_fast.log 2>&1 ;;
rehearsal)  # 4 ranks on one GPU (gloo) with --verify + the asynchronous exchange against a stub collective -> profiles/r04_rehearsal_4ranks.txt
  bash tools/r04_rehearsal.sh ;;
tests)
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r04_gputest.log 2>&1; tail -5 $out/r04_gputest.log ;;
*) echo "usage: $0 <section>"; exit 2 ;;
esac
