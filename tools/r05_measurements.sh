#!/bin/bash
# Every GPU measurement of round 5, as it was run through gpurun (one or more sections per call; records under profiles/r05_*).
#   bash tools/r05_measurements.sh <section> [<section> ...]
# Experiment libraries are built first in the build container:  make -C vulkan-compute-tests_amd exp EXP_NAME=<n> EXP_FLAGS="<flags>"
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
L=vulkan-compute-tests_amd/lib
for section in "$@"; do
case "$section" in
d2h)        # how the storage buffer should cross PCIe: pinned / pageable / registered / staged          -> profiles/r05_d2h_probe.txt
  timeout -k 10 300 tools/bin/d2h_probe > $out/r05_d2h_probe.txt 2>&1 || exit 1 ;;
census)     # fork census: exp libs fc_* (one fast-math shortcut switched off each; tools/fork_census.py) -> profiles/r05_fork_census.txt
  timeout -k 10 900 python tools/fork_census.py $L/libmc_compute.so $L/libmc_compute_exp_fc_*.so > $out/r05_fork_census.txt 2>&1 || exit 1 ;;
census2)    # what a more careful fast tier costs and buys: exp libs fc_c0 (no contraction), fc_short (division / sqrt / rsq rounded as the
            # reference rounds them: the strict short forms), fc_careful (both), on the test's scenes + specular-heavy ones -> profiles/r05_fork_census_careful.txt
  timeout -k 10 1100 python tools/fork_census.py --time --scenes "ref,4:1,5:2,6:1,7:1,8:1,8:3,8:1:7:spec,8:1:8:spec,6:1:9:spec,4:1:5:spec" \
     $L/libmc_compute.so $L/libmc_compute_exp_fc_c0.so $L/libmc_compute_exp_fc_short.so $L/libmc_compute_exp_fc_careful.so > $out/r05_fork_census_careful.txt 2>&1 || exit 1 ;;
bias)       # the sign-consistent mean shift, per sample: smooth part against forks                        -> profiles/r05_fork_bias.txt
  : > $out/r05_fork_bias.txt
  for sc in 8:1 6:1 ref; do timeout -k 10 600 python tools/fork_bias.py --scene $sc --samples 64 $L/libmc_compute.so $L/libmc_compute_exp_fc_c0.so \
     $L/libmc_compute_exp_fc_careful.so $L/libmc_compute_exp_fc_all.so >> $out/r05_fork_bias.txt 2>&1 || exit 1; done ;;
bias2)      # which identity of exact arithmetic carries the careful build's fork asymmetry (more samples lose than gain): exp libs fc_car_*
            # = fc_careful with directions re-normalised / the reference's det order / the root-form shadow test / all of them   -> profiles/r05_fork_bias_identities.txt
  timeout -k 10 900 python tools/fork_bias.py --scene 8:1 --samples 96 $L/libmc_compute_exp_fc_careful.so $L/libmc_compute_exp_fc_car_renorm.so \
     $L/libmc_compute_exp_fc_car_occ0.so $L/libmc_compute_exp_fc_car_renorm_occ0.so $L/libmc_compute_exp_fc_car_nodisj.so $L/libmc_compute_exp_fc_car_ids.so \
     > $out/r05_fork_bias_identities.txt 2>&1 || exit 1 ;;
tiers)      # the shipped tiers on the test's scenes, specular-heavy ones and two generic scenes: the request as a caller makes it (fast),
            # the fast tier forced (tier1), the careful tier; with kernel times                                -> profiles/r05_fast_tiers.txt
  timeout -k 10 1100 python tools/fork_census.py --time --modes fast,tier1,careful --scenes "ref,4:1,5:2,6:1,7:1,8:1,8:3,8:1:7:spec,8:1:8:spec,6:1:9:spec,g:6:12:2,g:6:40:3" \
     $L/libmc_compute.so > $out/r05_fast_tiers.txt 2>&1 || exit 1 ;;
bench)      # bench.py's default line (with the end_to_end block) and the other configurations            -> profiles/r05_bench_*.json
  python bench.py > $out/r05_bench_k2.json 2> $out/r05_bench_k2.err || { tail -5 $out/r05_bench_k2.err; exit 1; } ;;
benchall)
  : > $out/r05_bench_others.jsonl; for c in K1 K1ds K3 K4; do python bench.py --config $c >> $out/r05_bench_others.jsonl 2>> $out/r05_bench_others.err || exit 1; done &&
  python bench.py --math strict --no-secondary --no-end-to-end >> $out/r05_bench_others.jsonl 2>> $out/r05_bench_others.err ;;
wall)       # a light pushed up to / through a diffuse WALL (the guard compares lights with spheres only) -> profiles/r05_light_at_wall_sweep.txt
  timeout -k 10 600 python tools/light_at_wall_sweep.py > $out/r05_light_at_wall_sweep.txt 2>&1 || exit 1 ;;
e2e)        # where an app's wall time goes (tools/end_to_end.py: bin/pathtracer, bin/mandelbrot --timing-json, both routes, K2 / K1 / K4) -> profiles/r05_end_to_end.txt
  timeout -k 10 600 python tools/end_to_end.py > $out/r05_end_to_end.txt 2>&1 || exit 1 ;;
profile)    # the rocprofv3 evidence behind bench.py's lines, stamped with the build id                    -> profiles/r05_*_{kernel_stats.csv,pmc_summary.json}
  bash tools/profile_gpu.sh r05_pt_fast --no-end-to-end > $out/r05_profile_fast.log 2>&1 && python tools/summarize_prof.py r05_pt_fast $out/r05_pt_fast > /dev/null &&
  bash tools/profile_gpu.sh r05_pt_strict --math strict --no-end-to-end > $out/r05_profile_strict.log 2>&1 && python tools/summarize_prof.py r05_pt_strict $out/r05_pt_strict > /dev/null ;;
profile2)   # K1 / K1ds (and K4's kernel) re-profiled on the round-5 build: no round-3 figure behind a round-5 line
  bash tools/profile_gpu.sh r05_mandel --config K1 > $out/r05_profile_mandel.log 2>&1 && python tools/summarize_prof.py r05_mandel $out/r05_mandel > /dev/null &&
  bash tools/profile_gpu.sh r05_mandel_ds --config K1ds > $out/r05_profile_mandel_ds.log 2>&1 && python tools/summarize_prof.py r05_mandel_ds $out/r05_mandel_ds > /dev/null &&
  bash tools/profile_gpu.sh r05_k4 --config K4 --steps 2 > $out/r05_profile_k4.log 2>&1 && python tools/summarize_prof.py r05_k4 $out/r05_k4 > /dev/null ;;
fuzz)       # randomised campaigns on the final build: strict parity, fast scenes incl. 5 - 8 spheres weighted up (the careful tier) -> profiles/r05_fuzz_*.log
  timeout -k 10 330 python tools/fuzz_parity.py --seconds 240 --seed 51 > $out/r05_fuzz_parity.log 2>&1; r1=$?
  timeout -k 10 330 python tools/fuzz_fast.py --seconds 240 --seed 52 --enclose --many > $out/r05_fuzz_fast.log 2>&1; r2=$?
  tail -2 $out/r05_fuzz_parity.log $out/r05_fuzz_fast.log; [ $r1 -eq 0 ] && [ $r2 -eq 0 ] || exit 1 ;;
threshold)  # is "careful from five spheres on" safe?  the FAST tier on eight random boxes with 4 and eight with 3 spheres (any materials, all-specular ones too) -> profiles/r05_fast_tier_4_spheres.txt
  timeout -k 10 1100 python tools/fork_census.py --modes tier1 --scenes "4:1:201,4:1:202,4:1:203,4:2:204,4:1:205:spec,4:1:206:spec,4:1:207:spec,4:2:208:spec,3:1:211,3:1:212,3:1:213:spec,3:1:214:spec,4:1:215,4:1:216,4:1:217:spec,4:1:218:spec" \
     $L/libmc_compute.so > $out/r05_fast_tier_4_spheres.txt 2>&1 || exit 1 ;;
profile3)   # K3's launch (3840 x 2560 x 4096, 2 s) under the counters too: its own clock and instruction mix
  bash tools/profile_gpu.sh r05_k3 --config K3 --steps 1 --warmup 1 > $out/r05_profile_k3.log 2>&1 && python tools/summarize_prof.py r05_k3 $out/r05_k3 > /dev/null ;;
tests)
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r05_gputest.log 2>&1; rc=$?; tail -5 $out/r05_gputest.log; [ $rc -eq 0 ] || exit $rc ;;
*) echo "usage: $0 <section> ..."; exit 2 ;;
esac
done
