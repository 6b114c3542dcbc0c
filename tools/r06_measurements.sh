#!/bin/bash
# Every GPU measurement of round 6, as it was run through gpurun (one or more sections per call; records under profiles/r06_*).
#   bash tools/r06_measurements.sh <section> [<section> ...]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
for section in "$@"; do
case "$section" in
timeline)   # where a cold process spends its first 100 ms, what overlaps (tools/cold_timeline.cpp)        -> profiles/r06_cold_timeline.txt
  for m in seq alloc warm both seq both; do tools/bin/cold_timeline $m; done > $out/r06_cold_timeline.txt 2>&1 || exit 1 ;;
hostmem)    # the storage buffer's host side: hipHostMalloc / huge pages + parallel touch + register / pageable -> profiles/r06_hostmem_probe.txt
  timeout -k 10 300 tools/bin/cold_timeline hostmem > $out/r06_hostmem_probe.txt 2>&1 || exit 1 ;;
wall)       # process wall time with RCCL loaded on demand / preloaded (the rounds 1-5 link line)           -> profiles/r06_process_wall.txt
  timeout -k 10 300 python tools/process_wall.py > $out/r06_process_wall.txt 2>&1 || exit 1 ;;
e2e)        # the apps as cold processes, overlapped start against --serial-start, both routes, K2 / K1 / K4 -> profiles/r06_end_to_end.txt
  timeout -k 10 900 python tools/end_to_end.py > $out/r06_end_to_end.txt 2>&1 || exit 1 ;;
bench)      # bench.py's default line (with end_to_end, app_default, cpu_baseline)                          -> profiles/r06_bench_k2.json
  python bench.py > $out/r06_bench_k2.json 2> $out/r06_bench_k2.err || { tail -5 $out/r06_bench_k2.err; exit 1; } ;;
benchall)   # the other configurations on one GPU                                                           -> profiles/r06_bench_others.jsonl
  : > $out/r06_bench_others.jsonl; for c in K1 K1ds K3 K4; do python bench.py --config $c >> $out/r06_bench_others.jsonl 2>> $out/r06_bench_others.err || exit 1; done &&
  python bench.py --math strict --no-secondary --no-end-to-end >> $out/r06_bench_others.jsonl 2>> $out/r06_bench_others.err &&
  python bench.py --math careful --no-secondary --no-end-to-end --no-cpu-baseline >> $out/r06_bench_others.jsonl 2>> $out/r06_bench_others.err ;;
rehearsal)  # the driver's PLAIN multi-GPU command, 2 / 4 / 8 ranks sharing the one GPU (gloo: timings are not measurements): the `multi`
            # block with K3 and K4 at full size, per-rank evidence, bit-equality, retention                   -> profiles/r06_rehearsal_plain.jsonl
  : > $out/r06_rehearsal_plain.jsonl
  for n in 2 4; do MC_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus $n >> $out/r06_rehearsal_plain.jsonl 2>> $out/r06_rehearsal_plain.err || { tail -5 $out/r06_rehearsal_plain.err; exit 1; }; done ;;
profile)    # the rocprofv3 evidence behind bench.py's K2 lines, stamped with the build id                  -> profiles/r06_pt_{fast,strict}_*
  bash tools/profile_gpu.sh r06_pt_fast --no-end-to-end > $out/r06_profile_fast.log 2>&1 && python tools/summarize_prof.py r06_pt_fast $out/r06_pt_fast > /dev/null &&
  bash tools/profile_gpu.sh r06_pt_strict --math strict --no-end-to-end > $out/r06_profile_strict.log 2>&1 && python tools/summarize_prof.py r06_pt_strict $out/r06_pt_strict > /dev/null ;;
profile2)   # K1 / K1ds / K4 on the round-6 build                                                           -> profiles/r06_{mandel,mandel_ds,k4}_*
  bash tools/profile_gpu.sh r06_mandel --config K1 > $out/r06_profile_mandel.log 2>&1 && python tools/summarize_prof.py r06_mandel $out/r06_mandel > /dev/null &&
  bash tools/profile_gpu.sh r06_mandel_ds --config K1ds > $out/r06_profile_mandel_ds.log 2>&1 && python tools/summarize_prof.py r06_mandel_ds $out/r06_mandel_ds > /dev/null &&
  bash tools/profile_gpu.sh r06_k4 --config K4 --steps 2 > $out/r06_profile_k4.log 2>&1 && python tools/summarize_prof.py r06_k4 $out/r06_k4 > /dev/null ;;
profile3)   # K3's launch (3840 x 2560 x 4096 spp, 2 s) under the counters                                   -> profiles/r06_k3_*
  bash tools/profile_gpu.sh r06_k3 --config K3 --steps 1 --warmup 1 > $out/r06_profile_k3.log 2>&1 && python tools/summarize_prof.py r06_k3 $out/r06_k3 > /dev/null ;;
fuzz)       # randomised campaigns on the final build                                                       -> profiles/r06_fuzz_*.log
  timeout -k 10 330 python tools/fuzz_parity.py --seconds 240 --seed 61 > $out/r06_fuzz_parity.log 2>&1; r1=$?
  timeout -k 10 330 python tools/fuzz_fast.py --seconds 240 --seed 62 --enclose --many > $out/r06_fuzz_fast.log 2>&1; r2=$?
  tail -n 2 $out/r06_fuzz_parity.log; tail -n 2 $out/r06_fuzz_fast.log; [ $r1 -eq 0 ] && [ $r2 -eq 0 ] || exit 1 ;;
fuzz2)      # the same, 500 s each, other seeds, on the build with the specular tier rule                     -> profiles/r06_fuzz_*_c.log
  timeout -k 10 560 python tools/fuzz_parity.py --seconds 500 --seed 65 > $out/r06_fuzz_parity_c.log 2>&1; r1=$?
  timeout -k 10 560 python tools/fuzz_fast.py --seconds 500 --seed 66 --enclose --many > $out/r06_fuzz_fast_c.log 2>&1; r2=$?
  tail -n 2 $out/r06_fuzz_parity_c.log; tail -n 2 $out/r06_fuzz_fast_c.log; [ $r1 -eq 0 ] && [ $r2 -eq 0 ] || exit 1 ;;
threshold)  # the fast tier's margin at FOUR spheres on a larger sample — the fast tier forced on 32 more random boxes with four
            # spheres (one or two lights; half of them all-specular)                                           -> profiles/r06_fast_tier_4_spheres.txt
  timeout -k 10 1100 python tools/fork_census.py --modes tier1 --scenes "4:1:301:spec,4:1:302,4:1:303:spec,4:1:304,4:1:305:spec,4:1:306,4:1:307:spec,4:1:308,4:1:309:spec,4:1:310,4:1:311:spec,4:1:312,4:1:313:spec,4:1:314,4:1:315:spec,4:1:316,4:1:317:spec,4:1:318,4:1:319:spec,4:1:320,4:1:321:spec,4:1:322,4:1:323:spec,4:1:324,4:2:331:spec,4:2:332,4:2:333:spec,4:2:334,4:2:335:spec,4:2:336,4:2:337:spec,4:2:338" \
     vulkan-compute-tests_amd/lib/libmc_compute.so > $out/r06_fast_tier_4_spheres.txt 2>&1 || exit 1 ;;
mirrors)    # the mirror rule: the 36 jittered three-sphere rooms it was set on (seed 7), as a caller requests them — the tier that ran, and the
            # fast tier forced beside every promoted scene                                                  -> profiles/r06_fast_tolerance_scenes.txt
  timeout -k 10 1100 python tools/fast_tolerance_scenes.py --scenes 36 --spp 500 --seed 7 > $out/r06_fast_tolerance_scenes.txt 2>&1 || exit 1
  tail -n 3 $out/r06_fast_tolerance_scenes.txt ;;
mirrors2)   # ... 48 more (seed 8: the specular-wall half of the rule was set here)                         -> profiles/r06_fast_tolerance_scenes_seed8.txt
  timeout -k 10 1100 python tools/fast_tolerance_scenes.py --scenes 48 --spp 500 --seed 8 > $out/r06_fast_tolerance_scenes_seed8.txt 2>&1 || exit 1
  tail -n 3 $out/r06_fast_tolerance_scenes_seed8.txt ;;
mirrors3)   # ... and 64 rooms the rule has not seen (seed 9)                                               -> profiles/r06_fast_tolerance_scenes_validation.txt
  timeout -k 10 1100 python tools/fast_tolerance_scenes.py --scenes 64 --spp 500 --seed 9 > $out/r06_fast_tolerance_scenes_validation.txt 2>&1 || exit 1
  tail -n 3 $out/r06_fast_tolerance_scenes_validation.txt ;;
mirrors4)   # ... 64 more unseen rooms (seed 10), the rule unchanged whatever they show                         -> profiles/r06_fast_tolerance_scenes_validation2.txt
  timeout -k 10 1100 python tools/fast_tolerance_scenes.py --scenes 64 --spp 500 --seed 10 > $out/r06_fast_tolerance_scenes_validation2.txt 2>&1 || exit 1
  tail -n 3 $out/r06_fast_tolerance_scenes_validation2.txt ;;
mirrors5)   # ... and 64 more (seed 11)                                                                          -> profiles/r06_fast_tolerance_scenes_validation3.txt
  timeout -k 10 1100 python tools/fast_tolerance_scenes.py --scenes 64 --spp 500 --seed 11 > $out/r06_fast_tolerance_scenes_validation3.txt 2>&1 || exit 1
  tail -n 3 $out/r06_fast_tolerance_scenes_validation3.txt ;;
tests)
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/r06_gputest.log 2>&1; rc=$?; tail -5 $out/r06_gputest.log; [ $rc -eq 0 ] || exit $rc ;;
*) echo "usage: $0 <section> ..."; exit 2 ;;
esac
done
