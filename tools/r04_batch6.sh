#!/bin/bash
# round-4 GPU batch 6: the one-transcendental strict inversesqrt (exhaustive check + A/B timing), bench.py lines of every configuration.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
L=vulkan-compute-tests_amd/lib
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "short_forms or mc_math or strict" > $out/r04_gputest6a.log 2>&1; rc=$?; tail -5 $out/r04_gputest6a.log; [ $rc -eq 0 ] || exit 1
MC_TIME_MATH=strict python tools/time_libs.py $L/libmc_compute_exp_rsq2t.so $L/libmc_compute.so $L/libmc_compute_exp_rsq2t.so $L/libmc_compute.so > $out/r04_strict_rsqrt.txt 2>&1 || exit 1
cat $out/r04_strict_rsqrt.txt
python bench.py > $out/r04_bench_k2.json 2> $out/r04_bench_k2.err || { tail $out/r04_bench_k2.err; exit 1; }
cut -c1-1500 $out/r04_bench_k2.json
for c in K1 K1ds K3 K4; do python bench.py --config $c >> $out/r04_bench_others.jsonl 2>> $out/r04_bench_others.err || exit 1; done
python bench.py --math strict --no-secondary >> $out/r04_bench_others.jsonl 2>> $out/r04_bench_others.err || exit 1
cut -c1-400 $out/r04_bench_others.jsonl
