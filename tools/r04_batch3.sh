#!/bin/bash
# round-4 GPU batch 3: K4 dispatch-order probe, the widened rows (n spheres, fast progressive ranges), the tests that changed.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
python tools/k4_order_probe.py 8 > $out/r04_k4_order_probe.txt 2>&1 || { tail -20 $out/r04_k4_order_probe.txt; exit 1; }
cat $out/r04_k4_order_probe.txt
python tools/bench_widened.py --only f3,f4box > $out/r04_bench_widened_a.jsonl 2> $out/r04_bench_widened_a.err || { tail -20 $out/r04_bench_widened_a.err; exit 1; }
cat $out/r04_bench_widened_a.jsonl
timeout -k 10 600 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_pool.py tests/test_gpu_edges.py tests/test_gpu_fast_scenes.py -m gpu -x -q > $out/r04_gputest3.log 2>&1; echo "pytest rc $?"; tail -12 $out/r04_gputest3.log; grep "spheres /\|enclosed\|unguarded" $out/r04_gputest3.log
