#!/bin/bash
# round-4 GPU batch 7: the GPU test-suite on the final build, then randomised campaigns — strict bit-identity (1..8 spheres, progressive
# ranges, row tiles, every kernel variant) and the fast kernels (finite, within forked samples; guarded scenes bit-identical to the oracle).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/r04_gputest7.log 2>&1; echo "pytest rc $?"; tail -6 $out/r04_gputest7.log; grep "overlapping" $out/r04_gputest7.log
timeout -k 10 420 python tools/fuzz_parity.py --seconds 300 --seed 41 > $out/r04_fuzz_parity.log 2>&1; echo "fuzz_parity rc $?"; tail -3 $out/r04_fuzz_parity.log
timeout -k 10 330 python tools/fuzz_fast.py --seconds 240 --seed 42 --enclose > $out/r04_fuzz_fast.log 2>&1; echo "fuzz_fast rc $?"; tail -3 $out/r04_fuzz_fast.log
MC_TIME_MATH=fast python tools/time_libs.py vulkan-compute-tests_amd/lib/libmc_compute.so
