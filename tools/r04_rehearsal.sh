#!/bin/bash
# 4-rank rehearsal of bench.py on ONE MI355X (MC_BENCH_BACKEND=gloo: the ranks share the GPU, the gather is staged through the host — timings are
# not measurements) with --verify, plus the single-GPU test of the ASYNCHRONOUS exchange branch against a stub collective.  -> gpurun_out/r04_rehearsal_4ranks.txt
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
r=$out/r04_rehearsal_4ranks.txt; : > $r
run() { MC_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 4 --steps 2 --warmup 1 --verify --no-cpu-baseline "${@:2}" 2> $out/r04_rehearsal.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']
        print(json.dumps({'workload': c['workload'], 'world_size': c['world_size'], 'tiling': c['tiling'], 'exchange': c['exchange'], 'exchange_async': c['exchange_async'], 'kernel': d['roofline']['kernel'], 'gather_bytes_per_rank': c['gather_bytes_per_rank'], 'verified_equal_to_single_gpu': c['verified_equal_to_single_gpu'], 'rows_per_rank': [x['rows'] for x in c['ranks']]}))
" >> $r || { tail -5 $out/r04_rehearsal.err; return 1; }; }
run 29611 --spp 64 && run 29612 --config K3 --spp 16 && run 29613 --config K4 --width 1536 --height 1040 || exit 1
python -m pytest tests/test_gpu_multi.py -m gpu -q -k "asynchronous_exchange" -rA 2>&1 | grep -E "PASSED|FAILED|passed|failed" >> $r
cat $r
