#!/bin/bash
# 4-rank rehearsal of the PLAIN driver command on ONE MI355X (round 5): `python bench.py --gpus 4 ...` with no launcher around it starts
# torch.distributed.run itself as a child process (bench.py: spawn_ranks); MC_BENCH_BACKEND=gloo: the ranks share the GPU and the gather is
# staged through the host — timings are not measurements; --verify: the gathered image must equal the single-GPU render bit for bit.
#   -> gpurun_out/r05_rehearsal_4ranks.txt
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out; mkdir -p $out
r=$out/r05_rehearsal_4ranks.txt; : > $r
run() { MC_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 4 --steps 2 --warmup 1 --verify --no-cpu-baseline "$@" 2> $out/r05_rehearsal.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']
        print(json.dumps({'command': 'python bench.py --gpus 4 (plain: ranks started by bench.py itself)', 'workload': c['workload'], 'world_size': c['world_size'], 'world_size_seen': [x['world_size_seen'] for x in c['ranks']], 'tiling': c['tiling'], 'exchange': c['exchange'], 'kernel': d['roofline']['kernel'], 'gather_bytes_per_rank': c['gather_bytes_per_rank'], 'verified_equal_to_single_gpu': c['verified_equal_to_single_gpu'], 'rows_per_rank': [x['rows'] for x in c['ranks']]}))
" >> $r || { tail -5 $out/r05_rehearsal.err; return 1; }; }
run --spp 64 && run --config K3 --spp 16 && run --config K4 --width 1536 --height 1040 || exit 1
cat $r
