#!/bin/bash
# Sweeps the two-slot scheduler's REGEN / SPEC thresholds on K2 (fast math); prints ms per launch.
cd "${GRAFT_REPO_ROOT:-.}"
for t in 8,8 12,8 16,8 16,16 24,8 24,16 32,16 32,24 48,16; do
  echo -n "thresholds $t: "
  MC_PT_PQ_THRESHOLDS=$t timeout -k 10 60 python tools/pt_sweep.py 2>/dev/null | grep "fast PQ" || exit 1
done
