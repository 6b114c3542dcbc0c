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
tests)
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r05_gputest.log 2>&1; rc=$?; tail -5 $out/r05_gputest.log; [ $rc -eq 0 ] || exit $rc ;;
*) echo "usage: $0 <section> ..."; exit 2 ;;
esac
done
