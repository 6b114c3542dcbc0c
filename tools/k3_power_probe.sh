#!/bin/bash
# What the board does during a K3 step (3840x2560, 4096 spp: 2.4 s of back-to-back VALU work): rocm-smi power / clocks /
# temperature sampled twice a second while `bench.py --config K3` runs, and the same while the GPU idles.  GPU box.
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03_k3_power_probe.txt
{
  echo "== idle"; rocm-smi --showpower --showclocks --showtemp --showperflevel 2>&1 | grep -v "^$\|====" | head -30
  python bench.py --config K3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_k3_power_bench.json 2>/dev/null &
  pid=$!
  sleep 6      # imports + first step
  for i in $(seq 1 14); do
    echo "== under K3, sample $i"; rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|temperature (sensor edge)\|junction\|memory)" | head -8
    sleep 0.5
  done
  wait $pid
  echo "== bench line"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_k3_power_bench.json").read().strip().split("\n")[-1])
print("ms_per_step", d["ms_per_step"], "sclk_mhz_under_valu_load", d["config"]["sclk_mhz_under_valu_load"])
PY
  echo "== power cap"; rocm-smi --showmaxpower 2>&1 | grep -v "^$\|====" | head -6
} > $out 2>&1
cat $out | tail -60
