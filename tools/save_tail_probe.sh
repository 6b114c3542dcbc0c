#!/bin/bash
# Where the tail of K4's streamed save goes (bin/mandelbrot --timing-json: png = png_join + png_assemble + png_write), both routes, 4 cold
# processes each.   bash tools/save_tail_probe.sh > gpurun_out/r06_save_tail_probe.txt        -> profiles/r06_save_tail_probe.txt
cd "${GRAFT_REPO_ROOT:-.}"
app=vulkan-compute-tests_amd/bin/mandelbrot
view="--width 7680 --height 5120 --max-iter 50000 --precision ds --centre -0.7436438870371587 0.13182590420531198 --scale 1e-08 6.666666666666667e-09"
echo "# $($app --width 8 --height 8 --quiet --out /tmp/_t.png | grep -c . ) lines from a tiny run (the binary works); K4 = $view"
for route in "" "--gpu-postprocess"; do
  for i in 1 2 3 4; do
    $app $view $route --quiet --timing-json --out /tmp/k4_tail.png | grep timing_ms | python3 -c "
import json,sys
t=json.loads(sys.stdin.read())['timing_ms']
print('route %-18s kernel %6.1f copy %5.1f  png %5.1f = join %5.1f + assemble %4.1f + write %4.1f   total %6.1f  bands %d' % ('${route:-host_buffer}', t['kernel'], t['copy'], t['png'], t['png_join'], t['png_assemble'], t['png_write'], t['total'], t['streamed_bands']))"
  done
done
ls -l /tmp/k4_tail.png | awk '{print "# file bytes", $5}'
