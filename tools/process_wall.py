#!/usr/bin/env python3
"""What a single-GPU process pays for RCCL being a LINK dependency of the library (rounds 1-5) rather than loaded on demand (round 6:
multi.hip dlopens librccl.so.1 in mc_multi_create, for more than one device only).  The apps' own `total` starts in main(); mapping
and relocating a 573 MB shared object, and the HIP runtime registering its device code, happen before and inside the first HIP call.
Wall time of the whole process from fork to exit, best and median of N cold runs of `bin/pathtracer 16 64` (a 96 x 64 x 16 render: start-up
is all there is), as shipped and with LD_PRELOAD=librccl.so.1 — which is what the old link line amounted to.
    GPU box:  python tools/process_wall.py > gpurun_out/r06_process_wall.txt"""
import os
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "vulkan-compute-tests_amd", "bin", "pathtracer")
RCCL = "/opt/rocm/lib/librccl.so.1"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 7

with tempfile.TemporaryDirectory() as tmp:
    for label, env in (("as shipped (RCCL loaded on demand)", {}), ("LD_PRELOAD=librccl.so.1 (the rounds 1-5 link line)", {"LD_PRELOAD": RCCL}),
                       ("as shipped, again", {})):
        walls, inits = [], []
        for _ in range(N):
            t = time.perf_counter()
            r = subprocess.run([APP, "16", "64", "--quiet", "--timing-json", "--out", os.path.join(tmp, "o.png")], capture_output=True, text=True,
                               env=dict(os.environ, **env))
            walls.append((time.perf_counter() - t) * 1e3)
            if r.returncode != 0:
                print(label, "FAILED", r.stdout[-300:], r.stderr[-300:])
                break
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"timing_ms"')][0]
            import json
            inits.append(json.loads(line)["timing_ms"]["init"])
        else:
            print(f"{label:52s} process wall: best {min(walls):7.1f} ms, median {statistics.median(walls):7.1f} ms;  init() inside it: best "
                  f"{min(inits):6.1f} ms, median {statistics.median(inits):6.1f} ms   ({N} runs)", flush=True)
