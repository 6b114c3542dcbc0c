"""Lane utilisation of the lane-regrouping path tracer per code region (diagnostic build: `make stats`):
how often a wave executes each region per 64 samples and with how many active lanes.
Usage (GPU box): python tools/rg_region_stats.py [spp]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MC_LIB_PATH", os.path.join(ROOT, "vulkan-compute-tests_amd", "lib", "libmc_compute_stats.so"))
import __graft_entry__ as entry  # noqa: E402

B = entry.load_package().bindings
L = B.lib()
L.mc_debug_pt_region_stats.argtypes = [C.c_void_p, C.c_void_p]
W, H = 900, 600
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
names = {9: "scheduler iteration", 11: "batch (spill + specular/camera refill)", 12: "idle spin", 0: "head: ray generation",
         3: "head: diffuse (NEE + shadow ray + bounce)", 5: "head: mirror", 6: "head: glass", 1: "intersect + prologue",
         2: "prologue (hit)", 10: "retire to the reorder ring"}
with B.Context(0) as ctx:
    ex, ln = np.zeros(16, np.uint64), np.zeros(16, np.uint64)
    L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)   # reset
    ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=B.PT_KERNEL_REGROUP))
    L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)
samples = W * H * spp
print(f"{W}x{H}, {spp} spp, regroup kernel: {samples} samples")
print(f"{'region':46s} {'exec per 64 samples':>20s} {'lanes/exec':>10s}")
for r in (9, 11, 12, 0, 3, 5, 6, 1, 2, 10):
    if ex[r]:
        print(f"{names[r]:46s} {ex[r] * 64.0 / samples:20.3f} {ln[r] / ex[r]:10.2f}")
