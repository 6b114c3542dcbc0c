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
names = {9: "scheduler iteration", 11: "batch (spill + specular/camera refill)", 14: "mixed iteration (no batch possible)",
         13: "reservation gave up (contended)", 15: "SQ full: specular lanes kept", 12: "idle spin", 7: "avg camera items available at a swap",
         8: "avg parked specular paths at a swap", 2: "avg items in flight (next_item - committed)", 0: "head: ray generation",
         3: "head: diffuse (NEE + shadow ray + bounce)", 5: "head: mirror", 6: "head: glass", 1: "intersect + prologue",
         10: "retire to the reorder ring"}
with B.Context(0) as ctx:
    ex, ln = np.zeros(16, np.uint64), np.zeros(16, np.uint64)
    L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)   # reset
    ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=B.PT_KERNEL_REGROUP))
    L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)
samples = W * H * spp
print(f"{W}x{H}, {spp} spp, regroup kernel: {samples} samples")
print(f"{'region':46s} {'exec per 64 samples':>20s} {'lanes/exec (or avg)':>20s}")
for r in (9, 11, 14, 13, 15, 12, 7, 8, 2, 0, 3, 5, 6, 1, 10):
    if ex[r]:
        print(f"{names[r]:46s} {ex[r] * 64.0 / samples:20.3f} {ln[r] / ex[r]:20.2f}")
