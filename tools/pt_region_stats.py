"""Divergence picture of the path tracer kernel (diagnostic build: `make -C vulkan-compute-tests_amd stats`).
Runs K2 at reduced spp with lib/libmc_compute_stats.so and prints, per code region, how many times a wave executed
it per sample round and how many lanes were active — the numbers quoted in DESIGN.md §3.3.
Usage (GPU box): MC_LIB_PATH=vulkan-compute-tests_amd/lib/libmc_compute_stats.so python tools/pt_region_stats.py [S]"""
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
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H, spp = 900, 600, 64
names = {0: "ray generation", 1: "primary intersect (per depth iteration)", 2: "bounce prologue", 3: "diffuse: NEE + shadow ray",
         4: "NEE contribution", 8: "diffuse bounce dir", 5: "mirror", 6: "glass", 7: "glass refraction branch"}
with B.Context(0) as ctx:
    ex, ln = np.zeros(16, np.uint64), np.zeros(16, np.uint64)
    L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)   # reset
    ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=B.pt_force_s(S)))
    L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)
rounds = ex[0]                                   # one ray generation per wave per round
print(f"K2-like {W}x{H}, {spp} spp, S={S}: {int(rounds)} wave-rounds, {int(ln[0])} samples ({int(ln[0]) / rounds:.1f} lanes/round)")
print(f"{'region':42s} {'exec/round':>10s} {'lanes/exec':>10s} {'lane-execs/sample':>18s}")
for r in (0, 1, 2, 3, 4, 8, 5, 6, 7):
    if ex[r]:
        print(f"{names[r]:42s} {ex[r] / rounds:10.3f} {ln[r] / ex[r]:10.2f} {ln[r] / ln[0]:18.3f}")
