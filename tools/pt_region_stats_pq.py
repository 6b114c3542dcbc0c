"""Phase statistics of the two-slot path tracer scheduler (diagnostic build, see tools/pt_region_stats.py)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MC_LIB_PATH", os.path.join(ROOT, "vulkan-compute-tests_amd", "lib", "libmc_compute_stats.so"))
import __graft_entry__ as entry
B = entry.load_package().bindings
L = B.lib(); L.mc_debug_pt_region_stats.argtypes = [C.c_void_p, C.c_void_p]
W, H, spp = 900, 600, int(os.environ.get("PQ_STATS_SPP", "64"))
names = {9: "scheduler iterations", 10: "REGEN phase", 0: "ray generation", 11: "SPEC phase", 14: "swap-to-runnable", 1: "BOUNCE phase", 3: "diffuse block"}
with B.Context(0) as ctx:
    ex, ln = np.zeros(16, np.uint64), np.zeros(16, np.uint64)
    L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)
    ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=B.PT_KERNEL_PQ))
    L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)
samples = float(ln[0])
print(f"{int(samples)} samples; per 64 samples:")
names.update({12: "idle: has SPEC-pending slot", 13: "idle: has DEAD slot", 15: "idle: both FRESH/EMPTY"})
for r in (9, 10, 0, 11, 14, 1, 3, 12, 13, 15):
    if ex[r]:
        print(f"  {names[r]:24s} executions {ex[r] / samples * 64:8.3f}   lanes/execution {ln[r] / ex[r]:6.2f}")
