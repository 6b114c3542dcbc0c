"""Where a wave of the lane-regrouping kernel spends its wall time (diagnostic build: `make stats_time`): s_memtime deltas
between marks, summed over all waves.  Waves share a SIMD, so a region's share includes waiting for the others' issue slots;
the split is what matters.  Usage (GPU box): python tools/rg_region_time.py [spp]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MC_LIB_PATH", os.path.join(ROOT, "vulkan-compute-tests_amd", "lib", "libmc_compute_time.so"))
import __graft_entry__ as entry  # noqa: E402
B = entry.load_package().bindings
L = B.lib()
L.mc_debug_pt_region_stats.argtypes = [C.c_void_p, C.c_void_p]
W, H = 900, 600
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
names = {4: "swap: control words + reservation", 5: "swap: record writes + reads", 6: "swap: camera items", 7: "swap: commit", 0: "swap: rest (classification, idle check)", 1: "heads (camera / diffuse / mirror / glass)", 2: "intersect + prologue", 3: "retire"}
for mode, mname in ((B.PT_MATH_FAST, "fast"),):
    with B.Context(0) as ctx:
        ex, ln = np.zeros(16, np.uint64), np.zeros(16, np.uint64)
        L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)   # reset
        ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=mode, flags=B.PT_KERNEL_REGROUP))
        L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)
    tot = float(ln[:8].sum())
    print(f"{mname}: {W}x{H}, {spp} spp")
    for r in (4, 5, 6, 7, 0, 1, 2, 3):
        print(f"  {names[r]:44s} {100.0 * ln[r] / tot:6.2f} %   {ln[r] / max(1, ex[r]):9.1f} cycles per execution")
