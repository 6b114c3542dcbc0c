"""Where the iterations of the sample-pool kernel go (diagnostic build: `make -C vulkan-compute-tests_amd stats`): per block of
csrc/pathtrace_pool.h how often a wave executes it per iteration and how many lanes are active when it does — K2 in fast math.
The dynamic counterpart of tools/isa_blocks.py (static instructions per block); together they give the instruction budget of an
iteration (profiles/r03_pool_region_stats.txt).  GPU box:
  MC_LIB_PATH=vulkan-compute-tests_amd/lib/libmc_compute_stats.so python tools/pool_region_stats.py [spp] [fast|careful|strict]
Round 6: the counters of all three tiers' translation units are summed (mc_debug_pt_region_stats), so a careful or strict request —
explicit, or what the host makes of a fast request on five or more spheres / an enclosed light — is counted too."""
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
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 500
tier = sys.argv[2] if len(sys.argv) > 2 else "fast"
mode = {"fast": B.PT_MATH_FAST, "careful": B.PT_MATH_FAST_CAREFUL, "strict": B.PT_MATH_STRICT}[tier]
names = [(0, "iteration"), (1, "refill bookkeeping (some lane is free)"), (2, "batch of 64 camera rays + their intersection"),
         (22, "  root block, sphere 0 (camera rays)"), (23, "  root block, sphere 1"), (24, "  root block, sphere 2"),
         (3, "free lanes take a stash entry"), (4, "bounce: prologue"), (5, "  sphere normal"), (12, "  emission of the hit"),
         (13, "diffuse: light sample + shadow test"), (8, "  light contribution"), (9, "  bounce off a wall (permutation)"),
         (10, "  bounce, general basis (a lane is on a diffuse sphere)"), (11, "mirror / glass"), (6, "  glass"), (7, "  glass: refraction branch"),
         (14, "intersection of the next depth"), (16, "  root block, sphere 0"), (17, "  root block, sphere 1"), (18, "  root block, sphere 2"),
         (15, "roulette (depth > 5)")]
with B.Context(0) as ctx:
    ex, ln = np.zeros(32, np.uint64), np.zeros(32, np.uint64)
    L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)   # reset
    ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=mode))
    L.mc_debug_pt_region_stats(ex.ctypes.data, ln.ctypes.data)
it = float(ex[0])
print(f"K2 {W}x{H}x{spp}, {tier} sample-pool kernel: {int(ex[0])} wave-iterations ({it / (W * H / 4):.1f} per wave), "
      f"{int(ln[4])} lane-bounces = {float(ln[4]) / (W * H * spp):.2f} per sample, {float(ln[4]) / it:.1f} of 64 lanes in a bounce")
print(f"{'block':58s} {'runs per iteration':>18s} {'lanes when it runs':>18s} {'lane share':>10s}")
for r, name in names:
    if ex[r]:
        print(f"{name:58s} {float(ex[r]) / it:18.3f} {float(ln[r]) / float(ex[r]):18.2f} {float(ln[r]) / float(ex[r]) / 64:10.2f}")
