"""Where the TIME of a bounce iteration goes (diagnostic build: `make -C vulkan-compute-tests_amd wavetime EXP_NAME=x`).
Runs K2 at reduced spp with lib/libmc_compute_wt_<name>.so (per-region s_memtime stamps, pathtrace_kernel.h MC_WT) and prints,
per region, the wave cycles per sample round and per execution.  Wave cycles are latency of ONE wave with 5 others resident on
its SIMD, so shares — not absolute issue costs — are the result.
Usage (GPU box): python tools/pt_wave_time.py <lib.so> [flags]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MC_LIB_PATH"] = os.path.abspath(sys.argv[1])
import __graft_entry__ as entry  # noqa: E402

B = entry.load_package().bindings
L = B.lib()
L.mc_debug_pt_wave_time.argtypes = [C.c_void_p, C.c_void_p]
flags = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
W, H, spp = 900, 600, 64
names = {7: "ray generation + camera intersect", 0: "prologue", 1: "light sample set-up", 2: "shadow ray", 3: "light contribution",
         4: "diffuse bounce direction", 5: "mirror / glass", 6: "next intersect", 8: "round tail"}
with B.Context(0) as ctx:
    cy, mk = np.zeros(32, np.uint64), np.zeros(32, np.uint64)
    ctx.pathtrace(B.pathtrace_params(W, H, 16, math_mode=B.PT_MATH_FAST, flags=flags | B.pt_force_s(16)))   # warm-up
    L.mc_debug_pt_wave_time(cy.ctypes.data, mk.ctypes.data)   # reset
    ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=flags | B.pt_force_s(16)))
    L.mc_debug_pt_wave_time(cy.ctypes.data, mk.ctypes.data)
rounds = float(mk[7])
tot = float(cy.sum())
print(f"{os.path.basename(sys.argv[1])} flags={flags}: {int(rounds)} wave-rounds, {tot / rounds:.0f} wave cycles per round")
print(f"{'region':36s} {'marks/round':>11s} {'cycles/mark':>11s} {'cycles/round':>12s} {'share':>6s}")
for r in (7, 0, 1, 2, 3, 4, 5, 6, 8):
    if mk[r]:
        print(f"{names[r]:36s} {mk[r] / rounds:11.3f} {cy[r] / mk[r]:11.1f} {cy[r] / rounds:12.1f} {100.0 * cy[r] / tot:5.1f}%")
