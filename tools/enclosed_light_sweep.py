"""Calibration of the fast-math guard (csrc/pathtrace.hip, light_nearly_enclosed; scene class bit 3): the scene tools/fuzz_fast.py
found in round 3 — a light poking 0.01 out of an opaque sphere — with the light moved along the line of centres so that it pokes
out by f light radii, f from -0.5 (well inside) over 2 (the spheres touch) to 5 (well clear of the sphere).  For each f the UNGUARDED fast kernel
(MC_PT_NO_FAST_GUARD) against the oracle with libm at 300 x 200 x 256 spp — the fast tolerance is RMSE 0.5 / p99.9 L2 4 — beside
the oracle's own spread (explicit fp32 math against libm).  The guard's margin must sit where the fast kernel is back inside the
bound with headroom.   GPU box:  python tools/enclosed_light_sweep.py > gpurun_out/r04_enclosed_light_sweep.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry  # noqa: E402
from fast_tolerance_scenes import stats  # noqa: E402
from test_abi import ENCLOSED_LIGHT_PLANES, ENCLOSED_LIGHT_SPHERES  # noqa: E402

B, O = entry.load_package().bindings, entry.load_oracle()
# MC_SWEEP_MATH=careful: the same sweep through the careful tier (round 5: does the tier with a fifth of the forks hold the bound where
# the fast tier does not?)
MODE = B.PT_MATH_FAST_CAREFUL if os.environ.get("MC_SWEEP_MATH") == "careful" else B.PT_MATH_FAST
P, S0 = np.float32(ENCLOSED_LIGHT_PLANES), np.float32(ENCLOSED_LIGHT_SPHERES)
W, H, spp = 300, 200, 256
with B.Context(0) as ctx:
    for r_light, material in ((float(S0[1, 3]), 1.0), (0.3, 1.0), (float(S0[1, 3]), 2.0)):
        print(f"# light radius {r_light:.3f}, enclosing sphere radius {S0[2, 3]:.3f}, material {int(material)}; {W}x{H}x{spp}")
        print(f"# {'poke-out / r_light':>18s} {'class':>5s} | {'fast rmse':>9s} {'p99.9':>8s} {'mean':>8s} | {'box rmse':>8s} {'p99.9':>8s} | oracle mc vs libm rmse / p99.9")
        for f in (-0.5, 0.0, 0.075, 0.25, 0.5, 1.0, 1.5, 1.75, 2.0, 2.5, 3.0, 3.5, 4.0, 5.0):
            S = S0.copy()
            S[1, 3] = r_light
            S[2, 11] = material
            big = S[2]
            axis = (S0[1, :3] - big[:3]) / np.linalg.norm(S0[1, :3] - big[:3])
            S[1, :3] = big[:3] + axis * (big[3] - r_light + f * r_light)
            cls = B.pathtrace_scene_class(P, S)
            libm = O.pathtrace(W, H, spp, planes=P, spheres=S, math_mode=O.MATH_LIBM)
            mc = O.pathtrace(W, H, spp, planes=P, spheres=S, math_mode=O.MATH_MC)
            fast = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=MODE, flags=B.PT_NO_FAST_GUARD), planes=P, spheres=S)
            nobox = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=MODE, flags=B.PT_NO_FAST_GUARD | B.PT_NO_BOX_KERNEL), planes=P, spheres=S)
            a, b, c = stats(fast, libm), stats(nobox, libm), stats(mc, libm)
            print(f"  {f:18.3f} {cls:5d} | {a[0]:9.4f} {a[1]:8.3f} {a[2]:+8.4f} | {b[0]:8.4f} {b[1]:8.3f} | {c[0]:.4f} / {c[1]:.3f}", flush=True)
