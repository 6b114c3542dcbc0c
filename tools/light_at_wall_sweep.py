"""The fast-math guard's blind spot the round-4 advisor named: light_nearly_enclosed (csrc/pathtrace.hip) compares emissive spheres with
diffuse and mirror SPHERES only.  A light that touches or intersects a diffuse PLANE — a wall, or any plane of a generic scene — is
sampled at point-blank range from that plane in the same way.  Does fast math leave its tolerance there?

The reference scene's light (radius 0.2; the fast tolerance is stated on this scene) is moved towards the ceiling (plane y = 2, diffuse) so
that its top pokes through by f of its radius, f from -1.5 (a radius and a half clear) over 0 (touching) to 1.5 (the centre above the
ceiling).  For each f, against the oracle with libm at 300 x 200 x 256 spp (bound: RMSE 0.5 / p99.9 L2 4, stated at 500 spp):
    slab   — the scene as it is (the host leaves the closed-box class once the light is within its margin of a wall: the round-synchronous
             slab kernel, fast math, unguarded);
    generic — the same scene with the planes in another index order (not a slab scene: the generic fast kernel);
and the oracle's own spread (explicit fp32 math against libm) as the yardstick.
    GPU box:  python tools/light_at_wall_sweep.py > gpurun_out/r05_light_at_wall_sweep.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry  # noqa: E402
from fast_tolerance_scenes import stats  # noqa: E402

B, O = entry.load_package().bindings, entry.load_oracle()
P = O.DEFAULT_PLANES.copy().reshape(6, 12)
S0 = O.DEFAULT_SPHERES.copy().reshape(3, 12)
W, H, spp = 300, 200, 256
r = float(S0[2, 3])
ceiling = float(P[2, 3])
with B.Context(0) as ctx:
    print(f"# light radius {r:.2f} under the diffuse ceiling y = {ceiling:.1f}; {W}x{H}x{spp}; poke-through = (top of the light - ceiling) / radius")
    print(f"# {'poke-through':>12s} {'class':>5s} {'kernel':>7s} | {'slab rmse':>9s} {'p99.9':>8s} {'mean':>8s} | {'generic rmse':>12s} {'p99.9':>8s} {'mean':>8s} | oracle mc vs libm rmse / p99.9")
    for f in (-1.5, -0.5, -0.1, -0.02, 0.0, 0.02, 0.1, 0.5, 1.0, 1.5):
        S = S0.copy()
        S[2, 1] = np.float32(ceiling - r + f * r)
        cls = B.pathtrace_scene_class(P, S)
        q = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=B.PT_NO_FAST_GUARD)
        ki = B.pathtrace_select_kernel(q, P, S)
        libm = O.pathtrace(W, H, spp, planes=P, spheres=S, math_mode=O.MATH_LIBM)
        mc = O.pathtrace(W, H, spp, planes=P, spheres=S, math_mode=O.MATH_MC)
        slab = ctx.pathtrace(q, planes=P, spheres=S)
        perm = P[[2, 3, 0, 1, 4, 5]]
        gen = ctx.pathtrace(q, planes=perm, spheres=S)
        libm_g = O.pathtrace(W, H, spp, planes=perm, spheres=S, math_mode=O.MATH_LIBM)
        a, b, c = stats(slab, libm), stats(gen, libm_g), stats(mc, libm)
        print(f"  {f:12.2f} {cls:5d} {B.PT_KERNEL_NAMES[ki.kernel]:>7s} | {a[0]:9.4f} {a[1]:8.3f} {a[2]:+8.4f} | {b[0]:12.4f} {b[1]:8.3f} {b[2]:+8.4f} | {c[0]:.4f} / {c[1]:.3f}",
              flush=True)
