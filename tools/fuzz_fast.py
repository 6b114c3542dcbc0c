#!/usr/bin/env python3
"""Randomised sanity campaign for the FAST (toleranced) path tracer kernels on the GPU box — the counterpart of tools/fuzz_parity.py,
which holds the strict kernels to bit-identity.  Fast math has no bit-exact reference, so every case is held to what a shortcut
gone wrong would break: the output is finite (a NaN ray must gather nothing, csrc/pathtrace_kernel.h: intersect_slab<Closed>), and
against the oracle evaluated with libm on the same sample keys only a few pixels may differ by more than a forked sample can move
them, with no drift of the mean.  Scenes: random closed boxes with the camera and the lights inside — what the host selects the
sample-pool kernel (disjoint spheres) or the closed-box round-synchronous kernel (overlapping spheres, forced widths) for — with any
materials on the spheres, mirror walls, lights as any sphere, radii from specks to spheres that fill the room.

    python tools/fuzz_fast.py [--seconds 300] [--seed 1]
"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def scene(rng, O, enclose=False, many=False):
    planes = O.DEFAULT_PLANES.copy().reshape(6, 12)
    # 1 .. 8 spheres take the specialised kernels (round 4); --many weights 5 .. 8 up: the counts an MC_PT_MATH_FAST request is rendered
    # by the careful tier for (round 5)
    ns = int(rng.choice([6, 7, 8, 8, 5, 6, 7, 3, 4] if many else [3, 3, 3, 1, 2, 4, 5, 6, 8]))
    spheres = np.zeros((ns, 12), np.float32)
    planes[:, 3] *= rng.uniform(0.85, 1.25, 6).astype(np.float32)
    planes[:, 8:11] = rng.uniform(0.05, 0.999, (6, 3)).astype(np.float32)
    if rng.random() < 0.4:
        planes[rng.integers(6), 11] = 2.0                       # a mirror wall (a wall of glass leaves the closed-box class)
    lo = np.array([-planes[0, 3], -planes[3, 3], -planes[4, 3]]) + 0.3
    hi = np.array([planes[1, 3], planes[2, 3], min(planes[5, 3], 3.0)]) - 0.3
    for i in range(ns):
        spheres[i, 3] = np.float32(rng.choice([rng.uniform(0.05, 0.3), rng.uniform(0.3, 1.0), rng.uniform(1.0, 1.6)], p=[0.3, 0.6, 0.1]))
        spheres[i, 0:3] = rng.uniform(lo, hi).astype(np.float32)
        spheres[i, 8:11] = rng.uniform(0.0, 0.999, 3).astype(np.float32)
        spheres[i, 11] = float(rng.choice([1, 2, 3]))
        spheres[i, 4:7] = 0
    for i in rng.choice(ns, int(rng.integers(1, min(ns, 2) + 1)), replace=False):   # one or two lights, small, well inside the room
        spheres[i, 4:7] = rng.uniform(5, 120, 3).astype(np.float32)
        spheres[i, 8:11] = 0
        spheres[i, 11] = 1.0
        spheres[i, 3] = np.float32(rng.uniform(0.05, 0.4))
        spheres[i, 0:3] = rng.uniform(lo + 0.4, hi - 0.4).astype(np.float32)
    # A light (all but) ENCLOSED by an opaque sphere is outside what fast math can hold (found by this campaign in round 3, seed 42:
    # fast kernels of every kind 2-4 % of the pixels off, RMSE 3.6): the host classifies such scenes (scene class bit 3) and renders
    # them with the strict kernels — main() expects the oracle's bits there.  `enclose` aims a share of the scenes AT that boundary:
    # a light placed so that it pokes out of an opaque sphere by -0.5 ... 5 of its own radii.
    if enclose and rng.random() < 0.35:
        lights = [i for i in range(ns) if spheres[i, 4:7].any()]
        darks = [j for j in range(ns) if not spheres[j, 4:7].any()]
        if lights and darks:
            i, j = int(rng.choice(lights)), int(rng.choice(darks))
            spheres[j, 3] = np.float32(rng.uniform(0.5, 1.1))
            if spheres[j, 11] == 3.0:
                spheres[j, 11] = float(rng.choice([1, 2]))
            spheres[j, 0:3] = rng.uniform(lo + 0.9, hi - 0.9).astype(np.float32)
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            out = rng.uniform(-0.5, 5.0) * spheres[i, 3]   # (the criterion: < 3.5 radii for a diffuse sphere, < 0.25 for a mirror)
            spheres[i, 0:3] = (spheres[j, 0:3] + u * (spheres[j, 3] - spheres[i, 3] + out)).astype(np.float32)
    return planes, spheres


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--enclose", action="store_true", help="aim a third of the scenes at the enclosed-light boundary (scene class bit 3)")
    ap.add_argument("--many", action="store_true", help="weight scenes with 5 .. 8 spheres up (rendered by the careful tier)")
    args = ap.parse_args()
    B, O = entry.load_package().bindings, entry.load_oracle()
    ctx = B.Context(0)
    rng = np.random.default_rng(args.seed)
    n = {"cases": 0, "closed_box": 0, "disjoint": 0, "guarded": 0, "careful": 0}
    worst = {"frac_far": 0.0, "mean": 0.0}
    bad = []
    t0 = last = time.time()
    while time.time() - t0 < args.seconds:
        planes, spheres = scene(rng, O, args.enclose, args.many)
        cls = B.pathtrace_scene_class(planes, spheres)
        W, H = int(rng.integers(8, 40)), int(rng.integers(8, 28))
        spp = int(rng.choice([16, 24, 33, 64, 100]))
        depth = int(rng.choice([12, 12, 5, 8]))
        flags = int(rng.choice([0, 0, 0, B.PT_NO_POOL_KERNEL, B.pt_force_s(16), B.pt_force_s(1)]))
        q = B.pathtrace_params(W, H, spp, max_depth=depth, math_mode=B.PT_MATH_FAST, flags=flags)
        out = ctx.pathtrace(q, planes=planes, spheres=spheres)
        n["cases"] += 1
        ran = B.pathtrace_select_kernel(q, planes, spheres).math_mode      # the tier that rendered the request
        n["careful"] += int(ran == B.PT_MATH_FAST_CAREFUL)
        if (ran == B.PT_MATH_STRICT) != bool(cls & B.PT_SCENE_LIGHT_ENCLOSED) or \
           (ran == B.PT_MATH_FAST_CAREFUL) != (bool(cls & (B.PT_SCENE_MANY_SPHERES | B.PT_SCENE_SPECULAR)) and not cls & B.PT_SCENE_LIGHT_ENCLOSED):
            bad.append(f"tier {ran} does not follow the scene class {cls} with {len(spheres)} spheres")
        if cls & B.PT_SCENE_LIGHT_ENCLOSED:   # classified outside the fast tolerance: the strict kernels must have rendered it
            n["guarded"] += 1
            mc = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC, max_depth=depth)
            if not np.array_equal(out.view(np.uint32), mc.view(np.uint32)) and len(bad) < 5:
                bad.append(f"GUARDED scene not bit-identical to the oracle: class={cls} W={W} H={H} spp={spp} depth={depth} flags={flags}\n"
                           f"planes={planes.tolist()}\nspheres={spheres.tolist()}")
                print("SUSPECT", bad[-1], flush=True)
            continue
        ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_LIBM, max_depth=depth)
        n["closed_box"] += int(cls & 3 == 3)
        n["disjoint"] += int(cls & 4 == 4)
        d = out[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64)
        l2 = np.sqrt((d ** 2).sum(-1))
        # A forked sample (a near-tie decided the other way) moves ONE pixel — usually by a few of its 255 / spp weights, up to
        # saturation when it reaches a light through mirrors (emission ~100 on one of spp samples).  So: few pixels "far" (more than
        # three weights), the typical pixel essentially equal, and the mean within what two saturated pixels could do.
        far = float((l2 > 3.0 * 255.0 / spp).mean())
        ok = (bool(np.isfinite(out).all()) and far <= 0.03 and float(np.median(l2)) <= 0.05
              and abs(d.mean()) <= 0.05 + 2.0 * 255.0 / (W * H))
        worst["frac_far"] = max(worst["frac_far"], far)
        worst["mean"] = max(worst["mean"], abs(float(d.mean()))) if np.isfinite(d).all() else float("nan")
        if not ok and len(bad) < 5:
            bad.append(f"class={cls} W={W} H={H} spp={spp} depth={depth} flags={flags} finite={bool(np.isfinite(out).all())} far={far:.4f} "
                       f"mean={d.mean():+.4f}\nplanes={planes.tolist()}\nspheres={spheres.tolist()}")
            print("SUSPECT", bad[-1], flush=True)
        if time.time() - last > 30:
            last = time.time()
            print(f"[{last - t0:5.0f} s] {n} worst {worst} suspects {len(bad)}", flush=True)
    print(f"done: {n}, worst {worst}, suspects {len(bad)}")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
