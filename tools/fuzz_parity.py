#!/usr/bin/env python3
"""Randomised bit-exact parity campaign: HIP path (through the C ABI) vs the CPU oracle, on the GPU box.

    python tools/fuzz_parity.py [--seconds 300] [--seed 1]

Not part of the test suite (run time is open-ended); it exists to hunt rare mismatches in the exact shortcuts:
the two-float Mandelbrot's fast block (fma error term, one-add escape filter, literal redo), the fp32 Mandelbrot's
bit filter, the path tracer's slab specialisation and the shadow rays that skip the walls.  Every case must be
bit-identical; the script prints a reproducer for the first mismatch of each family and exits non-zero.
"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# points on / near the boundary of the set (deep zooms stay interesting), plus points whose orbits hit exact zeros
INTERESTING = [(-0.7436438870371587, 0.13182590420531198), (-0.743643887037151, 0.131825904205330), (-0.75, 0.0),
               (0.25, 0.0), (-1.25, 0.0), (-0.1011, 0.9563), (0.0, 1.0), (-1.401155, 0.0), (-0.16, 1.0405),
               (0.001643721971153, -0.822467633298876), (-1.7687788, 0.0017389), (-0.235125, 0.827215), (0.0, 0.0),
               (-2.0, 0.0), (-1.0, 0.0), (0.0, -1.0), (-0.5, 0.0)]


def fuzz_mandel_f32(rng, ctx, B, O):
    W, H = int(rng.integers(1, 200)), int(rng.integers(1, 150))
    M = int(rng.integers(1, 600))
    if rng.random() < 0.5:
        c = INTERESTING[rng.integers(len(INTERESTING))]
        scale = 10.0 ** rng.uniform(-6, 0.5)
    else:
        c = (rng.uniform(-2.1, 0.8), rng.uniform(-1.3, 1.3))
        scale = 10.0 ** rng.uniform(-3, 0.7)
    sc = (scale, scale * rng.uniform(0.5, 1.5))
    p = B.mandelbrot_params(W, H, max_iter=M, centre=c, scale=sc)
    _, it = ctx.mandelbrot(p, want_rgba=False)
    ref = O.mandelbrot_iters(W, H, M, view=O.make_view(c[0], c[1], sc[0], sc[1]))
    return np.array_equal(it, ref), f"f32 W={W} H={H} M={M} centre={c} scale={sc}"


def fuzz_mandel_ds(rng, ctx, B, O):
    W, H = int(rng.integers(1, 96)), int(rng.integers(1, 64))
    M = int(rng.integers(1, 3000))
    c = INTERESTING[rng.integers(len(INTERESTING))]
    if rng.random() < 0.3:   # exact symmetric rows / columns: zeros in the orbit, hazard fallback
        H = 2 * (H // 2) + 2
    scale = 10.0 ** rng.uniform(-15, -1)
    sc = (scale, scale * rng.uniform(0.5, 1.5))
    p = B.mandelbrot_params(W, H, max_iter=M, precision=B.PRECISION_DS, centre=c, scale=sc)
    _, it = ctx.mandelbrot(p, want_rgba=False)
    ref = O.mandelbrot_iters(W, H, M, view=O.make_view(c[0], c[1], sc[0], sc[1]), precision=1)
    return np.array_equal(it, ref), f"ds W={W} H={H} M={M} centre={c} scale={sc}"


def fuzz_pathtrace(rng, ctx, B, O):
    planes = O.DEFAULT_PLANES.copy().reshape(6, 12)
    ns = int(rng.choice([3, 3, 3, 1, 2, 4, 5, 6, 7, 8]))       # 1 .. 8 spheres take the specialised kernels (round 4)
    spheres = np.zeros((ns, 12), np.float32)
    # walls: jitter offsets and colours, sometimes make one specular
    planes[:, 3] *= rng.uniform(0.8, 1.3, 6).astype(np.float32)
    planes[:, 8:11] = rng.uniform(0.1, 0.999, (6, 3)).astype(np.float32)
    if rng.random() < 0.3:
        planes[rng.integers(6), 11] = float(rng.choice([2, 3]))
    # spheres: positions anywhere in / around the room, any material, one or more lights
    lo = np.array([-planes[0, 3], -planes[3, 3], -planes[4, 3]]) - 0.5
    hi = np.array([planes[1, 3], planes[2, 3], min(planes[5, 3], 3.0)]) + 0.5
    for i in range(ns):
        spheres[i, 0:3] = rng.uniform(lo, hi).astype(np.float32)
        spheres[i, 3] = np.float32(rng.uniform(0.05, 1.0))
        spheres[i, 8:11] = rng.uniform(0.0, 0.999, 3).astype(np.float32)
        spheres[i, 11] = float(rng.choice([1, 1, 2, 3]))
        spheres[i, 4:7] = 0
    for i in rng.choice(ns, int(rng.integers(1, min(ns, 3) + 1)), replace=False):
        spheres[i, 4:7] = rng.uniform(5, 120, 3).astype(np.float32)
        spheres[i, 8:11] = 0
        spheres[i, 11] = 1.0
        if rng.random() < 0.6:   # pull most lights inside the room so that the shadow-ray shortcut is exercised
            spheres[i, 3] = np.float32(rng.uniform(0.05, 0.4))
            spheres[i, 0:3] = rng.uniform(lo + 1.2, hi - 1.2).astype(np.float32)
    if rng.random() < 0.08:   # a material code outside 1..3 leaves the ray where it was (no branch of :400-448 matches)
        code = float(rng.choice([0.0, 4.0, 0.4, 3.5, -1.0, 7.0]))
        if rng.random() < 0.5:
            planes[rng.integers(6), 11] = code
        else:
            spheres[rng.integers(ns), 11] = code
    if rng.random() < 0.05:   # extreme radii: roots far outside / tiny discriminants
        spheres[rng.integers(ns), 3] = np.float32(rng.choice([1e-3, 1e-2, 5.0, 30.0]))
    W, H, spp = int(rng.integers(1, 40)), int(rng.integers(1, 28)), int(rng.integers(1, 20))
    if rng.random() < 0.25:   # sample counts beyond the pool kernel's batch (16) and result-ring (64) sizes, on a small image
        W, H, spp = int(rng.integers(1, 12)), int(rng.integers(1, 10)), int(rng.integers(20, 200))
    depth = int(rng.choice([12, 12, 12, 3, 7, 15]))
    # 0: the automatic choice (the strict sample-pool kernel for closed-box scenes, else the round-synchronous kernels)
    flags = int(rng.choice([0, 0, 0, B.pt_force_s(1), B.pt_force_s(4), B.pt_force_s(16), B.PT_GENERIC_KERNEL, B.PT_NO_POOL_KERNEL,
                            B.PT_GENERIC_KERNEL | B.PT_SCENE_IN_MEMORY]))   # (the last: the generic kernel reading the scene from memory)
    cls = B.pathtrace_scene_class(planes, spheres)
    # a third of the cases as a progressive render in two sample ranges and / or as two row tiles (any alignment)
    cut = int(rng.integers(1, spp)) if spp > 1 and rng.random() < 0.3 else 0
    rcut = int(rng.integers(1, H)) if H > 1 and rng.random() < 0.3 else 0
    tiles = []
    for r0, r1 in ([(0, rcut), (rcut, H)] if rcut else [(0, H)]):
        acc = None
        for s0, s1 in ([(0, cut), (cut, spp)] if cut else [(0, spp)]):
            acc = ctx.pathtrace(B.pathtrace_params(W, H, spp, max_depth=depth, flags=flags, sample_begin=s0, sample_end=s1,
                                                   row_begin=r0, row_end=r1), planes=planes, spheres=spheres, acc=acc)
        tiles.append(acc)
    out = np.concatenate(tiles)
    ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC, max_depth=depth)
    ok = np.array_equal(bits(out), bits(ref))
    return ok, (f"pt class={cls} W={W} H={H} spp={spp} depth={depth} flags={flags} sample cut {cut} row cut {rcut}\nplanes={planes.tolist()}\n"
                f"spheres={spheres.tolist()}"), cls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    pkg = entry.load_package()
    B, O = pkg.bindings, entry.load_oracle()
    ctx = B.Context(0)
    rng = np.random.default_rng(args.seed)
    counts = {"f32": 0, "ds": 0, "pt": 0}
    classes = {}
    bad = {}
    t0 = last = time.time()
    while time.time() - t0 < args.seconds:
        for name, fn in (("f32", fuzz_mandel_f32), ("ds", fuzz_mandel_ds), ("pt", fuzz_pathtrace), ("pt", fuzz_pathtrace)):
            r = fn(rng, ctx, B, O)
            counts[name] += 1
            if name == "pt":
                classes[r[2]] = classes.get(r[2], 0) + 1
            if not r[0] and name not in bad:
                bad[name] = r[1]
                print("MISMATCH", r[1], flush=True)
        if time.time() - last > 30:
            last = time.time()
            print(f"[{last - t0:5.0f} s] cases {counts} pt scene classes {classes} mismatching families {list(bad)}", flush=True)
    print(f"done: cases {counts}, pt scene classes (bit 0 slab, 1 lights inside, 2 disjoint, 3 light intersects a diffuse sphere) {classes}, "
          f"mismatching families {list(bad)}")
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
