"""Fast-math tolerance beyond the default scene: random slab scenes (jittered walls, some specular, spheres moved / resized /
re-materialised, one or two lights) rendered in MC_PT_MATH_FAST and compared with the CPU oracle evaluated with libm, the same
statistics as tests/test_gpu_fullsize.py::test_k2_fast_math_within_the_stated_tolerance (RMSE, 99.9-percentile per-pixel RGB L2,
mean difference; 8-bit units of the tonemapped storage buffer).  The oracle's own mc-vs-libm spread is printed beside it.
The request is made as a caller makes it: the tier the host chose for it (mc_pathtrace_select_kernel: 1 fast, 2 careful, 0 strict) is
printed, and for a scene the host promoted the fast tier forced with MC_PT_NO_FAST_GUARD is measured beside it — what the promotion avoided.
  python tools/fast_tolerance_scenes.py [--scenes 6] [--size 300 200] [--spp 256] [--seed 3]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def scene(rng, O):
    planes = O.DEFAULT_PLANES.copy().reshape(6, 12)
    spheres = O.DEFAULT_SPHERES.copy().reshape(3, 12)
    planes[:, 3] *= rng.uniform(0.9, 1.2, 6).astype(np.float32)
    planes[:, 8:11] = rng.uniform(0.2, 0.95, (6, 3)).astype(np.float32)
    if rng.random() < 0.5:
        planes[rng.integers(6), 11] = float(rng.choice([2, 3]))
    for i in range(2):
        spheres[i, 0] += np.float32(rng.uniform(-0.4, 0.4)); spheres[i, 2] += np.float32(rng.uniform(-0.4, 0.4))
        spheres[i, 3] = np.float32(rng.uniform(0.4, 0.9)); spheres[i, 1] = np.float32(-planes[3, 3] + spheres[i, 3])
        spheres[i, 11] = float(rng.choice([1, 2, 3]))
        if spheres[i, 11] == 1:
            spheres[i, 8:11] = rng.uniform(0.3, 0.9, 3).astype(np.float32)
    spheres[2, 0] = np.float32(rng.uniform(-1.0, 1.0)); spheres[2, 2] = np.float32(rng.uniform(-1.0, 1.0))
    spheres[2, 3] = np.float32(rng.uniform(0.15, 0.35)); spheres[2, 1] = np.float32(planes[2, 3] - spheres[2, 3] - 0.2)
    return planes, spheres


def stats(a, b):
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return float(np.sqrt((d ** 2).mean())), float(np.percentile(np.sqrt((d ** 2).sum(-1)), 99.9)), float(d.mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=6)
    ap.add_argument("--size", type=int, nargs=2, default=[300, 200])
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--seed", type=int, default=3)
    a = ap.parse_args()
    B = entry.load_package().bindings
    O = entry.load_oracle()
    rng = np.random.default_rng(a.seed)
    W, H = a.size
    by_tier = {}
    with B.Context(0) as ctx:
        for k in range(a.scenes):
            planes, spheres = scene(rng, O)
            cls = B.pathtrace_scene_class(planes, spheres)
            q = B.pathtrace_params(W, H, a.spp, math_mode=B.PT_MATH_FAST)
            ran = B.pathtrace_select_kernel(q, planes, spheres).math_mode
            fast = ctx.pathtrace(q, planes=planes, spheres=spheres)
            strict = ctx.pathtrace(B.pathtrace_params(W, H, a.spp), planes=planes, spheres=spheres)
            libm = O.pathtrace(W, H, a.spp, planes=planes, spheres=spheres, math_mode=O.MATH_LIBM)
            mc = O.pathtrace(W, H, a.spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
            exact = bool(np.array_equal(strict.view(np.uint32), mc.view(np.uint32)))
            r, p, m = stats(fast, libm)
            yr, yp, _ = stats(mc, libm)
            mats = [int(x) for x in list(planes[:, 11]) + list(spheres[:, 11])]
            forced = ""
            if ran != B.PT_MATH_FAST:
                f1 = ctx.pathtrace(B.pathtrace_params(W, H, a.spp, math_mode=B.PT_MATH_FAST, flags=B.PT_NO_FAST_GUARD), planes=planes, spheres=spheres)
                fr, fp, _ = stats(f1, libm)
                forced = f"   (fast tier forced: rmse {fr:.4f} p99.9 {fp:.3f})"
            by_tier.setdefault(ran, []).append((p, r, k))
            print(f"scene {k}: class {cls} materials {mats} tier {ran}  fast vs libm: rmse {r:.4f} p99.9 {p:.3f} mean {m:+.5f}   "
                  f"oracle mc vs libm: rmse {yr:.4f} p99.9 {yp:.3f}   strict == oracle: {exact}{forced}", flush=True)
    for tier, rows in sorted(by_tier.items()):
        worst = max(rows)
        print(f"# tier {tier}: {len(rows)} scenes, worst p99.9 {worst[0]:.3f} (scene {worst[2]}), worst rmse {max(r[1] for r in rows):.4f}, "
              f"outside the bound (rmse 0.5 / p99.9 4.0): {sum(1 for r in rows if r[0] > 4.0 or r[1] > 0.5)}")


if __name__ == "__main__":
    main()
