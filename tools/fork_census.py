#!/usr/bin/env python3
"""Fork census of the FAST path tracer kernels: which approximation of fast math carries how many of the samples
that take another path than the reference arithmetic's, on the box scenes of tests/test_gpu_scenes.py (1 .. 8 spheres).

For each library given — the shipped one and `make exp` builds with ONE fast-math shortcut switched off (MC_PT_FAST_*), the hardware
seeds replaced by correctly rounded operations (MC_PT_FAST_IEEE), the hardware sine / cosine by the strict pair
(MC_PT_FAST_ACCURATE_SINCOS), contraction off (MC_PT_FAST_CONTRACT=0/1), directions re-normalised (MC_PT_FAST_RENORMALISE) — every
scene is rendered at the size the bound is asserted at (300 x 200, 500 spp) and compared with the CPU oracle evaluated with libm:
    rmse, p99.9 = the bound's two statistics (8-bit units; <= 0.5 / <= 4);   far = share of pixels further than 4;
    forked     = share of pixels whose largest channel difference exceeds 0.5 (a pixel that visibly took another sample);
    mean +- se = mean difference over the image and its standard error (per-pixel means as the sample);
    lin        = relative difference of the image's clamped LINEAR radiance sum (rendered with spp + 1 so that :453 is not applied).
The oracle's own implementation-defined spread (mc math against libm) is printed as the yardstick.

    python tools/fork_census.py [--scenes 8:1,8:3] [--size 300 200] [--spp 500] lib1.so lib2.so ...
Each library is loaded in its own child process (MC_LIB_PATH)."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402

CHILD = r"""
import sys, os, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as entry
B = entry.load_package().bindings
z = np.load({scenes!r})
W, H, spp = {W}, {H}, {spp}
out = {{}}
with B.Context(0) as ctx:
    for k in range(int(z["n"])):
        planes, spheres = z[f"planes{{k}}"], z[f"spheres{{k}}"]
        flags, MODE = {flags}, {mode}
        q = B.pathtrace_params(W, H, spp, math_mode=MODE, flags=flags)
        out[f"tm{{k}}"] = ctx.pathtrace(q, planes=planes, spheres=spheres)
        q = B.pathtrace_params(W, H, spp + 1, math_mode=MODE, flags=flags, sample_begin=0, sample_end=spp)
        out[f"lin{{k}}"] = ctx.pathtrace(q, planes=planes, spheres=spheres)
        ki = B.pathtrace_select_kernel(q, planes, spheres)
        out[f"kernel{{k}}"] = np.int32(ki.kernel + 10 * ki.math_mode)
    if {time}:   # kernel time at 900 x 600, 100 spp (device form, HIP events on the launch stream)
        import torch
        st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
        buf = torch.zeros((600, 900, 4), dtype=torch.float32, device="cuda")
        for k in range(int(z["n"])):
            planes, spheres = z[f"planes{{k}}"], z[f"spheres{{k}}"]
            q = B.pathtrace_params(900, 600, 100, math_mode={mode}, flags={flags})
            for _ in range(2): ctx.pathtrace_device(q, buf.data_ptr(), planes=planes, spheres=spheres, stream=s)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3): ctx.pathtrace_device(q, buf.data_ptr(), planes=planes, spheres=spheres, stream=s)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 3)
            out[f"ms{{k}}"] = np.float64(best)
np.savez({out!r}, **out)
"""


def stats(img, ref):
    d = img[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64)
    l2 = np.sqrt((d ** 2).sum(-1))
    pm = d.mean(-1)
    return dict(rmse=float(np.sqrt((d ** 2).mean())), p999=float(np.percentile(l2, 99.9)), far=float((l2 > 4.0).mean()),
                forked=float((np.abs(d).max(-1) > 0.5).mean()), mean=float(pm.mean()), se=float(pm.std() / np.sqrt(pm.size)))


def lin_rel(img, ref):
    a = np.clip(img[..., :3].astype(np.float64), 0.0, 1.0).sum()
    b = np.clip(ref[..., :3].astype(np.float64), 0.0, 1.0).sum()
    return (a - b) / b


def fmt(tag, s, lin):
    return (f"  {tag:28s} rmse {s['rmse']:.4f}  p99.9 {s['p999']:6.3f}  far {100 * s['far']:.3f} %  forked {100 * s['forked']:.2f} %  "
            f"mean {s['mean']:+.5f} +- {s['se']:.5f}  lin {lin:+.2e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--scenes", default="ref,5:2,6:1,7:1,8:1,8:3",
                    help="ref = the reference scene; n:l = tests' box scene with n spheres, l lights (the test's seed); n:l:seed = another "
                         "seed; n:l:seed:spec = every sphere that is not a light made specular (mirror / glass alternating)")
    ap.add_argument("--time", action="store_true", help="also time each scene at 900 x 600, 100 spp")
    ap.add_argument("--modes", default="fast", help="comma list of: fast (the request as a caller makes it: the host may render it with the careful "
                    "tier or strict), tier1 (the fast tier itself, MC_PT_NO_FAST_GUARD), careful (MC_PT_MATH_FAST_CAREFUL), strict")
    ap.add_argument("--size", type=int, nargs=2, default=[300, 200])
    ap.add_argument("--spp", type=int, default=500)
    a = ap.parse_args()
    from test_gpu_scenes import box_scene
    O = entry.load_oracle()
    W, H = a.size
    scenes, names = [], []
    for tok in a.scenes.split(","):
        if tok == "ref":
            scenes.append((O.DEFAULT_PLANES.copy().reshape(6, 12), O.DEFAULT_SPHERES.copy().reshape(3, 12)))
        elif tok.startswith("g:"):    # g:planes:spheres:lights[:seed] — the tests' random GENERIC scene (any planes, spheres anywhere)
            from test_gpu_scenes import random_scene
            f = [int(x) for x in tok.split(":")[1:]]
            scenes.append(random_scene(np.random.default_rng(f[3] if len(f) > 3 else 100 + f[1]), f[0], f[1], f[2]))
        else:
            f = tok.split(":")
            n, l = int(f[0]), int(f[1])
            pl, sp = box_scene(O, n, np.random.default_rng(int(f[2]) if len(f) > 2 else 40 + 10 * n + l), l)   # default: the seeds of the test
            if len(f) > 3 and f[3] == "spec":
                k = 0
                for q in sp:
                    if not q[4:7].any():
                        q[11] = 2.0 + (k & 1); k += 1
            scenes.append((pl, sp))
        names.append(tok)
    tmp = tempfile.mkdtemp(prefix="fork_census_")
    sfile = os.path.join(tmp, "scenes.npz")
    np.savez(sfile, n=len(scenes), **{f"planes{k}": s[0] for k, s in enumerate(scenes)}, **{f"spheres{k}": s[1] for k, s in enumerate(scenes)})
    refs = []
    t0 = time.time()
    for k, (pl, sp) in enumerate(scenes):
        tm = O.pathtrace(W, H, a.spp, planes=pl, spheres=sp, math_mode=O.MATH_LIBM)
        lin = O.pathtrace(W, H, a.spp + 1, planes=pl, spheres=sp, math_mode=O.MATH_LIBM, sample_end=a.spp)
        mc = O.pathtrace(W, H, a.spp, planes=pl, spheres=sp, math_mode=O.MATH_MC)
        mcl = O.pathtrace(W, H, a.spp + 1, planes=pl, spheres=sp, math_mode=O.MATH_MC, sample_end=a.spp)
        refs.append((tm, lin, mc, mcl))
    print(f"# fork census, {W} x {H} x {a.spp} spp against the oracle with libm (oracle renders: {time.time() - t0:.0f} s)", flush=True)
    for k, nm in enumerate(names):
        mats = [int(x) for x in scenes[k][1][:, 11]]
        print(f"scene {nm}: sphere materials {mats}, emitters {[int(bool(s[4:7].any())) for s in scenes[k][1]]}")
        print(fmt("oracle mc math (yardstick)", stats(refs[k][2], refs[k][0]), lin_rel(refs[k][3], refs[k][1])), flush=True)
    MODES = {"fast": (1, 0), "tier1": (1, 64), "careful": (2, 0), "strict": (0, 0)}
    for lib, mode in [(l, m) for l in a.libs for m in a.modes.split(",")]:
        out = os.path.join(tmp, os.path.basename(lib) + "." + mode + ".npz")
        r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, scenes=sfile, W=W, H=H, spp=a.spp, out=out, time=bool(a.time),
                                                               mode=MODES[mode][0], flags=MODES[mode][1])],
                           env=dict(os.environ, MC_LIB_PATH=os.path.abspath(lib)), capture_output=True, text=True)
        if r.returncode != 0:
            print(f"{os.path.basename(lib)}: FAILED {r.stderr[-400:]}", flush=True)
            continue
        z = np.load(out)
        print(f"{os.path.basename(lib)}, request: {mode}", flush=True)
        for k, nm in enumerate(names):
            ms = f"  {float(z[f'ms{k}']):.3f} ms / 100 spp at 900 x 600" if a.time else ""
            print(fmt(f"scene {nm} ({['strict', 'fast', 'careful'][int(z[f'kernel{k}']) // 10]} {int(z[f'kernel{k}']) % 10})", stats(z[f"tm{k}"], refs[k][0]), lin_rel(z[f"lin{k}"], refs[k][1])) + ms, flush=True)


if __name__ == "__main__":
    main()
