#!/usr/bin/env python3
"""Where does the sign-consistent mean difference of the fast kernels come from (+0.005 of 255 at 8 spheres, 6 to 10
standard errors, against +-0.0005 on the reference scene)?  Per-SAMPLE comparison with the oracle (libm): every sample s of every pixel
is rendered alone (sample range [s, s + 1) of a 500-spp render: the buffer then holds accrad_s / spp, linear, no tonemap) by the fast
kernel and by the oracle, and the samples are split into
    same path : the two radiances agree to 1e-3 relative (rounding of the same path's arithmetic) — their summed difference is the
                SMOOTH part of the bias, a systematic shift of every sample;
    forked    : the sample took another path — counted by sign, with the radiance gained and lost.
The relative bias of the image's linear radiance is the sum of the two parts; the table says which one carries it.

    python tools/fork_bias.py [--scene 8:1] [--samples 64] [--size 300 200] lib1.so ...      (one child process per library)"""
import argparse
import os
import subprocess
import sys
import tempfile

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
z = np.load({scene!r})
W, H, spp, K = {W}, {H}, {spp}, {K}
out = np.empty((K, H, W, 3), np.float32)
with B.Context(0) as ctx:
    for s in range(K):
        q = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, sample_begin=s, sample_end=s + 1)
        out[s] = ctx.pathtrace(q, planes=z["planes"], spheres=z["spheres"], acc=np.zeros((H, W, 4), np.float32))[..., :3]
np.save({out!r}, out)
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--scene", default="8:1")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--size", type=int, nargs=2, default=[300, 200])
    a = ap.parse_args()
    from test_gpu_scenes import box_scene
    O = entry.load_oracle()
    W, H = a.size
    spp, K = 500, a.samples
    if a.scene == "ref":
        pl, sp = O.DEFAULT_PLANES.copy().reshape(6, 12), O.DEFAULT_SPHERES.copy().reshape(3, 12)
    else:
        n, l = (int(x) for x in a.scene.split(":")[:2])
        pl, sp = box_scene(O, n, np.random.default_rng(40 + 10 * n + l), l)
    tmp = tempfile.mkdtemp(prefix="fork_bias_")
    sfile = os.path.join(tmp, "scene.npz")
    np.savez(sfile, planes=pl, spheres=sp)
    ref = np.empty((K, H, W, 3), np.float32)
    mc = np.empty((K, H, W, 3), np.float32)
    for s in range(K):
        ref[s] = O.pathtrace(W, H, spp, planes=pl, spheres=sp, math_mode=O.MATH_LIBM, sample_begin=s, sample_end=s + 1)[..., :3]
        mc[s] = O.pathtrace(W, H, spp, planes=pl, spheres=sp, math_mode=O.MATH_MC, sample_begin=s, sample_end=s + 1)[..., :3]
    print(f"# scene {a.scene}, {W} x {H}, samples 0..{K - 1} of {spp}, each alone: {K * W * H} samples; radiance in units of the image's mean sample", flush=True)

    def report(name, img):
        r = ref.astype(np.float64).sum(-1) * spp          # accrad (RGB sum) per sample
        f = img.astype(np.float64).sum(-1) * spp
        unit = r.mean()
        d = f - r
        same = np.abs(d) <= 1e-3 * np.abs(r) + 1e-9
        fork = ~same
        gain, loss = fork & (d > 0), fork & (d < 0)
        big = 20.0 * unit
        print(f"{name}\n  same path: {same.mean() * 100:.4f} % of the samples, summed difference {d[same].sum() / r.sum():+.2e} of the image's radiance "
              f"(mean relative difference of a sample {np.mean(d[same & (r > 0)] / r[same & (r > 0)]):+.2e})\n"
              f"  forked   : {fork.mean() * 100:.4f} % — {gain.sum()} gain {d[gain].sum() / r.sum():+.2e}, {loss.sum()} lose {d[loss].sum() / r.sum():+.2e}; "
              f"net {d[fork].sum() / r.sum():+.2e}\n"
              f"             of them by more than 20 mean samples (a light seen through specular surfaces): {int((gain & (d > big)).sum())} gain "
              f"{d[gain & (d > big)].sum() / r.sum():+.2e}, {int((loss & (d < -big)).sum())} lose {d[loss & (d < -big)].sum() / r.sum():+.2e}\n"
              f"  total    : {d.sum() / r.sum():+.2e}  (standard error of the forked sum {np.sqrt((d[fork] ** 2).sum()) / r.sum():.1e})", flush=True)

    report("oracle mc math (yardstick)", mc)
    for lib in a.libs:
        out = os.path.join(tmp, os.path.basename(lib) + ".npy")
        r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, scene=sfile, W=W, H=H, spp=spp, K=K, out=out)],
                           env=dict(os.environ, MC_LIB_PATH=os.path.abspath(lib)), capture_output=True, text=True)
        if r.returncode != 0:
            print(f"{os.path.basename(lib)}: FAILED {r.stderr[-400:]}", flush=True)
            continue
        report(os.path.basename(lib), np.load(out))


if __name__ == "__main__":
    main()
