"""Fast-math tolerance at the headline size (K2: 900x600, 500 spp), per contraction variant of the fast kernel.

For each library given (default: the shipped lib plus lib/libmc_compute_c0.so / _c1.so from `make variants`) renders K2 in
MC_PT_MATH_FAST, compares the float storage buffer with the CPU oracle evaluated with libm (RMSE and the 99.9
percentile of the per-pixel RGB L2, 8-bit units) and times the kernel.  The yardstick SURVEY H5 names — the oracle's own
implementation-defined spread — is printed beside it (oracle mc-math vs oracle libm).

  python tools/fast_tolerance_k2.py [--spp 500] [--libs path,path,...]
Each library is loaded in its own child process (MC_LIB_PATH), so the variants never share a HIP module.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, {root!r})
import __graft_entry__ as entry
B = entry.load_package().bindings
W, H, spp = {W}, {H}, {spp}
ctx = B.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
p = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=int(os.environ.get("MC_PT_FLAGS", "0"), 0))
for _ in range(2): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
e1.record(); torch.cuda.synchronize()
np.save({out!r}, buf.cpu().numpy())
print("MS", e0.elapsed_time(e1) / 5)
"""


def stats(a, b):
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    l2 = np.sqrt((d ** 2).sum(-1))
    return dict(rmse=float(np.sqrt((d ** 2).mean())), p999=float(np.percentile(l2, 99.9)), max=float(l2.max()),
                frac_gt_half=float((np.abs(d).max(-1) > 0.5).mean()), mean_diff=float(d.mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--width", type=int, default=900)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--libs", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fast_tolerance_k2.json"))
    a = ap.parse_args()
    import __graft_entry__ as entry
    O = entry.load_oracle()
    libdir = os.path.join(ROOT, "vulkan-compute-tests_amd", "lib")
    libs = a.libs.split(",") if a.libs else [os.path.join(libdir, n) for n in
                                             ("libmc_compute.so", "libmc_compute_c1.so", "libmc_compute_c0.so")]
    W, H, spp = a.width, a.height, a.spp
    t = time.perf_counter()
    ref = O.pathtrace(W, H, spp, math_mode=O.MATH_LIBM)
    print(f"oracle libm {time.perf_counter() - t:.1f} s", flush=True)
    t = time.perf_counter()
    ref_mc = O.pathtrace(W, H, spp, math_mode=O.MATH_MC)
    print(f"oracle mc   {time.perf_counter() - t:.1f} s", flush=True)
    res = {"config": [W, H, spp], "yardstick_oracle_mc_vs_libm": stats(ref_mc, ref)}
    print("yardstick (oracle mc vs libm):", res["yardstick_oracle_mc_vs_libm"], flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    for lib in libs:
        if not os.path.exists(lib):
            print("skip (not built):", lib)
            continue
        out = os.path.join("/tmp", os.path.basename(lib) + ".npy")
        env = dict(os.environ, MC_LIB_PATH=lib)
        r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, W=W, H=H, spp=spp, out=out)], env=env,
                           capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(lib, "FAILED", r.stderr[-2000:])
            continue
        ms = float([l for l in r.stdout.splitlines() if l.startswith("MS")][0].split()[1])
        img = np.load(out)
        s = stats(img, ref)
        s["vs_strict_oracle"] = stats(img, ref_mc)
        s["kernel_ms"] = ms
        res[os.path.basename(lib)] = s
        print(os.path.basename(lib), json.dumps(s), flush=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
