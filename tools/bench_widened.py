#!/usr/bin/env python3
"""Throughput of the rows SURVEY §8(f) widens into (2: the extended-precision sphere branches on the sphere-walled scene, 3: progressive
sample ranges, 4: general scenes through the LDS-resident generic kernel), measured the way bench.py measures the headline: outputs
resident in HBM, kernels on a torch stream, HIP events around a batch of launches, the CPU oracle on a bounded sample beside it where it
takes seconds.  One JSON line per case (profiles/r03_bench_widened.jsonl).  Every case renders 900 x 600 like K2 (spp stated per case).

    python tools/bench_widened.py [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402


def room(rng, n_spheres, n_lights):
    """The reference's six walls given in an order the slab analysis does not accept (so the GENERIC kernel runs) + random spheres."""
    O = entry.load_oracle()
    planes = O.DEFAULT_PLANES.copy().reshape(6, 12)[[2, 3, 0, 1, 4, 5]]
    spheres = np.zeros((n_spheres, 12), np.float32)
    spheres[:, 0] = rng.uniform(-2.2, 2.2, n_spheres); spheres[:, 1] = rng.uniform(-1.8, 1.2, n_spheres)
    spheres[:, 2] = rng.uniform(-2.4, 2.5, n_spheres); spheres[:, 3] = rng.uniform(0.05, 0.35, n_spheres)
    spheres[:, 8:11] = rng.uniform(0.2, 0.95, (n_spheres, 3)); spheres[:, 11] = rng.choice([1, 1, 1, 2, 3], n_spheres)
    lights = rng.choice(n_spheres, n_lights, replace=False)
    spheres[lights, 4:7] = rng.uniform(20, 80, (n_lights, 3)); spheres[lights, 8:11] = 0; spheres[lights, 11] = 1
    spheres[lights, 1] = rng.uniform(1.2, 1.7, n_lights); spheres[lights, 3] = 0.15
    return planes.astype(np.float32), spheres


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="", help="comma-separated rows to run (f2,f3,f4,f4box); default all")
    a = ap.parse_args()
    only = set(x for x in a.only.split(",") if x)
    B, O = entry.load_package().bindings, entry.load_oracle()
    ctx = B.Context(0)
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    W, H = 900, 600
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")

    def timed(launches, reps=a.reps):
        """launches(): enqueue one whole render; returns the best mean ms over 3 batches of `reps`."""
        launches(); torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                launches()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        return best

    def cpu(spp_sample, rows, **kw):
        t0 = time.time()
        O.pathtrace(W, H, kw.pop("spp"), sample_begin=0, sample_end=spp_sample, row_begin=0, row_end=rows, **kw)
        dt = time.time() - t0
        return W * rows * spp_sample / dt, dt

    def emit(row, name, spp, ms, note, cpu_rate=None, cpu_note=None):
        rec = {"row": row, "case": name, "image": [W, H], "spp": spp, "ms_per_render": round(ms, 3), "samples_per_s": W * H * spp / (ms * 1e-3),
               "note": note}
        if cpu_rate:
            rec["cpu_oracle_samples_per_s"] = cpu_rate
            rec["cpu_note"] = cpu_note
        print(json.dumps(rec), flush=True)

    # ---- (f)2: the sphere-walled scene of TEST_PRECISION_WITH_LARGE_SPHERE_WALLS through the extended-precision sphere tests
    LP, LS = O.LARGE_SPHERE_PLANES, O.LARGE_SPHERE_SPHERES
    spp = 100
    for prec, pname in () if only and "f2" not in only else ((B.PT_PREC_F32, "fp32 (the reference's default build)"), (B.PT_PREC_FP64, "native fp64 (:132-143)"),
                        (B.PT_PREC_DS, "DS_f32_f32 (:144-213)"), (B.PT_PREC_DF64, "DF64_F32_F32 (:214-256)")):
        for mode, mname in ((B.PT_MATH_STRICT, "strict"), (B.PT_MATH_FAST, "fast")):
            p = B.pathtrace_params(W, H, spp, math_mode=mode, flags=B.pt_precision(prec))
            ms = timed(lambda: ctx.pathtrace_device(p, buf.data_ptr(), planes=LP, spheres=LS, stream=s))
            c = None
            if mode == B.PT_MATH_STRICT:
                c = cpu(2, 64, spp=spp, planes=LP, spheres=LS, math_mode=O.MATH_MC, precision=prec)
            emit("f2", f"sphere-walled scene, sphere test {pname}, {mname}", spp, ms,
                 "1 plane + 9 spheres (six of radius 1e5): generic kernel with the run-time precision switch",
                 c[0] if c else None, f"oracle on the host cores, 2 samples x 64 rows ({c[1]:.1f} s)" if c else None)
    # ---- (f)3: progressive sample ranges (the samps.x protocol), K2 scene, both math modes (fast ranges run the pool kernel since round 4)
    spp = 500
    for mode, mname in () if only and "f3" not in only else ((B.PT_MATH_STRICT, "strict"), (B.PT_MATH_FAST, "fast")):
        whole = B.pathtrace_params(W, H, spp, math_mode=mode)
        ms_whole = timed(lambda: ctx.pathtrace_device(whole, buf.data_ptr(), stream=s))
        emit("f3", f"K2 {mname}, one launch", spp, ms_whole, f"{mname} sample-pool kernel")
        for parts in (5, 25):
            cuts = [round(i * spp / parts) for i in range(parts + 1)]
            ps = [B.pathtrace_params(W, H, spp, math_mode=mode, sample_begin=b, sample_end=e) for b, e in zip(cuts[:-1], cuts[1:])]
            assert all(B.pathtrace_select_kernel(q).kernel == B.PT_KERNEL_POOL for q in ps)

            def run():
                for q in ps:
                    ctx.pathtrace_device(q, buf.data_ptr(), stream=s)
            ms = timed(run)
            emit("f3", f"K2 {mname} in {parts} ranges of {spp // parts} samples (accumulator continued in the buffer)", spp, ms,
                 f"{mname} sample-pool kernel per range; {ms / ms_whole:.3f} x the one-launch time")
    # ---- (f)4, the specialised kernels beyond three spheres (round 4): the reference box with 1 .. 8 disjoint spheres takes the
    # slab / closed-box / sample-pool kernels instantiated per sphere count; beside each the GENERIC kernel on the same scene
    if not only or "f4box" in only:
        from test_gpu_scenes import box_scene
        spp = 100
        base = {}
        for ns, lights in ((3, 1), (1, 1), (2, 1), (4, 1), (5, 1), (5, 2), (8, 1)):
            if ns == 3:
                planes, spheres = O.DEFAULT_PLANES.copy().reshape(6, 12), O.DEFAULT_SPHERES.copy().reshape(3, 12)
            else:
                planes, spheres = box_scene(O, ns, np.random.default_rng(40 + 10 * ns + lights), lights)
            for mode, mname in ((B.PT_MATH_STRICT, "strict"), (B.PT_MATH_FAST, "fast")):
                p = B.pathtrace_params(W, H, spp, math_mode=mode)
                assert B.pathtrace_select_kernel(p, planes, spheres).kernel == B.PT_KERNEL_POOL
                ms = timed(lambda: ctx.pathtrace_device(p, buf.data_ptr(), planes=planes, spheres=spheres, stream=s))
                g = B.pathtrace_params(W, H, spp, math_mode=mode, flags=B.PT_GENERIC_KERNEL)
                ms_g = timed(lambda: ctx.pathtrace_device(g, buf.data_ptr(), planes=planes, spheres=spheres, stream=s))
                if ns == 3:
                    base[mname] = ms
                emit("f4box", f"the box with {ns} spheres ({lights} light{'s' if lights > 1 else ''}), {mname}: sample-pool kernel", spp, ms,
                     f"{ms / base[mname]:.2f} x the reference scene's (3 spheres) time; the generic kernel on the same scene: {ms_g:.3f} ms ({ms_g / ms:.2f} x)")
    # ---- (f)4: general scenes through the generic (LDS-resident scene) kernel
    rng = np.random.default_rng(5)
    spp = 100
    DP, DS = O.DEFAULT_PLANES.copy().reshape(6, 12), O.DEFAULT_SPHERES.copy().reshape(3, 12)
    for name, planes, spheres in () if only and "f4" not in only else (("the default scene with its planes permuted (generic kernel on K2's geometry)", DP[[2, 3, 0, 1, 4, 5]], DS),
                                  ("6 planes + 64 spheres", *room(rng, 64, 3)), ("6 planes + 256 spheres", *room(rng, 256, 4)),
                                  ("6 planes + 512 spheres", *room(rng, 512, 5)), ("6 planes + 1500 spheres", *room(rng, 1500, 8)),
                                  ("6 planes + 6000 spheres (beyond the LDS store)", *room(rng, 6000, 8))):
        assert B.pathtrace_scene_class(planes, spheres) == 0
        for mode, mname in ((B.PT_MATH_STRICT, "strict"), (B.PT_MATH_FAST, "fast")):
            for where, flag in (("LDS", B.PT_SCENE_IN_LDS), ("memory", B.PT_SCENE_IN_MEMORY)):
                if len(spheres) > 3000 and where == "LDS":
                    continue
                p = B.pathtrace_params(W, H, spp, math_mode=mode, flags=flag)
                reps = 1 if len(spheres) > 1000 else max(1, a.reps // 2)
                ms = timed(lambda: ctx.pathtrace_device(p, buf.data_ptr(), planes=planes, spheres=spheres, stream=s), reps=reps)
                c = None
                if mode == B.PT_MATH_STRICT and where == "memory":
                    rows = 64 if len(spheres) <= 64 else (8 if len(spheres) <= 1500 else 2)
                    c = cpu(2, rows, spp=spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
                emit("f4", f"{name}, {mname}, records in {where}", spp, ms, f"{len(planes)} planes + {len(spheres)} spheres through the generic kernel",
                     c[0] if c else None, f"oracle on the host cores, 2 samples x {rows} rows ({c[1]:.1f} s)" if c else None)
    ctx.close()


if __name__ == "__main__":
    main()
