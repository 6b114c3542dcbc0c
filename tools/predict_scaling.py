"""What one rank of an N-GPU run renders, timed on ONE GPU: the per-rank kernel time of BASELINE's strong-scaling configurations (K3: path
trace 3840x2560x4096, K4: two-float Mandelbrot 7680x5120 M = 50 000) for N = 1, 2, 4, 8 — rank 0's interleaved 8-row blocks, exactly the
tile bench.py gives it.  T(1) / (N * T(N)) is the scaling efficiency the render itself allows (exchange excluded).
    python tools/predict_scaling.py [K3|K4|K2 ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402  (CONFIGS, K4_VIEW)

pkg = entry.load_package()
B, S = pkg.bindings, pkg.sharding
ctx = B.Context(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
for name in (sys.argv[1:] or ["K4", "K3"]):
    cfg = bench.CONFIGS[name]
    weak = cfg["scaling"] == "weak"        # weak: the image grows with N (H * N rows), every rank keeps its N = 1 share
    base = None
    for n in (1, 2, 4, 8):
        W, H = cfg["W"], cfg["H"] * (n if weak else 1)
        worst = 0.0
        for rank in sorted({0, n - 1, n // 2}):
            if cfg["kind"] == "pt":
                p = S.shard(B.pathtrace_params(W, H, cfg["spp"], math_mode=B.PT_MATH_FAST), rank, n)
            else:
                kw = dict(max_iter=cfg["M"])
                if cfg["ds"]:
                    kw.update(precision=B.PRECISION_DS, centre=bench.K4_VIEW["centre"], scale=bench.K4_VIEW["scale"])
                p = S.shard(B.mandelbrot_params(W, H, **kw), rank, n)
                p.flags |= B.MANDEL_ITERS_U16 if p.max_iter <= 65535 else 0
            rows = B.tile_rows(p)
            if cfg["kind"] == "pt":
                buf = torch.empty((rows, W, 4), dtype=torch.float32, device="cuda")
                run = lambda: ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
            else:
                buf = torch.empty((rows, W), dtype=torch.int16, device="cuda")
                run = lambda: ctx.mandelbrot_device(p, 0, buf.data_ptr(), stream=s)
            run(); torch.cuda.synchronize()
            reps = 1 if cfg["kind"] == "pt" else 3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            worst = max(worst, e0.elapsed_time(e1) / reps)
        base = base or worst
        eff = base / worst if weak else base / (n * worst)
        print(f"{name}  N={n}  slowest of ranks 0 / N/2 / N-1: {worst:10.2f} ms   {'T1 / T_N' if weak else 'T1 / (N T_N)'} = {eff:.3f}", flush=True)
