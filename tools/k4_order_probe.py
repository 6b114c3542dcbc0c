"""K4 at N = 8: what a long-tiles-first dispatch order gives the two-float Mandelbrot tile ONE rank of eight renders
(7680 x 640 rows in interleaved 8-row blocks, M = 50 000).  The kernel's tail is its longest tile running alone at the end
(9068 iterations against a median of ~1000): 7.8 ms where 57.0 / 8 = 7.1 would be linear scaling (profiles/r03_predict_scaling.txt).
Orders tried through the experiment hook mc_debug_mandelbrot_tile_order: natural; descending TRUE tile cost (the upper bound, from a
finished render); and the realisable predictor — the tile's CENTRE pixel iterated to a low cap, "reached the cap" = long, long tiles
first (the cap's pre-pass is priced separately).  GPU box:  python tools/k4_order_probe.py [N]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402

pkg = entry.load_package()
B, S = pkg.bindings, pkg.sharding
L = B.lib()
L.mc_debug_mandelbrot_tile_order.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
ctx = B.Context(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = bench.CONFIGS["K4"]
W, H, M = cfg["W"], cfg["H"], cfg["M"]
kw = dict(max_iter=M, precision=B.PRECISION_DS, centre=bench.K4_VIEW["centre"], scale=bench.K4_VIEW["scale"])
for rank in (0, N - 1):
    p = S.shard(B.mandelbrot_params(W, H, **kw), rank, N)
    rows = B.tile_rows(p)
    it = torch.empty((rows, W), dtype=torch.int32, device="cuda")

    def timed(name, reps=5):
        for _ in range(2):
            ctx.mandelbrot_device(p, 0, it.data_ptr(), stream=s)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ctx.mandelbrot_device(p, 0, it.data_ptr(), stream=s)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        print(f"rank {rank} of {N}: {name:58s} {best:8.3f} ms", flush=True)
        return it.clone()

    ref = timed("natural order (shipped)")
    tiles = ref.reshape(rows // 8, 8, W // 8, 8).to(torch.int64)
    cost = tiles.amax(dim=(1, 3)).reshape(-1)                       # a wave runs as long as its slowest pixel
    centre = tiles[:, 4, :, 4].reshape(-1)
    print(f"   tiles {cost.numel()}, tile max: median {int(cost.median())}, p99 {int(cost.float().quantile(0.99))}, max {int(cost.max())}; "
          f"sum of tile maxima / sum of pixels' counts = {float(cost.sum() * 64) / float(tiles.sum()):.3f}")
    orders = [("descending true tile cost (upper bound)", torch.argsort(cost, descending=True, stable=True))]
    for cap in (1024, 1536, 2048, 3072):
        long_first = torch.argsort((centre >= cap).to(torch.int32), descending=True, stable=True)
        n_long = int((centre >= cap).sum())
        miss = int(((cost >= 2 * cap) & (centre < cap)).sum())
        orders.append((f"centre pixel >= {cap} first ({n_long} tiles; {miss} tiles >= {2 * cap} missed)", long_first))
    for name, o in orders:
        o = o.to(torch.int32).contiguous()
        L.mc_debug_mandelbrot_tile_order(ctx._h, o.data_ptr(), o.numel())
        got = timed(name)
        assert torch.equal(got, ref), name
    L.mc_debug_mandelbrot_tile_order(ctx._h, None, 0)
    # what the pre-pass would cost: one lane per tile, the centre pixels only, capped — timed as a render of a (W / 8) x (rows / 8)
    # image of the same view at max_iter = cap (same pixel pitch x 8: the same orbits' statistics, one pixel per tile)
    for cap in (1024, 2048):
        q = B.mandelbrot_params(W // 8, H // 8, **dict(kw, max_iter=cap))
        q = S.shard(q, rank, N)
        small = torch.empty((B.tile_rows(q), W // 8), dtype=torch.int32, device="cuda")
        for _ in range(2):
            ctx.mandelbrot_device(q, 0, small.data_ptr(), stream=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ctx.mandelbrot_device(q, 0, small.data_ptr(), stream=s)
        e1.record(); torch.cuda.synchronize()
        print(f"   pre-pass stand-in ({W // 8} x {B.tile_rows(q)} pixels, cap {cap}): {e0.elapsed_time(e1) / 5:.3f} ms")
