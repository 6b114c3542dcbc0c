"""Upper bound of what a cost-aware dispatch order could give the fp32 Mandelbrot (K1): the tiles are dispatched in descending
order of their TRUE cost (sum of iterations of the tile, from a first render) and the kernel is timed against the natural order.
GPU box.  Experiment hook: mc_debug_mandelbrot_tile_order."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

B = entry.load_package().bindings
L = B.lib()
L.mc_debug_mandelbrot_tile_order.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
ctx = B.Context(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
W, H, M = 3200, 2400, 1000
rg = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
it = torch.empty((H, W), dtype=torch.int32, device="cuda")
p = B.mandelbrot_params(W, H, max_iter=M)


def timed(name, reps=30):
    for _ in range(3):
        ctx.mandelbrot_device(p, rg.data_ptr(), it.data_ptr(), stream=s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ctx.mandelbrot_device(p, rg.data_ptr(), it.data_ptr(), stream=s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    print(f"{name:46s} {best:.4f} ms", flush=True)
    return it.clone()


ref = timed("natural order (shipped)")
tiles = ref.reshape(H // 8, 8, W // 8, 8).to(torch.int64)
cost = tiles.amax(dim=(1, 3)).reshape(-1)                  # a wave runs as long as its slowest pixel
order = torch.argsort(cost, descending=True, stable=True).to(torch.int32).contiguous()
nat = torch.arange(cost.numel(), dtype=torch.int32, device="cuda")
for name, o in (("natural order through the order table", nat), ("descending true tile cost (oracle order)", order),
                ("ascending true tile cost", torch.flip(order, dims=(0,)).contiguous())):
    L.mc_debug_mandelbrot_tile_order(ctx._h, o.data_ptr(), o.numel())
    got = timed(name)
    assert torch.equal(got, ref), name
L.mc_debug_mandelbrot_tile_order(ctx._h, None, 0)
# a cheap predictor: the cost of the tile's centre pixel at a low iteration limit
print("tiles running to the limit:", int((cost >= M).sum()), "of", cost.numel())
