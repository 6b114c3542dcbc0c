"""Is K1 (fp32 Mandelbrot, 3200 x 2400, M = 1000: 0.20 ms, 3.4 SIMD cycles per VALU instruction against the two-float kernel's 2.07) short of
work or short of efficiency?  The same view rendered at 1x, 2x, 4x, 8x the height (same cost per row on average: the rows are resampled,
not extended): if the time per pixel falls with the size, what K1 loses is a FIXED cost — the launch ramp and the tail of the last
boundary tiles — that a 0.2 ms kernel cannot amortise; if it stays, the loop itself issues slowly.
    GPU box:  python tools/mandel_size_probe.py > gpurun_out/r05_mandel_size_probe.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

B = entry.load_package().bindings
ctx = B.Context(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
s = st.cuda_stream
W, M = 3200, 1000
print("# K1's view at 3200 x H, M = 1000; kernel time (HIP events, best of 3 x 20 launches), pixel-iterations of the reference algorithm")
base = None
for H in (2400, 4800, 9600, 19200):
    rg = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    it = torch.empty((H, W), dtype=torch.int32, device="cuda")
    p = B.mandelbrot_params(W, H, max_iter=M)
    for _ in range(3):
        ctx.mandelbrot_device(p, rg.data_ptr(), it.data_ptr(), stream=s)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ctx.mandelbrot_device(p, rg.data_ptr(), it.data_ptr(), stream=s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    i64 = it.to(torch.int64)
    pi = int(torch.where(i64 < M, i64 + 1, torch.full_like(i64, M)).sum().item())
    base = base or best / H
    print(f"H = {H:6d}: {best:8.4f} ms  {pi / (best * 1e-3):.3e} pixel-iters/s  {best / H * 1e6:7.3f} ns per row  ({best / H / base:.3f} of H = 2400's)", flush=True)
    del rg, it
