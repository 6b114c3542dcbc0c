"""Full BASELINE config K4 on one GPU: 7680x5120, M=50000, two-float, frozen view.  Prints the pixel-iter sum."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
B = entry.load_package().bindings
ctx = B.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
W, H, M = 7680, 5120, 50000
p = B.mandelbrot_params(W, H, max_iter=M, precision=B.PRECISION_DS, centre=(-0.7436438870371587, 0.13182590420531198),
                        scale=(1e-8, 1e-8 * 2.0 / 3.0))
it = torch.zeros((H, W), dtype=torch.int32, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.mandelbrot_device(p, 0, it.data_ptr(), stream=st.cuda_stream); torch.cuda.synchronize()
    dt = time.perf_counter() - t
i64 = it.to(torch.int64)
pi = int(torch.where(i64 < M, i64 + 1, torch.full_like(i64, M)).sum().item())
print(f"K4 {W}x{H} M{M} ds: {dt:.4f} s  pixel_iters {pi}  mean {pi/(W*H):.1f}/px  max n {int(i64.max())}  interior {int((i64==M).sum())} px  "
      f"{pi/dt:.4g} pixel-iters/s  {pi*142/dt/1e12:.2f} TFLOP/s  checksum {int((i64 * (torch.arange(W, device='cuda') % 251 + 1)).sum().item())}")
