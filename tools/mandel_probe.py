"""Mandelbrot loop-efficiency probe: all-interior view (no divergence), K1 view, iteration plane only vs both outputs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
B = entry.load_package().bindings
ctx = B.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
W, H, M = 3200, 2400, 1000
rg = torch.empty((H, W, 4), dtype=torch.float32, device="cuda"); it = torch.empty((H, W), dtype=torch.int32, device="cuda")
def run(name, p, d_rgba, d_it, reps=20):
    for _ in range(3): ctx.mandelbrot_device(p, d_rgba, d_it, stream=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ctx.mandelbrot_device(p, d_rgba, d_it, stream=s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ctx.mandelbrot_device(p, 0, it.data_ptr(), stream=s); torch.cuda.synchronize()
    i64 = it.to(torch.int64); pi = int(torch.where(i64 < p.max_iter, i64 + 1, torch.full_like(i64, p.max_iter)).sum().item())
    print(f"{name:34s} {ms:.4f} ms  {pi/ms/1e9:.3f}e12 pixel-iters/s  lane-op frac {pi*8/ms/1e9/78.6:.3f}  cycles/wave-iter@2.34GHz {1024*64*2.34e9*ms*1e-3/pi:.2f}")
run("interior view, both outputs", B.mandelbrot_params(W, H, max_iter=M, centre=(-0.2, 0.0), scale=(0.1, 0.1)), rg.data_ptr(), it.data_ptr())
run("interior view, iters only", B.mandelbrot_params(W, H, max_iter=M, centre=(-0.2, 0.0), scale=(0.1, 0.1)), 0, it.data_ptr())
run("interior view M=10000 iters only", B.mandelbrot_params(W, H, max_iter=10000, centre=(-0.2, 0.0), scale=(0.1, 0.1)), 0, it.data_ptr(), reps=5)
run("K1 view, both outputs", B.mandelbrot_params(W, H, max_iter=M), rg.data_ptr(), it.data_ptr())
run("K1 view, iters only", B.mandelbrot_params(W, H, max_iter=M), 0, it.data_ptr())
run("K1 view, rgba only", B.mandelbrot_params(W, H, max_iter=M), rg.data_ptr(), 0)
rg2 = torch.empty((2000, 2000, 4), dtype=torch.float32, device="cuda"); it2 = torch.empty((2000, 2000), dtype=torch.int32, device="cuda")
it = it2
W, H = 2000, 2000
run("reference default 2000x2000 M=128", B.mandelbrot_params(2000, 2000, max_iter=128), rg2.data_ptr(), it2.data_ptr(), reps=50)
W, H = 3200, 2400
it = torch.empty((H, W), dtype=torch.int32, device="cuda")
run("exterior view (c far away)", B.mandelbrot_params(W, H, max_iter=M, centre=(3.0, 3.0), scale=(0.1, 0.1)), rg.data_ptr(), it.data_ptr())
