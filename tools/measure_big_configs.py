"""Times the large BASELINE configs on ONE GPU (K4 needs 8 GPUs to be quick; here it just has to finish):
K3's per-GPU tile (3840 x 320 rows of the 3840x2560 image, 4096 spp) and K4 (7680x5120, M=50000, two-float) at
1/8 of the rows (interleaved tile of rank 0 of 8) — prints samples/s, pixel-iters and seconds."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); B, S = pkg.bindings, pkg.sharding
ctx = B.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); stream = st.cuda_stream
which = sys.argv[1:] or ["k3", "k4"]
if "k3" in which:
    W, H, spp = 3840, 2560, int(os.environ.get("K3_SPP", "4096"))
    p = S.shard(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST), 0, 8)
    rows = B.tile_rows(p)
    tile = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.pathtrace_device(p, tile.data_ptr(), stream=stream); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"K3 rank-0 tile {W}x{rows} spp{spp}: {dt:.3f} s  {W*rows*spp/dt:.4g} samples/s", flush=True)
if "k4" in which:
    W, H, M = 7680, 5120, int(os.environ.get("K4_M", "50000"))
    p = S.shard(B.mandelbrot_params(W, H, max_iter=M, precision=B.PRECISION_DS,
                                    centre=(-0.7436438870371587, 0.13182590420531198), scale=(1e-8, 1e-8 * 2.0 / 3.0)), 0, 8)
    rows = B.tile_rows(p)
    it = torch.zeros((rows, W), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.mandelbrot_device(p, 0, it.data_ptr(), stream=stream); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    i64 = it.to(torch.int64)
    pi = int(torch.where(i64 < M, i64 + 1, torch.full_like(i64, M)).sum().item())
    print(f"K4 rank-0 tile {W}x{rows} M{M}: {dt:.3f} s  pixel-iters {pi}  {pi/dt:.4g} pixel-iters/s  interior {float((i64==M).float().mean()):.4f}", flush=True)
