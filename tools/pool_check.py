"""Sample-pool kernel against the round-synchronous closed-box kernel (same per-sample arithmetic, different order of the fp32
additions of a pixel): largest difference of the storage buffers, and the time of both, at a few sizes.  GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

B = entry.load_package().bindings
ctx = B.Context(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)


def render(W, H, spp, flags, reps=0, **kw):
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    p = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=flags, **kw)
    ctx.pathtrace_device(p, buf.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    ms = None
    if reps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ctx.pathtrace_device(p, buf.data_ptr(), stream=st.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    return buf.cpu().numpy(), ms


for (W, H, spp, reps) in [(8, 8, 16, 0), (33, 9, 37, 0), (64, 48, 100, 0), (300, 200, 64, 3), (900, 600, 500, 5), (1920, 1080, 256, 2)]:
    a, ta = render(W, H, spp, 0, reps)
    b, tb = render(W, H, spp, B.PT_NO_POOL_KERNEL, reps)
    d = np.abs(a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64))
    again, _ = render(W, H, spp, 0, 0)
    print(f"{W}x{H} spp {spp}: max |pool - rounds| = {d.max():.3e} (8-bit units), mean {d.mean():.3e}; nan {np.isnan(a).sum()}; "
          f"pool repeatable {np.array_equal(a.view(np.uint32), again.view(np.uint32))}; pool {ta} ms, rounds {tb} ms", flush=True)
