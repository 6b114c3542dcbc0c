"""Sweeps the path tracer's launch variants on K2 (sample-parallel width S, slab vs generic kernel, math mode)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
B = entry.load_package().bindings
ctx = B.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
W, H, spp = 900, 600, 500
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
def run(name, p, reps=5):
    for _ in range(2): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:40s} {ms:8.3f} ms  {W*H*spp/ms/1e6:.3f}e9 samples/s", flush=True)
for mode, mname in ((B.PT_MATH_FAST, "fast"), (B.PT_MATH_STRICT, "strict")):
    for S in (1, 4, 16):
        run(f"{mname} slab S={S}", B.pathtrace_params(W, H, spp, math_mode=mode, flags=B.pt_force_s(S)))
    run(f"{mname} generic(LDS scene) S=16", B.pathtrace_params(W, H, spp, math_mode=mode, flags=B.pt_force_s(16) | B.PT_GENERIC_KERNEL))
