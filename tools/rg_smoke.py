"""First-contact check of the lane-regrouping kernel: small strict renders against the oracle, each announced before it
starts (a hang shows where), then K2 timings.  Run under `timeout`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MC_LIB_PATH", os.path.join(ROOT, "vulkan-compute-tests_amd", "lib", "libmc_compute_regroup.so"))   # diagnostic library
import __graft_entry__ as entry
B = entry.load_package().bindings
O = entry.load_oracle()
ctx = B.Context(0)
def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
bad = 0
for (W, H, spp, depth) in [(4, 4, 1, 12), (4, 4, 16, 12), (8, 8, 40, 12), (24, 16, 64, 12), (33, 9, 37, 12), (48, 32, 130, 12), (24, 16, 70, 7), (24, 16, 33, 2), (16, 16, 600, 12)]:
    print(f"render {W}x{H} spp {spp} depth {depth} ...", end=" ", flush=True)
    t = time.time()
    out = ctx.pathtrace(B.pathtrace_params(W, H, spp, max_depth=depth, flags=B.PT_KERNEL_REGROUP))
    ref = O.pathtrace(W, H, spp, math_mode=O.MATH_MC, max_depth=depth)
    ok = np.array_equal(bits(out), bits(ref))
    nd = int((bits(out) != bits(ref)).any(-1).sum())
    print("OK" if ok else f"MISMATCH ({nd} pixels)", f"{time.time() - t:.2f}s", flush=True)
    bad += not ok
if "--time" in sys.argv:
    import torch
    st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
    W, H, spp = 900, 600, 500
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    for mode, name in ((B.PT_MATH_FAST, "fast"), (B.PT_MATH_STRICT, "strict")):
        for flags, fname in ((0, "rounds"), (B.PT_KERNEL_REGROUP, "regroup")):
            p = B.pathtrace_params(W, H, spp, math_mode=mode, flags=flags)
            for _ in range(2): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
            e1.record(); torch.cuda.synchronize()
            ctx.synchronize()
            print(f"K2 {name} {fname}: {e0.elapsed_time(e1) / 4:.3f} ms", flush=True)
sys.exit(1 if bad else 0)
