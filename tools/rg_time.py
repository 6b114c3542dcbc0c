"""Times K2 with the rounds and the regroup kernels (fast math); used under rocprofv3 by tools/rg_prof.sh."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MC_LIB_PATH", os.path.join(ROOT, "vulkan-compute-tests_amd", "lib", "libmc_compute_regroup.so"))   # diagnostic library
import __graft_entry__ as entry
B = entry.load_package().bindings
ctx = B.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
W, H, spp = 900, 600, int(os.environ.get("RG_SPP", "496"))
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
modes = [(B.PT_MATH_FAST, "fast")] + ([(B.PT_MATH_STRICT, "strict")] if "--strict" in sys.argv else [])
for mode, name in modes:
    for flags, fname in ((0, "rounds"), (B.PT_KERNEL_REGROUP, "regroup")):
        p = B.pathtrace_params(W, H, spp, math_mode=mode, flags=flags)
        ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
        e1.record(); torch.cuda.synchronize()
        ctx.synchronize()
        print(f"K2 {name} {fname}: {e0.elapsed_time(e1) / 3:.3f} ms", flush=True)
