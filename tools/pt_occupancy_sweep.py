"""How sensitive is the (VALU-issue-bound) path tracer to resident waves per SIMD?  Pads the dynamic LDS of the
round-synchronous kernel (diagnostic library from `make pad`) so that 8, 6, 5, 4, 3, 2 blocks of 4 waves fit a CU and
times K2.  Decides how much LDS per wave the lane-regrouping kernel may spend on its queues."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os, torch
sys.path.insert(0, %r)
import __graft_entry__ as entry
B = entry.load_package().bindings
ctx = B.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
W, H, spp = 900, 600, 496
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
for mode, name in ((B.PT_MATH_FAST, "fast"), (B.PT_MATH_STRICT, "strict")):
    p = B.pathtrace_params(W, H, spp, math_mode=mode, flags=B.pt_force_s(16))
    for _ in range(2): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
    e1.record(); torch.cuda.synchronize()
    print(name, "%%.3f ms" %% (e0.elapsed_time(e1) / 4), flush=True)
""" % ROOT
lib = os.path.join(ROOT, "vulkan-compute-tests_amd", "lib", "libmc_compute_pad.so")
for pad_kb, waves in ((0, 8), (23, 6), (31, 5), (39, 4), (52, 3), (79, 2), (120, 1)):
    env = dict(os.environ, MC_LIB_PATH=lib, MC_PT_LDS_PAD=str(pad_kb * 1024))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print(f"pad {pad_kb:3d} KB (<= {waves} waves/SIMD):", " | ".join(r.stdout.split("\n")[:2]), r.stderr[-300:] if r.returncode else "", flush=True)
