"""Times the K1 Mandelbrot (3200x2400, M = 1000, fp32; vec4 + counts written) with each library given (MC_LIB_PATH, one child each)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os, torch
sys.path.insert(0, %r)
import __graft_entry__ as entry
B = entry.load_package().bindings
ctx = B.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
W, H, M = 3200, 2400, 1000
rg = torch.empty((H, W, 4), dtype=torch.float32, device="cuda"); it = torch.empty((H, W), dtype=torch.int32, device="cuda")
p = B.mandelbrot_params(W, H, max_iter=M)
for mode, (a, b), fl in (("vec4 + counts", (rg.data_ptr(), it.data_ptr()), 0), ("counts only", (0, it.data_ptr()), 0),
                         ):
    p.flags = fl
    for _ in range(3): ctx.mandelbrot_device(p, a, b, stream=s)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ctx.mandelbrot_device(p, a, b, stream=s)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30)
    print("MS %%-36s %%.4f  checksum %%d" %% (mode, best, int(it.sum())))
""" % ROOT
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MC_LIB_PATH=os.path.abspath(lib)), capture_output=True, text=True)
    print(os.path.basename(lib))
    lines = [l for l in r.stdout.splitlines() if l.startswith("MS")]
    print("\n".join(lines) if lines else "FAILED " + r.stderr[-600:], flush=True)
