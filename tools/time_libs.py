"""Times the K2 fast path trace (900x600, 500 spp) with each library given (MC_LIB_PATH, one child process per library).
  python tools/time_libs.py lib1.so lib2.so ...   [MC_PT_FLAGS=<n> applies to all; MC_TIME_MATH=strict times the strict kernel]"""
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
W, H, spp = 900, 600, 500
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
mode = B.PT_MATH_STRICT if os.environ.get("MC_TIME_MATH") == "strict" else B.PT_MATH_FAST
p = B.pathtrace_params(W, H, spp, math_mode=mode, flags=int(os.environ.get("MC_PT_FLAGS", "0"), 0))
for _ in range(3): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print("MS %%.4f  mean %%.6f" %% (best, float(buf[..., :3].double().mean())))
""" % ROOT
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MC_LIB_PATH=os.path.abspath(lib)), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("MS")]
    print(f"{os.path.basename(lib):40s} {line[0] if line else 'FAILED ' + r.stderr[-500:]}", flush=True)
