"""Where does the FIRST launch of a fresh process lose its time (the apps' cold K2 kernel reads 28 - 30 ms against 14.0 steady,
profiles/r05_end_to_end.txt)?  Fresh child processes, host wall time of blocking launches through the C ABI:
   A: K2 fast (900 x 600 x 500) twice;   B: a small Mandelbrot first (another translation unit's code object), then K2 twice;
   C: a 16 x 8 x 16 path trace first (the SAME code object as K2, a few microseconds of work), then K2 twice.
If C's first K2 is warm and B's is not, the cost is loading the path tracer's code object (1.3 MB, ~150 kernels): splitting it would help;
if B's is warm too, it is the device's own warm-up (queues, clocks), which nothing in this library changes.
    GPU box:  python tools/cold_start_probe.py > gpurun_out/r05_cold_start_probe.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time
sys.path.insert(0, %r)
import __graft_entry__ as entry
B = entry.load_package().bindings
import numpy as np
mode = sys.argv[1]
t0 = time.perf_counter()
ctx = B.Context(0)
t_ctx = time.perf_counter() - t0
out = []
def timed(label, f):
    t = time.perf_counter(); f(); out.append((label, (time.perf_counter() - t) * 1e3))
buf = B.HostBuffer((600, 900, 4))
small = np.zeros((8, 16, 4), np.float32)
if mode == "B":
    timed("mandelbrot 64x64 first", lambda: ctx.mandelbrot(B.mandelbrot_params(64, 64, max_iter=32)))
if mode == "C":
    timed("pathtrace 16x8x16 first", lambda: ctx.pathtrace(B.pathtrace_params(16, 8, 16, math_mode=B.PT_MATH_FAST), out=small))
p = B.pathtrace_params(900, 600, 500, math_mode=B.PT_MATH_FAST)
for k in range(3):
    timed("K2 call %%d (kernel %%.2f ms)" %% (k + 1, 0.0), lambda: ctx.pathtrace(p, out=buf.array))
    out[-1] = (out[-1][0].replace("0.00", "%%.2f" %% ctx.last_timing()[0]), out[-1][1])
print("context %%.1f ms; " %% (t_ctx * 1e3) + "; ".join("%%s: %%.2f ms" %% x for x in out))
""" % ROOT
for mode in ("A", "B", "C", "A"):
    r = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True)
    print(f"{mode}: {r.stdout.strip() or r.stderr[-300:]}", flush=True)
