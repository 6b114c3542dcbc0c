"""Where does a K2 launch of the sample-pool kernel lose time against K3?  (K3 does 13 % more samples per clock.)
Times the fast path trace over (a) spp at 900x600 — per-wave time = F + B * batches, F = fill / drain per tile — and
(b) image height at 500 spp — the same waves, more launch rounds: what the grid tail costs.  Prints ns per sample and the clock.
  python tools/pool_scaling_probe.py [lib.so ...]      (MC_TIME_MATH=strict for the strict kernel)"""
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
mode = B.PT_MATH_STRICT if os.environ.get("MC_TIME_MATH") == "strict" else B.PT_MATH_FAST
flags = int(os.environ.get("MC_PT_FLAGS", "0"), 0)
def run(W, H, spp, reps):
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    p = B.pathtrace_params(W, H, spp, math_mode=mode, flags=flags)
    for _ in range(2): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ctx.pathtrace_device(p, buf.data_ptr(), stream=s)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    clk = ctx.measure_clock() if hasattr(ctx, "measure_clock") else 0.0
    print("RES %%5d x %%5d spp %%5d : %%9.3f ms  %%7.4f ns/sample  clock %%s" %% (W, H, spp, best, best * 1e6 / (W * H * spp), clk), flush=True)
for spp in (16, 32, 64, 125, 250, 500, 1000, 2000, 4000):
    run(900, 600, spp, max(1, 4000 // spp))
for H in (152, 304, 600, 1200, 2400, 4800):
    run(900, H, 500, max(1, 2400 // H))
run(3840, 2560, 500, 1)
run(3840, 2560, 4096, 1)
""" % ROOT
libs = sys.argv[1:] or [os.path.join(ROOT, "vulkan-compute-tests_amd", "lib", "libmc_compute.so")]
for lib in libs:
    print("==", os.path.basename(lib), flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MC_LIB_PATH=os.path.abspath(lib)), capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RES")]
    print("\n".join(lines) if lines else "FAILED " + r.stderr[-800:], flush=True)
