"""Times the generic path tracer kernel (scenes of 6 planes + N spheres, 900x600, 100 spp; records in LDS / in memory; strict and fast)
with each library given (MC_LIB_PATH, one child process per library).   python tools/time_generic.py lib1.so lib2.so ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, os, torch, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import __graft_entry__ as entry
from bench_widened import room
B = entry.load_package().bindings
ctx = B.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
W, H, spp = 900, 600, 100
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
rng = np.random.default_rng(5)
for n, lights in ((64, 3), (512, 5), (1500, 8)):
    planes, spheres = room(rng, n, lights)
    for mode, mname in ((B.PT_MATH_STRICT, "strict"), (B.PT_MATH_FAST, "fast")):
        for where, flag in (("LDS", B.PT_SCENE_IN_LDS), ("mem", B.PT_SCENE_IN_MEMORY)):
            p = B.pathtrace_params(W, H, spp, math_mode=mode, flags=flag)
            ctx.pathtrace_device(p, buf.data_ptr(), planes=planes, spheres=spheres, stream=s); torch.cuda.synchronize()
            reps = 1 if n > 1000 else 3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): ctx.pathtrace_device(p, buf.data_ptr(), planes=planes, spheres=spheres, stream=s)
            e1.record(); torch.cuda.synchronize()
            print("MS %%5d spheres %%-6s %%-3s %%10.2f  mean %%.5f" %% (n, mname, where, e0.elapsed_time(e1) / reps, float(buf[..., :3].double().mean())), flush=True)
""" % (ROOT, ROOT)
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MC_LIB_PATH=os.path.abspath(lib)), capture_output=True, text=True)
    print(os.path.basename(lib))
    lines = [l for l in r.stdout.splitlines() if l.startswith("MS")]
    print("\n".join(lines) if lines else "FAILED " + r.stderr[-800:], flush=True)
