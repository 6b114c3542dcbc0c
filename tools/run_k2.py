"""Runs the K2 path trace a few times with one library (for rocprofv3 passes over experiment builds):
  rocprofv3 ... -- python3 tools/run_k2.py <lib.so> [fast|strict] [launches] [flags]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MC_LIB_PATH"] = os.path.abspath(sys.argv[1])
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

B = entry.load_package().bindings
mode = B.PT_MATH_STRICT if len(sys.argv) > 2 and sys.argv[2] == "strict" else B.PT_MATH_FAST
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
flags = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0
ctx = B.Context(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
buf = torch.zeros((600, 900, 4), dtype=torch.float32, device="cuda")
p = B.pathtrace_params(900, 600, 500, math_mode=mode, flags=flags)
for _ in range(n):
    ctx.pathtrace_device(p, buf.data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize()
print("done", float(buf[..., :3].double().mean()))
