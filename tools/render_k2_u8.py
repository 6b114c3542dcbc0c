"""Renders BASELINE config K2 (900x600, 500 spp) on the GPU in both math modes and stores the RGBA8 images
under gpurun_out/ so they can be compared per pixel with the reference's imageForReadme.png on a machine
that has the reference checkout (tools/compare_with_readme_image.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
B = entry.load_package().bindings
os.makedirs("gpurun_out", exist_ok=True)
with B.Context(0) as ctx:
    for name, mode in (("strict", B.PT_MATH_STRICT), ("fast", B.PT_MATH_FAST)):
        buf = ctx.pathtrace(B.pathtrace_params(900, 600, 500, math_mode=mode))
        np.save(f"gpurun_out/k2_{name}_u8.npy", ctx.convert_rgba8(buf, 1.0, rotate180=True))
print("ok")
