"""Parity of the lane-regrouping path tracer kernel (csrc/pathtrace_regroup.h).  The kernel is NOT part of the shipped
library — it measured slower than the round-synchronous kernels — and lives in the diagnostic library
lib/libmc_compute_regroup.so (`make regroup`; built on demand by this module, NOT by __graft_entry__.build()).  Its scheduling moves paths between
lanes and waves and finishes samples out of order, so bit-identity with the oracle is the meaningful test: strict math,
every size / depth / range / tile case below must match bit for bit.  Each case runs in a child process that loads the
diagnostic library through MC_LIB_PATH (the parent's library stays the shipped one)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

# A documented negative result is not part of the driver's `-m gpu` run: marker `diag`, opt in with MC_RUN_DIAG=1 on a GPU box.
pytestmark = [pytest.mark.diag,
              pytest.mark.skipif(not os.environ.get("MC_RUN_DIAG"), reason="diagnostic kernel: set MC_RUN_DIAG=1 on a GPU box")]

LIB = os.path.join(ROOT, "vulkan-compute-tests_amd", "lib", "libmc_compute_regroup.so")

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as entry
B = entry.load_package().bindings
O = entry.load_oracle()
def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
F = B.PT_KERNEL_REGROUP
with B.Context(0) as ctx:
    # sizes (tile overhang, single pixels), sample counts around the window / queue sizes, depth limits
    for (W, H, spp, depth) in [(4, 4, 1, 12), (1, 1, 3, 12), (3, 2, 17, 12), (33, 9, 37, 12), (48, 32, 130, 12), (24, 16, 70, 7),
                               (24, 16, 33, 2), (24, 16, 20, 1), (16, 16, 600, 12), (8, 4, 2100, 12), (20, 12, 9, 15)]:
        out = ctx.pathtrace(B.pathtrace_params(W, H, spp, max_depth=depth, flags=F))
        ref = O.pathtrace(W, H, spp, math_mode=O.MATH_MC, max_depth=depth)
        assert np.array_equal(bits(out), bits(ref)), (W, H, spp, depth)
    # progressive ranges compose, tiles (contiguous and interleaved) equal the whole image
    W, H, spp = 40, 24, 19
    whole = O.pathtrace(W, H, spp, math_mode=O.MATH_MC)
    part = ctx.pathtrace(B.pathtrace_params(W, H, spp, sample_begin=0, sample_end=7, flags=F))
    part = ctx.pathtrace(B.pathtrace_params(W, H, spp, sample_begin=7, sample_end=19, flags=F), acc=part)
    assert np.array_equal(bits(part), bits(whole))
    t = ctx.pathtrace(B.pathtrace_params(W, H, spp, row_begin=5, row_end=17, flags=F))
    assert np.array_equal(bits(t), bits(whole[5:17]))
    blk, n = 8, 2
    for rank in range(n):
        p = B.pathtrace_params(W, H, spp, row_begin=rank * blk, row_end=H, row_block=blk, row_stride=n * blk, flags=F)
        rows = [r for r in range(H) if (r // blk) %% n == rank]
        assert np.array_equal(bits(ctx.pathtrace(p)), bits(whole[rows])), rank
    # scenes: specular walls (long specular chains), two lights, light outside the shadow-ray shortcut
    planes, spheres = O.DEFAULT_PLANES.copy().reshape(6, 12), O.DEFAULT_SPHERES.copy().reshape(3, 12)
    planes[0, 11] = 2.0; planes[4, 11] = 3.0
    spheres[0, 4:7] = (30.0, 20.0, 10.0); spheres[0, 1] = -1.0; spheres[0, 8:11] = 0.0; spheres[0, 11] = 1.0
    out = ctx.pathtrace(B.pathtrace_params(40, 28, 24, flags=F), planes=planes, spheres=spheres)
    ref = O.pathtrace(40, 28, 24, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
    assert np.array_equal(bits(out), bits(ref))
    spheres = O.DEFAULT_SPHERES.copy().reshape(3, 12); spheres[2, 1] = 1.79
    out = ctx.pathtrace(B.pathtrace_params(32, 20, 12, flags=F), spheres=spheres, planes=O.DEFAULT_PLANES)
    assert np.array_equal(bits(out), bits(O.pathtrace(32, 20, 12, spheres=spheres, planes=O.DEFAULT_PLANES, math_mode=O.MATH_MC)))
    # outside its limits the request falls back to the round-synchronous kernels (material code 4, depth 70)
    spheres = O.DEFAULT_SPHERES.copy().reshape(3, 12); spheres[0, 11] = 4.0
    out = ctx.pathtrace(B.pathtrace_params(16, 12, 5, flags=F), spheres=spheres, planes=O.DEFAULT_PLANES)
    assert np.array_equal(bits(out), bits(O.pathtrace(16, 12, 5, spheres=spheres, planes=O.DEFAULT_PLANES, math_mode=O.MATH_MC)))
    out = ctx.pathtrace(B.pathtrace_params(16, 12, 3, max_depth=70, flags=F))
    assert np.array_equal(bits(out), bits(O.pathtrace(16, 12, 3, math_mode=O.MATH_MC, max_depth=70)))
    # fast math: same toleranced agreement with the oracle as the shipped kernel at a small size
    fast = ctx.pathtrace(B.pathtrace_params(96, 64, 64, math_mode=B.PT_MATH_FAST, flags=F))[..., :3].astype(np.float64)
    ref = O.pathtrace(96, 64, 64, math_mode=O.MATH_LIBM)[..., :3].astype(np.float64)
    assert np.sqrt(((fast - ref) ** 2).mean()) <= 0.75
print("REGROUP PARITY OK")
""" % ROOT


def test_regroup_kernel_bit_exact_in_the_diagnostic_library(B):
    subprocess.check_call(["make", "-s", "-j", "8", "-C", os.path.join(ROOT, "vulkan-compute-tests_amd"), "regroup"])
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MC_LIB_PATH=LIB), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "REGROUP PARITY OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
