"""The fast-math bound beyond the one scene and size it was stated on (the bound was the only guard of the fast
kernels' identities, tested on one scene at one size).  Six random slab scenes — jittered walls, some specular, spheres moved /
resized / re-materialised; closed boxes (the sample-pool and closed-box kernels) and open ones (the general slab kernel) — and the
two image sizes next to K2, all held to the SAME numbers as tests/test_gpu_fullsize.py: RMSE <= 0.5 and 99.9-percentile per-pixel
RGB L2 <= 4 against the oracle evaluated with libm, no bias; the strict kernel bit-identical on every scene."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
from fast_tolerance_scenes import scene, stats  # noqa: E402  (the scene generator of the measurement tool)

pytestmark = pytest.mark.gpu


def test_fast_math_bound_on_random_slab_scenes(ctx, B, O):
    rng = np.random.default_rng(3)
    W, H, spp = 300, 200, 256
    classes = set()
    for k in range(6):
        planes, spheres = scene(rng, O)
        classes.add(B.pathtrace_scene_class(planes, spheres))
        fast = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST), planes=planes, spheres=spheres)
        strict = ctx.pathtrace(B.pathtrace_params(W, H, spp), planes=planes, spheres=spheres)
        libm = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_LIBM)
        mc = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
        assert np.array_equal(strict.view(np.uint32), mc.view(np.uint32)), k
        rmse, p999, mean = stats(fast, libm)
        print(f"scene {k}: fast vs oracle(libm): rmse {rmse:.4f}  p99.9 L2 {p999:.3f}  mean diff {mean:+.5f}")
        assert rmse <= 0.5 and p999 <= 4.0 and abs(mean) < 0.02, k
    assert len(classes) >= 2          # both the closed-box kernels and the general slab kernel were exercised


@pytest.mark.parametrize("W,H", [(906, 604), (894, 596)])
def test_fast_math_bound_at_the_sizes_next_to_k2(ctx, B, O, W, H):
    spp = 500
    fast = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST))
    libm = O.pathtrace(W, H, spp, math_mode=O.MATH_LIBM)
    rmse, p999, mean = stats(fast, libm)
    print(f"{W}x{H}x{spp} fast vs oracle(libm): rmse {rmse:.4f}  p99.9 L2 {p999:.3f}  mean diff {mean:+.5f}  (headroom {100 * (1 - p999 / 4.0):.0f} %)")
    assert rmse <= 0.5 and p999 <= 4.0 and abs(mean) < 0.02
