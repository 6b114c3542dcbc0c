"""GPU tests at BASELINE.json's full sizes (K3, K4) through size-independent properties: rows sampled against the
oracle, the frozen pixel-iteration checksum, and invariance under the 8-way interleaved row tiling the 8-GPU run uses."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K4_CENTRE = (-0.7436438870371587, 0.13182590420531198)
K4_SCALE = (1e-8, 1e-8 * 2.0 / 3.0)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_k4_deep_zoom_full_size(ctx, B, O):
    """K4: 7680x5120, M = 50 000, two-float.  Frozen in BASELINE.md: sum of executed loop bodies 41 176 259 776,
    max n = 9068, no interior pixel."""
    W, H, M = 7680, 5120, 50000
    p = B.mandelbrot_params(W, H, max_iter=M, precision=B.PRECISION_DS, centre=K4_CENTRE, scale=K4_SCALE)
    _, it = ctx.mandelbrot(p, want_rgba=False)
    bodies = np.where(it < M, it.astype(np.int64) + 1, M)
    assert int(bodies.sum()) == 41176259776
    assert int(it.max()) == 9068 and int((it == M).sum()) == 0
    view = O.make_view(K4_CENTRE[0], K4_CENTRE[1], K4_SCALE[0], K4_SCALE[1])
    for r in (0, 2559, 2560, 5119):                      # rows against the oracle, bit-exact
        assert np.array_equal(it[r], O.mandelbrot_iters(W, H, M, view=view, precision=1, row_begin=r, row_end=r + 1)[0]), r
    # the tile rank 5 of 8 would render (interleaved 8-row blocks) equals the same rows of the whole image
    blk, n, rank = 8, 8, 5
    q = B.mandelbrot_params(W, H, max_iter=M, precision=B.PRECISION_DS, centre=K4_CENTRE, scale=K4_SCALE,
                            row_begin=rank * blk, row_end=H, row_block=blk, row_stride=n * blk)
    _, tile = ctx.mandelbrot(q, want_rgba=False)
    rows = np.array([r for r in range(H) if (r // blk) % n == rank])
    assert tile.shape[0] == len(rows) == H // n
    assert np.array_equal(tile, it[rows])


def test_k3_image_tiling_invariance(ctx, B, O):
    """K3 geometry (3840x2560, the 8-GPU path-trace config) at 4 spp: the union of the 8 ranks' interleaved tiles is
    bit-identical to the whole-image render, and sampled rows match the oracle (strict math)."""
    W, H, spp = 3840, 2560, 4
    whole = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_STRICT))
    blk, n = 8, 8
    for rank in (0, 3, 7):
        p = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_STRICT, row_begin=rank * blk, row_end=H, row_block=blk,
                               row_stride=n * blk)
        rows = np.array([r for r in range(H) if (r // blk) % n == rank])
        assert np.array_equal(bits(ctx.pathtrace(p)), bits(whole[rows])), rank
    for r in (0, 1279, 2559):
        ref = O.pathtrace(W, H, spp, math_mode=O.MATH_MC, row_begin=r, row_end=r + 1)
        assert np.array_equal(bits(whole[r:r + 1]), bits(ref)), r
