"""GPU parity tests of the two-path-slots-per-lane path tracer scheduler (csrc/pathtrace_pq.h, MC_PT_KERNEL_PQ):
samples finish out of order and are folded in through a per-pixel reorder ring, so the result must still be
bit-identical to the round-synchronous kernel and (strict math) to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("W,H,spp", [(48, 32, 8), (37, 23, 21), (8, 4, 1), (5, 3, 70), (96, 64, 40)])
def test_pq_strict_bit_exact(ctx, B, O, W, H, spp):
    out = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_STRICT, flags=B.PT_KERNEL_PQ))
    ref = O.pathtrace(W, H, spp, math_mode=O.MATH_MC)
    assert np.array_equal(bits(out), bits(ref))


def test_pq_fast_close_to_round_synchronous_fast(ctx, B):
    """Fast math allows a*b+c contraction (pathtrace_fast.hip), and the two kernels are different instruction
    sequences: equal within the fast-math tolerance, not bit for bit (strict math above IS bit for bit)."""
    W, H, spp = 64, 40, 33
    a = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=B.PT_KERNEL_PQ))
    b = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST))
    d = (a[..., :3] - b[..., :3]).astype(np.float64)
    assert np.sqrt((d ** 2).mean()) < 1.0 and abs(d.mean()) < 0.1


def test_pq_progressive_ranges_and_tiles(ctx, B, O):
    W, H, spp = 40, 28, 19
    whole = O.pathtrace(W, H, spp, math_mode=O.MATH_MC)
    part = ctx.pathtrace(B.pathtrace_params(W, H, spp, sample_begin=0, sample_end=7, flags=B.PT_KERNEL_PQ))
    part = ctx.pathtrace(B.pathtrace_params(W, H, spp, sample_begin=7, sample_end=19, flags=B.PT_KERNEL_PQ), acc=part)
    assert np.array_equal(bits(part), bits(whole))
    t = ctx.pathtrace(B.pathtrace_params(W, H, spp, row_begin=5, row_end=17, flags=B.PT_KERNEL_PQ))
    assert np.array_equal(bits(t), bits(whole[5:17]))
    n, blk = 3, 8
    for rank in range(n):
        p = B.pathtrace_params(W, H, spp, row_begin=rank * blk, row_end=H, row_block=blk, row_stride=n * blk, flags=B.PT_KERNEL_PQ)
        rows = [r for r in range(H) if (r // blk) % n == rank]
        assert np.array_equal(bits(ctx.pathtrace(p)), bits(whole[rows]))


def test_pq_many_samples_exercise_the_reorder_window(ctx, B, O):
    """300 spp on a tiny image: thousands of items per wave, the reorder window and item flow control are busy."""
    W, H, spp = 8, 6, 300
    out = ctx.pathtrace(B.pathtrace_params(W, H, spp, flags=B.PT_KERNEL_PQ))
    ref = O.pathtrace(W, H, spp, math_mode=O.MATH_MC)
    assert np.array_equal(bits(out), bits(ref))
