"""The sample-pool path tracer kernel (csrc/pathtrace_pool.h; MC_PT_MATH_FAST, closed-box scenes, whole sample ranges, tiles that
keep the whole image's wave tiles).  Every sample follows the round-synchronous closed-box kernel's arithmetic except where the pool
kernel uses a cheaper equivalent form (|c - x|^2 - r^2 formed once per bounce, shadow rays decided by comparing squares, a wall
bounce as a signed permutation), and a pixel's fp32 contributions are added in another order.  So: (1) against that kernel the storage
buffer agrees to the last few ulps except where a rounding forks a path (a few per cent of the pixels at these sample counts, each by at
most one sample's weight), (2) against the oracle it obeys
the fast-math tolerance, (3) it is deterministic and tiling-invariant bit for bit, (4) everything outside its domain runs the
round-synchronous kernels exactly as before.
The STRICT pool kernel keeps a per-path accrad and adds accrad / spp in sample order through a result ring in LDS: it must be
bit-identical to the oracle — and to the round-synchronous strict kernel — for every size, depth limit, sample count and tile."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def fast(ctx, B, W, H, spp, flags=0, **kw):
    return ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=flags, **kw))


@pytest.mark.parametrize("W,H,spp,depth", [(8, 8, 16, 12), (33, 9, 37, 12), (3, 5, 1, 12), (1, 1, 500, 12), (64, 48, 100, 12),
                                           (40, 24, 70, 7), (40, 24, 33, 2), (40, 24, 20, 1), (20, 12, 9, 15), (300, 200, 64, 12)])
def test_pool_kernel_agrees_with_the_round_synchronous_kernel(ctx, B, W, H, spp, depth):
    pool = fast(ctx, B, W, H, spp, max_depth=depth)[..., :3].astype(np.float64)
    rounds = fast(ctx, B, W, H, spp, flags=B.PT_NO_POOL_KERNEL, max_depth=depth)[..., :3].astype(np.float64)
    d = np.abs(pool - rounds)
    assert np.isfinite(pool).all()
    # a reassociated sum of spp x ~10 fp32 terms: a few 1e-5 of an 8-bit unit; a forked sample moves ONE pixel by up to 255 / spp
    # (a near-tie decided the other way by a differently associated discriminant: under 3 % of the components, no drift of the mean)
    assert d.mean() <= 5e-3, d.mean()
    assert (d > 1e-2).mean() <= 3e-2, (d > 1e-2).mean()
    assert abs((pool - rounds).mean()) <= 2e-3, (pool - rounds).mean()


def test_pool_kernel_is_deterministic_and_tiling_invariant(ctx, B):
    """The N-GPU == 1-GPU contract in fast math: a wave owns the same 2 x 2 pixels whatever the tile, its schedule depends on
    nothing else, so tiles equal the whole image's rows bit for bit — contiguous halves, and the interleaved 8-row blocks of
    the multi-GPU split (ranks 0..3 of 4)."""
    W, H, spp = 70, 48, 53
    whole = fast(ctx, B, W, H, spp)
    assert np.array_equal(bits(whole), bits(fast(ctx, B, W, H, spp)))
    for (r0, r1) in [(0, 24), (24, 48), (10, 30), (46, 48)]:
        assert np.array_equal(bits(fast(ctx, B, W, H, spp, row_begin=r0, row_end=r1)), bits(whole[r0:r1])), (r0, r1)
    blk, n = B.lib().mc_row_block(), 4
    for rank in range(n):
        tile = fast(ctx, B, W, H, spp, row_begin=rank * blk, row_end=H, row_block=blk, row_stride=n * blk)
        rows = np.array([r for r in range(H) if (r // blk) % n == rank])
        assert np.array_equal(bits(tile), bits(whole[rows])), rank


def test_outside_its_domain_the_round_synchronous_kernels_run(ctx, B):
    """Partial sample ranges (the accumulator continues in the buffer), forced widths, tiles that cut a wave tile, scenes that are
    not a closed box: the default flags and MC_PT_NO_POOL_KERNEL must give the same bits, i.e. the same kernel ran."""
    W, H, spp = 40, 24, 19
    cases = [dict(row_begin=5, row_end=17), dict(row_begin=0, row_end=23)]
    for kw in cases:
        assert np.array_equal(bits(fast(ctx, B, W, H, spp, **kw)), bits(fast(ctx, B, W, H, spp, flags=B.PT_NO_POOL_KERNEL, **kw))), kw
    for s in (1, 4, 16):
        a = fast(ctx, B, W, H, spp, flags=B.pt_force_s(s))
        assert np.array_equal(bits(a), bits(fast(ctx, B, W, H, spp, flags=B.pt_force_s(s) | B.PT_NO_POOL_KERNEL))), s
    # progressive ranges of the round-synchronous kernels still compose bit-exactly in fast math
    F = B.PT_NO_POOL_KERNEL
    part = fast(ctx, B, W, H, spp, flags=F, sample_begin=0, sample_end=7)
    part = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=F, sample_begin=7, sample_end=spp), acc=part)
    assert np.array_equal(bits(part), bits(fast(ctx, B, W, H, spp, flags=F)))


def test_fast_progressive_ranges_run_the_pool_kernel(ctx, B, O):
    """SURVEY §8(f)3 in fast math (round 4): a sample range is rendered by the pool kernel and its share added to the stored
    accumulator.  The split render agrees with the one-launch render to fp32 reassociation (same samples, same arithmetic per sample:
    no forked paths), is deterministic, composes identically on every tiling, and the one-launch image is what it was."""
    W, H, spp = 64, 40, 100
    assert B.pathtrace_select_kernel(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, sample_begin=20, sample_end=60)).kernel == B.PT_KERNEL_POOL
    whole = fast(ctx, B, W, H, spp)

    def split(cuts, **kw):
        acc = None
        for s0, s1 in zip(cuts[:-1], cuts[1:]):
            acc = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, sample_begin=s0, sample_end=s1, **kw), acc=acc)
        return acc
    for cuts in ([0, 20, 40, 60, 80, 100], [0, 7, 100], [0, 99, 100]):
        parts = split(cuts)
        assert np.array_equal(bits(parts), bits(split(cuts))), cuts                        # deterministic
        d = np.abs(parts[..., :3].astype(np.float64) - whole[..., :3].astype(np.float64))
        assert d.max() <= 2e-3, (cuts, d.max())                                            # 8-bit units: reassociation only
    # the same split on two row tiles = the same split on the whole image, bit for bit
    cuts = [0, 33, 100]
    top, bottom = split(cuts, row_begin=0, row_end=16), split(cuts, row_begin=16, row_end=H)
    assert np.array_equal(bits(np.concatenate([top, bottom])), bits(split(cuts)))
    # an unfinished range leaves the raw accumulator (no tonemap yet): the oracle's partial sum within the fast tolerance
    half = fast(ctx, B, W, H, spp, sample_begin=0, sample_end=50)[..., :3].astype(np.float64)
    ref = O.pathtrace(W, H, spp, math_mode=O.MATH_LIBM, sample_begin=0, sample_end=50)[..., :3].astype(np.float64)
    assert np.abs(half - ref).mean() < 2e-3 * max(1.0, ref.mean())


def test_pool_kernel_within_the_fast_tolerance_of_the_oracle(ctx, B, O):
    """96 x 64 at 256 spp against the oracle with libm: the small-size bound the other fast kernels are held to
    (tests/test_gpu_parity.py): RMSE <= 0.4, 99.9-percentile L2 <= 4, no bias."""
    W, H, spp = 96, 64, 256
    out = fast(ctx, B, W, H, spp)[..., :3].astype(np.float64)
    ref = O.pathtrace(W, H, spp, math_mode=O.MATH_LIBM)[..., :3].astype(np.float64)
    d = out - ref
    rmse = float(np.sqrt((d ** 2).mean()))
    p999 = float(np.percentile(np.sqrt((d ** 2).sum(-1)), 99.9))
    print(f"pool 96x64x256 vs oracle(libm): rmse {rmse:.4f}  p99.9 L2 {p999:.3f}  mean diff {d.mean():+.5f}")
    assert rmse <= 0.4 and p999 <= 4.0 and abs(d.mean()) < 0.05


def test_pool_kernel_k3_band_at_4096_spp(ctx, B, O):
    """K3 (3840x2560 at 4096 spp) on one 8-row block in fast math — the kernel `bench.py --config K3` times: RNG keys up to
    49 151, 256 batches per pixel, rows far from the tile origin — within the fast tolerance of the oracle with libm
    (1.26e8 samples: about 15 s of the oracle)."""
    W, H, spp = 3840, 2560, 4096
    blk = B.lib().mc_row_block()
    r0 = 161 * blk
    ref = O.pathtrace(W, H, spp, math_mode=O.MATH_LIBM, row_begin=r0, row_end=r0 + blk)[..., :3].astype(np.float64)
    band = fast(ctx, B, W, H, spp, row_begin=r0, row_end=r0 + blk)[..., :3].astype(np.float64)
    d = band - ref
    rmse = float(np.sqrt((d ** 2).mean()))
    p999 = float(np.percentile(np.sqrt((d ** 2).sum(-1)), 99.9))
    print(f"pool K3 band vs oracle(libm): rmse {rmse:.4f}  p99.9 L2 {p999:.3f}  mean diff {d.mean():+.5f}")
    assert rmse <= 0.5 and p999 <= 4.0 and abs(d.mean()) < 0.02


@pytest.mark.parametrize("W,H,spp,depth", [(8, 8, 16, 12), (33, 9, 37, 12), (3, 5, 1, 12), (1, 1, 500, 12), (64, 48, 100, 12),
                                           (40, 24, 70, 7), (40, 24, 33, 2), (40, 24, 20, 1), (20, 12, 9, 15), (16, 16, 600, 12)])
def test_strict_pool_kernel_is_bit_identical_to_the_oracle(ctx, B, O, W, H, spp, depth):
    """Sample counts around the batch (16) and result-ring (64) sizes, ragged last batches, one-sample pools, depth limits that
    end every path at once — the ordered sum through the result ring must be the oracle's, bit for bit."""
    ref = O.pathtrace(W, H, spp, math_mode=O.MATH_MC, max_depth=depth)
    pool = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_STRICT, max_depth=depth))
    assert np.array_equal(bits(pool), bits(ref))
    rounds = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_STRICT, max_depth=depth, flags=B.PT_NO_POOL_KERNEL))
    assert np.array_equal(bits(rounds), bits(ref))


def test_strict_pool_kernel_tiles_of_any_alignment(ctx, B, O):
    """Strict tiles need no alignment (the sum is in sample order whatever the wave holds): odd row ranges, interleaved blocks."""
    W, H, spp = 40, 24, 19
    whole = O.pathtrace(W, H, spp, math_mode=O.MATH_MC)
    for (r0, r1) in [(5, 17), (0, 23), (7, 8)]:
        t = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_STRICT, row_begin=r0, row_end=r1))
        assert np.array_equal(bits(t), bits(whole[r0:r1])), (r0, r1)
    blk, n = B.lib().mc_row_block(), 2
    for rank in range(n):
        t = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_STRICT, row_begin=rank * blk, row_end=H, row_block=blk,
                                             row_stride=n * blk))
        rows = np.array([r for r in range(H) if (r // blk) % n == rank])
        assert np.array_equal(bits(t), bits(whole[rows])), rank


def test_strict_pool_kernel_continues_an_accumulator(ctx, B, O):
    """The samps.x protocol (pathTracer.comp:451-453) through the strict pool kernel: ranges that start and end inside batches, a
    one-sample range, the last range applying :453 — every stage bit-identical to the oracle's accumulator at that stage and the
    final image to a one-launch render; also on an interleaved tile (the multi-GPU checkpoint case)."""
    W, H, spp = 40, 24, 53
    cuts = [0, 7, 8, 30, 52, 53]
    acc = ref = None
    for b, e in zip(cuts[:-1], cuts[1:]):
        acc = ctx.pathtrace(B.pathtrace_params(W, H, spp, sample_begin=b, sample_end=e), acc=acc)
        ref = O.pathtrace(W, H, spp, math_mode=O.MATH_MC, sample_begin=b, sample_end=e, acc=ref)
        assert np.array_equal(bits(acc), bits(ref)), (b, e)
    assert np.array_equal(bits(acc), bits(ctx.pathtrace(B.pathtrace_params(W, H, spp))))
    # the same through the round-synchronous kernels (what ran before the pool kernel took sample ranges)
    acc2 = None
    for b, e in zip(cuts[:-1], cuts[1:]):
        acc2 = ctx.pathtrace(B.pathtrace_params(W, H, spp, sample_begin=b, sample_end=e, flags=B.PT_NO_POOL_KERNEL), acc=acc2)
    assert np.array_equal(bits(acc2), bits(acc))
    blk = B.lib().mc_row_block()
    whole = O.pathtrace(W, H, spp, math_mode=O.MATH_MC)
    kw = dict(row_begin=blk, row_end=H, row_block=blk, row_stride=2 * blk)
    t = ctx.pathtrace(B.pathtrace_params(W, H, spp, sample_begin=0, sample_end=20, **kw))
    t = ctx.pathtrace(B.pathtrace_params(W, H, spp, sample_begin=20, sample_end=spp, **kw), acc=t)
    rows = np.array([r for r in range(H) if (r // blk) % 2 == 1])
    assert np.array_equal(bits(t), bits(whole[rows]))


def _scene(O):
    return O.DEFAULT_PLANES.copy().reshape(6, 12), O.DEFAULT_SPHERES.copy().reshape(3, 12)


def test_fast_pool_kernel_with_overlapping_spheres(ctx, B, O):
    """The fast pool kernel decides shadow rays without square roots by ordering the spheres a ray meets by the projections of their
    centres — the order of their hits only for DISJOINT spheres (pathtrace_kernel.h, shadow_visible_disjoint).  With two spheres pushed
    into each other the host launches the pool kernel's other instantiation (round 4; before: the round-synchronous kernel), which
    takes the roots (shadow_reaches_sphere): inside the fast tolerance of the oracle with libm, close to the round-synchronous
    closed-box kernel (different instruction sequences, forked samples only); the strict pool kernel has no such premise."""
    planes, spheres = _scene(O)
    spheres[1, 0:3] = spheres[0, 0:3] + np.float32([0.9, 0.0, 0.3])   # the glass sphere cuts into the mirror sphere
    assert B.pathtrace_scene_class(planes, spheres) & B.PT_SCENE_SPHERES_DISJOINT == 0
    W, H, spp = 96, 64, 256
    q = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST)
    assert B.pathtrace_select_kernel(q, planes, spheres).kernel == B.PT_KERNEL_POOL
    fast = ctx.pathtrace(q, planes=planes, spheres=spheres)
    rounds = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=B.PT_NO_POOL_KERNEL), planes=planes, spheres=spheres)
    libm = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_LIBM)
    for name, img in (("pool", fast), ("rounds", rounds)):
        d = img[..., :3].astype(np.float64) - libm[..., :3].astype(np.float64)
        rmse, p999 = float(np.sqrt((d ** 2).mean())), float(np.percentile(np.sqrt((d ** 2).sum(-1)), 99.9))
        print(f"overlapping spheres, {name}: rmse {rmse:.4f}  p99.9 L2 {p999:.3f}  mean {d.mean():+.5f}")
        assert np.isfinite(img).all() and rmse <= 0.4 and p999 <= 4.0 and abs(d.mean()) < 0.05, (name, rmse, p999)
    strict = ctx.pathtrace(B.pathtrace_params(W, H, spp), planes=planes, spheres=spheres)
    assert np.array_equal(bits(strict), bits(O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)))
    # tiling invariance holds for this instantiation too
    top = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, row_begin=0, row_end=32), planes=planes, spheres=spheres)
    assert np.array_equal(bits(top), bits(fast[:32]))


@pytest.mark.parametrize("case", ["light_is_sphere_0", "two_lights", "diffuse_sphere_and_mirror_wall"])
def test_pool_kernel_scene_variants_within_tolerance(ctx, B, O, case):
    """What the default scene does not exercise in the pool kernels: a light that is not the last sphere and two lights (the unrolled
    light loop, shadow rays whose target is sphere 0 / 1), a diffuse non-emitting sphere (the general cosine bounce beside the wall
    form, shadow rays that start ON an occluder) and a mirror wall (a specular bounce off a plane).  Fast within the small-size
    tolerance of the oracle with libm, strict bit-identical."""
    planes, spheres = _scene(O)
    if case == "light_is_sphere_0":
        spheres[[0, 2]] = spheres[[2, 0]]
    elif case == "two_lights":
        spheres[1, 4:7] = np.float32([40.0, 30.0, 20.0]); spheres[1, 8:11] = 0.0; spheres[1, 11] = 1.0; spheres[1, 3] = np.float32(0.3)
    else:
        spheres[0, 8:11] = np.float32([0.7, 0.5, 0.3]); spheres[0, 11] = 1.0
        planes[4, 11] = 2.0; planes[4, 8:11] = np.float32(0.9)
    W, H, spp = 96, 64, 256
    # still a closed-box slab scene; the mirror wall is more specular surface than the reference scene's, so a fast request for that case is
    # rendered by the careful tier (round 6) — the fast tier's bounce off a specular plane stays covered through the measurement switch
    cls = B.pathtrace_scene_class(planes, spheres)
    assert cls & ~B.PT_SCENE_SPECULAR == B.pathtrace_scene_class(*_scene(O)) and bool(cls & B.PT_SCENE_SPECULAR) == (case == "diffuse_sphere_and_mirror_wall")
    strict = ctx.pathtrace(B.pathtrace_params(W, H, spp), planes=planes, spheres=spheres)
    assert np.array_equal(bits(strict), bits(O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)))
    ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_LIBM)[..., :3].astype(np.float64)
    for flags in (0, B.PT_NO_FAST_GUARD) if cls & B.PT_SCENE_SPECULAR else (0,):
        q = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=flags)
        ki = B.pathtrace_select_kernel(q, planes, spheres)
        assert ki.kernel == B.PT_KERNEL_POOL
        assert ki.math_mode == (B.PT_MATH_FAST_CAREFUL if cls & B.PT_SCENE_SPECULAR and not flags else B.PT_MATH_FAST)
        d = ctx.pathtrace(q, planes=planes, spheres=spheres)[..., :3].astype(np.float64) - ref
        rmse = float(np.sqrt((d ** 2).mean()))
        p999 = float(np.percentile(np.sqrt((d ** 2).sum(-1)), 99.9))
        print(f"{case} (tier {ki.math_mode}): pool vs oracle(libm): rmse {rmse:.4f}  p99.9 L2 {p999:.3f}  mean diff {d.mean():+.5f}")
        assert rmse <= 0.5 and p999 <= 4.0 and abs(d.mean()) < 0.05


def test_a_tripped_scheduler_bound_is_reported_not_stored_silently(B):
    """the pool kernels bound their scheduling loop; if a defect ever exhausted the bound the image would be incomplete.
    A diagnostic build whose bound is FIVE iterations must make the blocking entry point fail (status word -> MC_ERR_HIP) instead of
    returning a partial image with MC_OK; the shipped library renders the same request fine."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    pkg = os.path.join(ROOT, "vulkan-compute-tests_amd")
    subprocess.check_call(["make", "-s", "-C", pkg, "exp", "EXP_NAME=bound", "EXP_FLAGS=-DMC_PT_POOL_TEST_BOUND=5"])
    child = ("import sys; sys.path.insert(0, %r)\n"
             "import __graft_entry__ as e\n"
             "B = e.load_package().bindings\n"
             "with B.Context(0) as ctx:\n"
             "    try:\n"
             "        ctx.pathtrace(B.pathtrace_params(16, 8, 64, math_mode=B.PT_MATH_FAST))\n"
             "        print('NO ERROR')\n"
             "    except B.McError as err:\n"
             "        print('STATUS', err.status, err)\n"
             "    ctx.pathtrace(B.pathtrace_params(16, 8, 64, math_mode=B.PT_MATH_FAST, flags=B.PT_NO_POOL_KERNEL))\n"
             "    print('ROUNDS OK')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True,
                       env=dict(os.environ, MC_LIB_PATH=os.path.join(pkg, "lib", "libmc_compute_exp_bound.so")))
    assert "STATUS 3" in r.stdout and "loop bound" in r.stdout and "ROUNDS OK" in r.stdout, r.stdout + r.stderr[-2000:]
