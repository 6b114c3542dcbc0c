"""GPU parity tests for general scene input: scenes of arbitrary size take the generic
kernel, whose plane/sphere records are staged from a device buffer into LDS (pathTracer.comp:116,127,403 loop
over `planes.length()` / `spheres.length()`).  Strict math must stay bit-identical to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_scene(rng, n_planes, n_spheres, n_lights):
    """A closed room (the reference's six walls, possibly repeated further out) plus random spheres."""
    base = np.array([
        -1.0, 0.0, 0.0, 2.6, 0, 0, 0, 0, .85, .25, .25, 1,
        +1.0, 0.0, 0.0, 2.6, 0, 0, 0, 0, .25, .35, .85, 1,
        0.0, +1.0, 0.0, 2.0, 0, 0, 0, 0, .75, .75, .75, 1,
        0.0, -1.0, 0.0, 2.0, 0, 0, 0, 0, .75, .75, .75, 1,
        0.0, 0.0, -1.0, 2.8, 0, 0, 0, 0, .85, .85, .25, 1,
        0.0, 0.0, +1.0, 7.9, 0, 0, 0, 0, 0.1, 0.7, 0.7, 1], np.float32).reshape(6, 12)
    planes = [base]
    while sum(len(p) for p in planes) < n_planes:          # extra, tilted planes outside the room (never the nearest hit
        extra = base.copy()                                # from inside, but every one is tested)
        extra[:, 3] += rng.uniform(0.5, 3.0, 6).astype(np.float32)
        tilt = rng.normal(0, 0.05, (6, 3)).astype(np.float32)
        n = extra[:, :3] + tilt
        extra[:, :3] = n / np.linalg.norm(n, axis=1, keepdims=True)
        planes.append(extra)
    planes = np.concatenate(planes)[:n_planes]
    spheres = np.zeros((n_spheres, 12), np.float32)
    spheres[:, 0] = rng.uniform(-2.2, 2.2, n_spheres)
    spheres[:, 1] = rng.uniform(-1.8, 1.2, n_spheres)
    spheres[:, 2] = rng.uniform(-2.4, 2.5, n_spheres)
    spheres[:, 3] = rng.uniform(0.05, 0.35, n_spheres)
    spheres[:, 8:11] = rng.uniform(0.2, 0.95, (n_spheres, 3))
    spheres[:, 11] = rng.choice([1, 1, 1, 2, 3], n_spheres)
    lights = rng.choice(n_spheres, n_lights, replace=False)
    spheres[lights, 4:7] = rng.uniform(20, 80, (n_lights, 3))
    spheres[lights, 8:11] = 0
    spheres[lights, 11] = 1
    spheres[lights, 1] = rng.uniform(1.2, 1.7, n_lights)
    spheres[lights, 3] = 0.15
    return planes.astype(np.float32), spheres


@pytest.mark.parametrize("n_planes,n_spheres,n_lights", [(6, 40, 3), (30, 17, 1), (12, 200, 5), (1, 1, 1)])
def test_large_generic_scenes_bit_exact(ctx, B, O, n_planes, n_spheres, n_lights):
    rng = np.random.default_rng(100 + n_spheres)
    planes, spheres = random_scene(rng, n_planes, n_spheres, n_lights)
    W, H, spp = 24, 16, 6
    ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
    for S in (1, 4):
        out = ctx.pathtrace(B.pathtrace_params(W, H, spp, flags=B.pt_force_s(S)), planes=planes, spheres=spheres)
        assert np.array_equal(bits(out), bits(ref)), S
    assert np.isfinite(ref).all() and ref[..., :3].max() > 1.0


def test_scene_beyond_48k_of_lds(ctx, B, O):
    """1500 objects = 72 KB of records: needs the opt-in dynamic-LDS window (gfx950 has 160 KB per CU)."""
    rng = np.random.default_rng(7)
    planes, spheres = random_scene(rng, 6, 1494, 4)
    W, H, spp = 16, 8, 2
    out = ctx.pathtrace(B.pathtrace_params(W, H, spp), planes=planes, spheres=spheres)
    ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
    assert np.array_equal(bits(out), bits(ref))


def test_scene_too_large_is_rejected(ctx, B):
    """Beyond the LDS-resident store (about 3000 objects) only the fp32 kernels run (from memory, below); a forced LDS copy, an
    extended-precision branch and more than 2^20 objects are MC_ERR_UNSUPPORTED."""
    planes = np.zeros((4000, 12), np.float32)
    planes[:, 0] = 1.0
    for flags in (B.PT_SCENE_IN_LDS, B.pt_precision(B.PT_PREC_FP64)):
        with pytest.raises(B.McError) as e:
            ctx.pathtrace(B.pathtrace_params(8, 8, 1, flags=flags), planes=planes, spheres=np.zeros((1, 12), np.float32))
        assert e.value.status == 5     # MC_ERR_UNSUPPORTED
    huge = np.zeros(((1 << 20) + 1, 12), np.float32)
    huge[:, 0] = 1.0
    with pytest.raises(B.McError) as e:
        ctx.pathtrace(B.pathtrace_params(8, 8, 1), planes=huge, spheres=np.zeros((1, 12), np.float32))
    assert e.value.status == 5


def test_scene_change_between_launches(ctx, B, O):
    """The device copy of a generic scene is cached by content: alternate two scenes and the default scene."""
    rng = np.random.default_rng(3)
    a = random_scene(rng, 6, 20, 2)
    b = random_scene(rng, 8, 33, 1)
    for planes, spheres in (a, b, a, (O.DEFAULT_PLANES, O.DEFAULT_SPHERES), b):
        out = ctx.pathtrace(B.pathtrace_params(20, 12, 3), planes=planes, spheres=spheres)
        ref = O.pathtrace(20, 12, 3, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
        assert np.array_equal(bits(out), bits(ref))


# ---------------------------------------------------------------------------------------------------------------
# shadow rays skip the slab tests when the host proves the lights lie inside the closed box (pathtrace.hip,
# lights_inside_box).  Strict math must stay bit-identical to the oracle (which never skips anything) whether the
# shortcut is taken, refused, or the scene sits close to the decision boundary.
# ---------------------------------------------------------------------------------------------------------------
def _box_scene(O, **light):
    planes = O.DEFAULT_PLANES.copy().reshape(6, 12)
    spheres = O.DEFAULT_SPHERES.copy().reshape(3, 12)
    for k, v in light.items():
        spheres[2, {"x": 0, "y": 1, "z": 2, "r": 3}[k]] = v
    return planes, spheres


@pytest.mark.parametrize("name,light", [
    ("default (shortcut taken)", {}),
    ("light 0.05 below the ceiling (taken, just above the 0.0436 margin)", {"y": 1.75, "r": 0.2}),
    ("light 0.04 below the ceiling (refused, just below the margin)", {"y": 1.76, "r": 0.2}),
    ("light 0.01 below the ceiling (refused)", {"y": 1.79, "r": 0.2}),
    ("small light 0.05 above the floor next to the back wall (taken)", {"x": 0.3, "y": -1.9, "z": -2.7, "r": 0.05}),
    ("light pokes through the ceiling (refused)", {"y": 1.95, "r": 0.2}),
    ("light touches the left wall (refused)", {"x": -2.45, "r": 0.15}),
    ("large light near the floor and back wall (taken)", {"x": 0.5, "y": -1.2, "z": -1.9, "r": 0.75}),
    ("light outside the room (refused, never reached)", {"y": 3.5, "r": 0.3}),
])
def test_shadow_ray_plane_skip_is_exact(ctx, B, O, name, light):
    planes, spheres = _box_scene(O, **light)
    W, H, spp = 48, 32, 12
    ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
    for flags in (0, B.pt_force_s(1), B.pt_force_s(16), B.PT_GENERIC_KERNEL):
        out = ctx.pathtrace(B.pathtrace_params(W, H, spp, flags=flags), planes=planes, spheres=spheres)
        assert np.array_equal(bits(out), bits(ref)), (name, flags)


def test_shadow_ray_plane_skip_with_two_lights_and_mirror_walls(ctx, B, O):
    """Two emissive spheres (both inside) and specular walls: vertices reached through mirrors still lie in the box."""
    planes, spheres = _box_scene(O)
    planes[0, 11] = 2.0            # left wall becomes a mirror
    planes[4, 11] = 2.0            # back wall too
    spheres[0, 4:7] = (30.0, 20.0, 10.0)   # the former mirror sphere now also emits ...
    spheres[0, 1] = -1.0                   # ... and is lifted off the floor, so that the shortcut is taken
    spheres[0, 8:11] = 0.0
    spheres[0, 11] = 1.0
    W, H, spp = 40, 28, 10
    ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
    for flags in (0, B.pt_force_s(4), B.pt_force_s(16)):
        out = ctx.pathtrace(B.pathtrace_params(W, H, spp, flags=flags), planes=planes, spheres=spheres)
        assert np.array_equal(bits(out), bits(ref)), flags


@pytest.mark.parametrize("which,code", [("sphere", 4.0), ("sphere", 0.0), ("sphere", -2.0), ("wall", 7.0), ("wall", 0.4)])
def test_unknown_material_codes_keep_their_ray(ctx, B, O, which, code):
    """A material code other than 1, 2, 3 matches none of pathTracer.comp:400-448: the ray is left as it was and the same
    intersection repeats at every later depth (colour multiplied again, Russian roulette drawn again).  The slab kernels carry
    |c - origin|^2 with the ray and must not advance it to the hit point for such an object."""
    planes, spheres = _box_scene(O)
    if which == "sphere":
        spheres[0, 11] = code
    else:
        planes[2, 11] = code
    W, H, spp = 40, 28, 9
    ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
    for flags in (0, B.pt_force_s(1), B.pt_force_s(16), B.PT_GENERIC_KERNEL):
        out = ctx.pathtrace(B.pathtrace_params(W, H, spp, flags=flags), planes=planes, spheres=spheres)
        assert np.array_equal(bits(out), bits(ref)), (which, code, flags)


def test_fast_mode_on_a_scene_with_non_unit_plane_normals(ctx, B, O):
    """The reference uses a plane's normal as given (pathTracer.comp:387): scaling (n, w) of a wall describes the same plane but
    a longer shading normal, and the normalize of :428 / :441 then really rescales.  The fast mode's "unit combination" shortcut
    is for the slab kernels only (axis normals of length exactly 1); this scene takes the generic kernel and must stay within
    the small-size fast tolerance of the oracle — a kernel that skipped the normalize there would be off by tens of units."""
    planes, spheres = _box_scene(O)
    planes[1, 0:4] *= 1.5          # right wall: normal (1.5, 0, 0), w scaled alike
    planes[2, 0:4] *= 0.75         # ceiling
    planes[4, 11] = 3.0            # the back wall refracts (tdir of :441 from a plane normal)
    assert B.pathtrace_scene_class(planes, spheres) == B.PT_SCENE_SPECULAR   # generic; the glass wall: a fast request runs the careful tier
    W, H, spp = 96, 64, 64
    ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_LIBM)[..., :3].astype(np.float64)
    strict = ctx.pathtrace(B.pathtrace_params(W, H, spp), planes=planes, spheres=spheres)
    assert np.array_equal(bits(strict), bits(O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)))
    # both tiers of the generic kernels: as requested (careful) and the fast tier through the measurement switch
    for flags in (0, B.pt_force_s(1), B.pt_force_s(16), B.PT_NO_FAST_GUARD, B.PT_NO_FAST_GUARD | B.pt_force_s(1), B.PT_NO_FAST_GUARD | B.pt_force_s(16)):
        fast = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=flags), planes=planes, spheres=spheres)
        d = fast[..., :3].astype(np.float64) - ref
        rmse, p999 = np.sqrt((d ** 2).mean()), np.percentile(np.sqrt((d ** 2).sum(-1)), 99.9)
        print(f"non-unit normals, fast generic flags={flags}: rmse {rmse:.4f} p99.9 {p999:.3f}")
        assert rmse <= 0.75 and p999 <= 5.0 and abs(d.mean()) < 0.25


@pytest.mark.parametrize("n_planes,n_spheres,n_lights", [(6, 40, 3), (1, 1, 1), (12, 700, 4)])
def test_generic_scene_read_from_memory_is_bit_identical(ctx, B, O, n_planes, n_spheres, n_lights):
    """Large generic scenes are not staged into LDS but read where they lie (csrc/pathtrace.hip: kSceneLdsAutoBytes; scalar loads in
    the intersection loops, a vector load for the material fetch).  Forced either way — MC_PT_SCENE_IN_LDS, MC_PT_SCENE_IN_MEMORY —
    and left to the host, strict renders are bit-identical to the oracle; fast renders of the two paths are two instruction
    sequences (the memory path fetches the next record ahead; the compiler contracts differently) that agree up to forked samples."""
    rng = np.random.default_rng(n_spheres)
    planes, spheres = random_scene(rng, n_planes, n_spheres, n_lights)
    W, H, spp = (24, 16, 6) if n_spheres > 100 else (48, 32, 9)
    ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
    for flags in (0, B.PT_SCENE_IN_LDS, B.PT_SCENE_IN_MEMORY, B.PT_SCENE_IN_MEMORY | B.pt_force_s(1), B.PT_SCENE_IN_MEMORY | B.pt_force_s(16)):
        out = ctx.pathtrace(B.pathtrace_params(W, H, spp, flags=flags), planes=planes, spheres=spheres)
        assert np.array_equal(bits(out), bits(ref)), flags
    f_lds = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=B.PT_SCENE_IN_LDS), planes=planes, spheres=spheres)
    f_mem = ctx.pathtrace(B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=B.PT_SCENE_IN_MEMORY), planes=planes, spheres=spheres)
    d = np.abs(f_lds[..., :3].astype(np.float64) - f_mem[..., :3].astype(np.float64))
    assert np.isfinite(f_mem).all() and (d > 1.0).mean() <= 0.03 and np.median(d) <= 1e-3, ((d > 1.0).mean(), np.median(d))


def test_scene_beyond_the_lds_store_renders_from_memory(ctx, B, O):
    """4000 spheres (192 KB of records: more than a CU's LDS) were MC_ERR_UNSUPPORTED before round 3's memory path; now the fp32
    kernels take them (bit-identical to the oracle on a few pixels), the extended-precision branches and a forced LDS copy still refuse."""
    rng = np.random.default_rng(9)
    planes, spheres = random_scene(rng, 6, 4000, 3)
    W, H, spp = 8, 6, 2
    out = ctx.pathtrace(B.pathtrace_params(W, H, spp), planes=planes, spheres=spheres)
    assert np.array_equal(bits(out), bits(O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)))
    for flags in (B.PT_SCENE_IN_LDS, B.pt_precision(B.PT_PREC_DS)):
        with pytest.raises(B.McError):
            ctx.pathtrace(B.pathtrace_params(W, H, spp, flags=flags), planes=planes, spheres=spheres)


def box_scene(O, n_spheres, rng, lights=1):
    """The reference room with `n_spheres` pairwise disjoint spheres placed on a jittered grid inside it (diffuse, mirror, glass),
    `lights` of them small emitters under the ceiling — what the specialised slab / closed-box / sample-pool kernels take for
    1 .. 8 spheres (round 4; pathTracer.comp:127,403 loop over spheres.length())."""
    planes = O.DEFAULT_PLANES.copy().reshape(6, 12)
    cells = [(x, y, z) for y in (-1.1, 0.4) for z in (-1.6, 0.6) for x in (-1.4, 1.4)]
    order = rng.permutation(len(cells))[:n_spheres]
    spheres = np.zeros((n_spheres, 12), np.float32)
    for k, c in enumerate(order):
        spheres[k, 0:3] = np.float32(cells[c]) + rng.uniform(-0.15, 0.15, 3).astype(np.float32)
        spheres[k, 3] = np.float32(rng.uniform(0.25, 0.6))
        spheres[k, 8:11] = rng.uniform(0.3, 0.999, 3).astype(np.float32)
        spheres[k, 11] = float(rng.choice([1, 2, 3]))
    for k in range(lights):
        spheres[k, 0:3] = np.float32([-1.2 + 2.4 * k / max(1, lights - 1) if lights > 1 else 0.0, 1.55, -0.3])
        spheres[k, 3] = np.float32(0.2)
        spheres[k, 4:7] = rng.uniform(40, 110, 3).astype(np.float32)
        spheres[k, 8:11] = 0
        spheres[k, 11] = 1.0
    return planes, rng.permutation(spheres)      # (the light at any index)


@pytest.mark.parametrize("n_spheres,lights", [(1, 1), (2, 1), (4, 1), (5, 2), (8, 1), (8, 3)])
def test_box_with_one_to_eight_spheres_runs_the_specialised_kernels(ctx, B, O, n_spheres, lights):
    """SURVEY §8(f)4 for the fast path: the axis-aligned box with 1 .. 8 spheres takes the slab / closed-box /
    sample-pool kernels (instantiated per sphere count) instead of the generic one.  Strict: the pool kernel, the round-synchronous
    slab kernel at every width and the generic kernel all equal the oracle bit for bit.  Fast: the stated bound — RMSE <= 0.5 and
    99.9-percentile L2 <= 4 at 500 spp, never edited — on EVERY scene (round 4 asserted 8 from six spheres on, where
    the fast tier measures 4.4 - 5.6).  From FOUR spheres on (round 6; round 5: five) the host renders an MC_PT_MATH_FAST request with the careful tier
    (MC_PT_MATH_FAST_CAREFUL: the same kernels without contraction, division / sqrt / rsq rounded as the reference rounds them —
    fewer differently rounded operations, fewer forked samples: 1.3 - 3.0 on these scenes), reported by mc_pathtrace_select_kernel."""
    rng = np.random.default_rng(40 + 10 * n_spheres + lights)
    planes, spheres = box_scene(O, n_spheres, rng, lights)
    cls = B.pathtrace_scene_class(planes, spheres)
    specular = cls & B.PT_SCENE_SPECULAR         # up to three spheres, more specular surface than the reference scene's: careful tier, too
    assert cls & ~B.PT_SCENE_SPECULAR == B.PT_SCENE_SLAB | B.PT_SCENE_LIGHTS_INSIDE | B.PT_SCENE_SPHERES_DISJOINT | (B.PT_SCENE_MANY_SPHERES if n_spheres >= 4 else 0), cls
    W, H, spp = 40, 24, 37
    ref = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_MC)
    p = B.pathtrace_params(W, H, spp)
    assert B.pathtrace_select_kernel(p, planes, spheres).kernel == B.PT_KERNEL_POOL
    assert np.array_equal(bits(ctx.pathtrace(p, planes=planes, spheres=spheres)), bits(ref))
    for flags in (B.PT_NO_POOL_KERNEL, B.pt_force_s(1), B.pt_force_s(4), B.pt_force_s(16), B.PT_GENERIC_KERNEL):
        out = ctx.pathtrace(B.pathtrace_params(W, H, spp, flags=flags), planes=planes, spheres=spheres)
        assert np.array_equal(bits(out), bits(ref)), flags
    # progressive ranges and an interleaved row tile through the strict pool kernel
    part = ctx.pathtrace(B.pathtrace_params(W, H, spp, sample_begin=0, sample_end=20), planes=planes, spheres=spheres)
    part = ctx.pathtrace(B.pathtrace_params(W, H, spp, sample_begin=20, sample_end=spp), planes=planes, spheres=spheres, acc=part)
    assert np.array_equal(bits(part), bits(ref))
    # fast math: the pool kernel and the round-synchronous closed-box kernel against the oracle with libm, at the sample count the
    # bound is stated for (K2's 500 spp: a forked sample's weight is part of the bound)
    W, H, spp = 300, 200, 500
    libm = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_LIBM)[..., :3].astype(np.float64)
    for flags in (0, B.PT_NO_POOL_KERNEL):
        q = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=flags)
        ki = B.pathtrace_select_kernel(q, planes, spheres)
        assert ki.kernel == (B.PT_KERNEL_BOX if flags else B.PT_KERNEL_POOL)
        assert ki.math_mode == (B.PT_MATH_FAST_CAREFUL if n_spheres >= 4 or specular else B.PT_MATH_FAST)
        d = ctx.pathtrace(q, planes=planes, spheres=spheres)[..., :3].astype(np.float64) - libm
        rmse, p999 = float(np.sqrt((d ** 2).mean())), float(np.percentile(np.sqrt((d ** 2).sum(-1)), 99.9))
        print(f"{n_spheres} spheres / {lights} lights, flags {flags}: fast vs oracle(libm) rmse {rmse:.4f}  p99.9 L2 {p999:.3f}  mean {d.mean():+.5f}")
        far = float((np.sqrt((d ** 2).sum(-1)) > 4.0).mean())
        assert np.isfinite(d).all() and rmse <= 0.5 and abs(d.mean()) < 0.03 and far <= 0.004, (flags, rmse, p999, far)
        assert p999 <= 4.0, (flags, rmse, p999)


@pytest.mark.parametrize("n_spheres,lights,seed,spec", [(4, 1, 305, True), (4, 2, 332, False)])
def test_four_sphere_boxes_that_broke_the_fast_tier_are_rendered_inside_the_bound(ctx, B, O, n_spheres, lights, seed, spec):
    """Round 6: round 5 set the careful tier's threshold at five spheres on a sample of twelve four-sphere boxes.  A
    sample of 32 more (profiles/r06_fast_tier_4_spheres.txt) holds two that the FAST tier renders outside the un-edited bound — these:
    p99.9 5.5 (three specular spheres) and 6.3 (two lights) — so the host now selects the careful tier from four spheres on.  As a
    caller makes the request (MC_PT_MATH_FAST, no flags) both are inside 0.5 / 4 with room (0.50, 0.81); the fast tier forced by the
    measurement switch still shows why (> 4)."""
    planes, spheres = box_scene(O, n_spheres, np.random.default_rng(seed), lights)
    if spec:                                    # tools/fork_census.py ":spec": every sphere that is not a light made specular
        k = 0
        for q in spheres:
            if not q[4:7].any():
                q[11] = 2.0 + (k & 1); k += 1
    W, H, spp = 300, 200, 500
    libm = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_LIBM)[..., :3].astype(np.float64)

    def p999(flags):
        q = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=flags)
        d = ctx.pathtrace(q, planes=planes, spheres=spheres)[..., :3].astype(np.float64) - libm
        return B.pathtrace_select_kernel(q, planes, spheres).math_mode, float(np.sqrt((d ** 2).mean())), float(np.percentile(np.sqrt((d ** 2).sum(-1)), 99.9))

    tier, rmse, p = p999(0)
    assert tier == B.PT_MATH_FAST_CAREFUL and rmse <= 0.5 and p <= 4.0, (tier, rmse, p)
    assert p < 1.5, p                            # measured 0.50 / 0.81
    tier1, _, p1 = p999(B.PT_NO_FAST_GUARD)
    assert tier1 == B.PT_MATH_FAST and p1 > 4.0, (tier1, p1)   # the reason for the threshold, kept visible


@pytest.mark.parametrize("index,why", [(0, "two mirror spheres and a mirror wall"), (30, "one mirror sphere, r = 0.88")])
def test_three_sphere_rooms_that_broke_the_fast_tier_are_rendered_inside_the_bound(ctx, B, O, index, why):
    """Round 6: 36 jittered three-sphere rooms (tools/fast_tolerance_scenes.py --seed 7, profiles/r06_fast_tolerance_scenes.txt) hold two
    that the FAST tier renders outside the un-edited bound — these: p99.9 5.8 and 4.1.  What they share with every worst scene of the
    censuses is mirror surface, so the host renders an MC_PT_MATH_FAST request for a scene with more of it than the reference scene has
    (a mirror wall, or mirror spheres with sum r^2 > 0.65) with the careful tier: MC_PT_SCENE_SPECULAR.  As a caller makes the request both
    are inside 0.5 / 4 with room; the fast tier forced by the measurement switch still shows why."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from fast_tolerance_scenes import scene
    rng = np.random.default_rng(7)
    for _ in range(index + 1):
        planes, spheres = scene(rng, O)
    cls = B.pathtrace_scene_class(planes, spheres)
    assert cls & B.PT_SCENE_SPECULAR and not cls & (B.PT_SCENE_MANY_SPHERES | B.PT_SCENE_LIGHT_ENCLOSED), (cls, why)
    W, H, spp = 300, 200, 500
    libm = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_LIBM)[..., :3].astype(np.float64)

    def p999(flags):
        q = B.pathtrace_params(W, H, spp, math_mode=B.PT_MATH_FAST, flags=flags)
        d = ctx.pathtrace(q, planes=planes, spheres=spheres)[..., :3].astype(np.float64) - libm
        return B.pathtrace_select_kernel(q, planes, spheres).math_mode, float(np.sqrt((d ** 2).mean())), float(np.percentile(np.sqrt((d ** 2).sum(-1)), 99.9))

    tier, rmse, p = p999(0)
    print(f"jittered room {index} ({why}): careful tier rmse {rmse:.4f} p99.9 {p:.3f}")
    assert tier == B.PT_MATH_FAST_CAREFUL and rmse <= 0.5 and p <= 4.0, (tier, rmse, p)
    tier1, _, p1 = p999(B.PT_NO_FAST_GUARD)
    assert tier1 == B.PT_MATH_FAST and p1 > 4.0, (tier1, p1)   # the reason for the rule, kept visible


def test_the_fast_tier_is_measured_not_guaranteed_the_known_room_at_its_limit(ctx, B, O):
    """Round 6, said plainly: after the tier rule was set, 192 more jittered rooms (seeds 9 .. 11) left 76 to the fast tier; 74 read at
    most 3.76 of the bound 4 and TWO read 4.21 and 4.19 — the first is seed 10's room 42: diffuse walls, a glass sphere of r = 0.77 (no larger than the
    reference scene's), a diffuse sphere, the light.  The rule was not bent around it (profiles/r06_fast_tolerance_scenes_validation2.txt;
    DESIGN.md §4: 2 of 174 fast-tier scenes).  Kept here so that the limit stays visible: the host still selects the fast tier, which is
    outside its bound by 5 % on this room; the careful tier, asked for, has an order of magnitude of room."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from fast_tolerance_scenes import scene
    rng = np.random.default_rng(10)
    for _ in range(43):
        planes, spheres = scene(rng, O)
    assert B.pathtrace_scene_class(planes, spheres) & (B.PT_SCENE_SPECULAR | B.PT_SCENE_MANY_SPHERES | B.PT_SCENE_LIGHT_ENCLOSED) == 0
    W, H, spp = 300, 200, 500
    libm = O.pathtrace(W, H, spp, planes=planes, spheres=spheres, math_mode=O.MATH_LIBM)[..., :3].astype(np.float64)

    def p999(mode):
        q = B.pathtrace_params(W, H, spp, math_mode=mode)
        d = ctx.pathtrace(q, planes=planes, spheres=spheres)[..., :3].astype(np.float64) - libm
        return B.pathtrace_select_kernel(q, planes, spheres).math_mode, float(np.sqrt((d ** 2).mean())), float(np.percentile(np.sqrt((d ** 2).sum(-1)), 99.9))

    tier, rmse, p = p999(B.PT_MATH_FAST)
    print(f"seed 10 room 42, fast tier: rmse {rmse:.4f} p99.9 {p:.3f}")
    assert tier == B.PT_MATH_FAST and rmse <= 0.5 and 4.0 < p < 4.5, (tier, rmse, p)     # the known limit: measured 0.244 / 4.207
    tier, rmse, p = p999(B.PT_MATH_FAST_CAREFUL)
    print(f"seed 10 room 42, careful tier: rmse {rmse:.4f} p99.9 {p:.3f}")
    assert tier == B.PT_MATH_FAST_CAREFUL and rmse <= 0.5 and p <= 1.0, (tier, rmse, p)
