"""CPU tests of the ORACLE itself: closed-form facts derivable from the reference source,
double-precision cross-checks, the committed golden vectors, and the reference's only artefact
(imageForReadme.png, as block means).  No GPU, no product code."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, REFERENCE


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ---- closed-form pins ---------------------------------------------------------------------------------
def test_lut_known_bytes(O):
    """n=0 -> (231,116,25), n=1 -> (231,121,26), n=128 -> (241,116,25) at M=128 with the x86 wrap."""
    f, u = O.mandel_lut(128)
    assert list(u[0]) == [231, 116, 25, 255]
    assert list(u[1]) == [231, 121, 26, 255]
    assert list(u[128]) == [241, 116, 25, 255]
    # f32 and f64 evaluation of the palette agree on every byte (the LUT is numerically stable)
    n = np.arange(129, dtype=np.float64)[:, None]
    t = n / 128.0
    d = np.array([0.1, 0.7, 0.6]); e = np.array([-0.2, -0.3, -0.5]); ff = np.array([2.1, 2.0, 3.0]); g = np.array([0, 0.1, 0])
    c64 = 255.0 * (np.float32(d).astype(np.float64) + np.float32(e).astype(np.float64) * np.cos(6.28318 * (np.float32(ff).astype(np.float64) * t + np.float32(g).astype(np.float64))))
    assert np.array_equal(np.trunc(c64).astype(np.int64) & 0xff, u[:, :3].astype(np.int64))
    assert np.all(f[:, 3] == 1.0)
    # 74 of the 387 channel values are outside [0,255] and wrap
    assert int(((c64 < 0) | (c64 >= 256)).sum()) == 74


def test_x86_cast_semantics(O):
    v = np.array([[-25.5, 280.5, 255.99, 1.0], [0.5, 255.5, -0.99, 1.0], [1e10, -1e10, np.nan, 1.0]], np.float32)
    out = O.float_to_rgba8(v, 1.0)
    assert out.tolist() == [[231, 24, 255, 255], [0, 255, 0, 255], [0, 0, 0, 255]]


def test_rotate180_matches_reference_loop(O):
    for W, H in [(6, 4), (7, 3), (1, 5), (2, 2)]:
        a = np.arange(W * H * 4, dtype=np.uint8).reshape(H, W, 4)
        got = O.rotate180(a, W, H)
        exp = a.copy()
        p = exp.reshape(-1, 4)
        for y in range(H):                       # pathtracerApp.h:237-243 restated in Python
            for x in range(W // 2):
                f, t = x + y * W, (W - 1) - x + ((H - 1) - y) * W
                p[[f, t]] = p[[t, f]]
        assert np.array_equal(got, exp)
        if W % 2 == 0:
            assert np.array_equal(got, a[::-1, ::-1])


def test_rand01_is_the_integer_hash(O):
    """pathTracer.comp:107-110 restated with numpy uint32 arithmetic; scale is exactly 2^-32 and 1.0 is reachable."""
    rng = np.random.default_rng(0)
    k = rng.integers(0, 2**32, size=(1000, 3), dtype=np.uint64).astype(np.uint32)
    x = k.copy()
    with np.errstate(over="ignore"):
        for _ in range(3):
            x = ((x >> np.uint32(8)) ^ x[:, [1, 2, 0]]) * np.uint32(1103515245)
    exp = x.astype(np.float32) * np.float32(2.0 ** -32)
    assert np.array_equal(bits(O.rand01(k)), bits(exp))
    assert O.rand01([[0, 0, 0]]).tolist() == [[0.0, 0.0, 0.0]]
    assert float(O.rand01(k).max()) <= 1.0


def test_ds_primitives_against_float64(O):
    rng = np.random.default_rng(1)
    n = 5000
    hi = rng.standard_normal(n).astype(np.float32)
    lo = (hi * rng.uniform(-1, 1, n) * 2.0 ** -25).astype(np.float32)
    hi2 = rng.standard_normal(n).astype(np.float32)
    lo2 = (hi2 * rng.uniform(-1, 1, n) * 2.0 ** -25).astype(np.float32)
    a, b = np.stack([hi, lo], 1), np.stack([hi2, lo2], 1)
    va, vb = hi.astype(np.float64) + lo, hi2.astype(np.float64) + lo2
    for op, ref in (("add", va + vb), ("sub", va - vb), ("mul", va * vb)):
        r = O.ds_op(op, a, b)
        got = r[:, 0].astype(np.float64) + r[:, 1]
        # ~44 significant bits relative to the operands' magnitude (cancellation loses relative, not absolute, accuracy)
        scale = np.abs(va * vb) if op == "mul" else np.abs(va) + np.abs(vb)
        err = np.abs(got - ref) / scale
        assert err.max() < 2.0 ** -42, (op, err.max())
    c = O.ds_op("compare", a, b)[:, 0]
    assert np.array_equal(c, np.sign(va - vb).astype(np.float32))
    # ds_set(x) op ds_set(0) identities
    z = np.zeros_like(a)
    assert np.array_equal(bits(O.ds_op("add", a, z)[:, 0] + O.ds_op("add", a, z)[:, 1]), bits((va).astype(np.float32)))


def test_mandelbrot_workload_facts(O):
    """~28 % interior pixels, ~39.8 loop bodies/px at M=128, ~281 at M=1000 (reduced sizes)."""
    it = O.mandelbrot_iters(500, 500, 128)
    assert abs((it == 128).mean() - 0.2806) < 0.003
    assert abs(O.mandel_pixel_iters(it, 128) / it.size - 39.83) < 0.3
    it = O.mandelbrot_iters(400, 300, 1000)
    assert abs(O.mandel_pixel_iters(it, 1000) / it.size - 281.3) < 2.0


def test_mandelbrot_f32_against_numpy_restatement(O):
    """Independent numpy fp32 restatement of mandelbrot.comp:30-46 (vectorised, unfused)."""
    W, H, M = 96, 64, 80
    gx, gy = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    x, y = gx / np.float32(W), gy / np.float32(H)
    cx = np.float32(-0.445) + (x - np.float32(0.5)) * np.float32(2.34)
    cy = np.float32(0.0) + (y - np.float32(0.5)) * np.float32(2.34)
    zx = np.zeros_like(cx); zy = np.zeros_like(cx); n = np.zeros(cx.shape, np.uint32); alive = np.ones(cx.shape, bool)
    with np.errstate(all="ignore"):
        for _ in range(M):
            nzx = (zx * zx - zy * zy) + cx
            nzy = (np.float32(2.0) * zx) * zy + cy
            zx, zy = nzx, nzy
            esc = (zx * zx + zy * zy) > np.float32(2.0)
            alive &= ~esc
            n += alive
    assert np.array_equal(O.mandelbrot_iters(W, H, M), n)


def test_ds_mandelbrot_agrees_with_float64_away_from_chaos(O):
    """At a moderate zoom the two-float iteration counts equal a float64 evaluation for almost every pixel
    (they differ only where rounding flips a chaotic orbit) while plain fp32 is already badly quantised."""
    W, H, M = 64, 48, 400
    c, s = (-0.7436438870371587, 0.13182590420531198), (1e-6, 0.75e-6)
    ds = O.mandelbrot_iters(W, H, M, view=O.make_view(c[0], c[1], s[0], s[1]), precision=1).astype(np.int64)
    f32 = O.mandelbrot_iters(W, H, M, view=O.make_view(c[0], c[1], s[0], s[1]), precision=0).astype(np.int64)
    gx, gy = np.meshgrid(np.arange(W), np.arange(H))
    x = (gx.astype(np.float32) / np.float32(W)).astype(np.float64) - 0.5
    y = (gy.astype(np.float32) / np.float32(H)).astype(np.float64) - 0.5
    cx, cy = c[0] + x * s[0], c[1] + y * s[1]
    zx = np.zeros_like(cx); zy = np.zeros_like(cx); n = np.zeros(cx.shape, np.int64); alive = np.ones(cx.shape, bool)
    with np.errstate(all="ignore"):
        for _ in range(M):
            zx, zy = zx * zx - zy * zy + cx, 2 * zx * zy + cy
            alive &= ~((zx * zx + zy * zy) > 2.0)
            n += alive
    assert (ds == n).mean() > 0.97
    assert (f32 == n).mean() < (ds == n).mean()


def test_mc_math_accuracy_vs_libm(O):
    x = np.random.default_rng(2).uniform(0, 6.2831855, 200000).astype(np.float32)
    assert np.max(np.abs(O.mc_math("sin", x).astype(np.float64) - np.sin(x.astype(np.float64)))) < 2.5e-7
    assert np.max(np.abs(O.mc_math("cos", x).astype(np.float64) - np.cos(x.astype(np.float64)))) < 2.5e-7
    u = np.random.default_rng(3).uniform(0, 1, 200000).astype(np.float32)
    p = O.mc_math("pow045", u).astype(np.float64)
    ref = u.astype(np.float64) ** np.float64(np.float32(0.45))
    assert np.max(np.abs(p - ref) / np.maximum(ref, 1e-30)) < 1e-6
    assert O.mc_math("pow045", np.array([0.0, 1.0], np.float32)).tolist() == [0.0, 1.0]


# ---- committed golden vectors ---------------------------------------------------------------------------
def test_golden_vectors(O, G):
    assert np.array_equal(O.mandelbrot_iters(64, 64, 128), G["mandel_ref_64x64_M128"])
    assert np.array_equal(O.mandelbrot_iters(80, 60, 300, view=G["mandel_zoom_view"]), G["mandel_zoom_80x60_M300"])
    assert np.array_equal(O.mandelbrot_iters(48, 32, 2000, view=G["mandel_ds_view"], precision=1), G["mandel_ds_48x32_M2000"])
    f, u = O.mandel_lut(128)
    assert np.array_equal(bits(f), bits(G["lut_M128_f32"])) and np.array_equal(u, G["lut_M128_u8"])
    assert np.array_equal(bits(O.rand01(G["rand01_keys"])), bits(G["rand01_out"]))
    for op in ("add", "sub", "mul", "compare"):
        assert np.array_equal(bits(O.ds_op(op, G["ds_a"], G["ds_b"])), bits(G["ds_" + op])), op
    assert np.array_equal(bits(O.mc_math("sin", G["mc_angles"])), bits(G["mc_sin"]))
    assert np.array_equal(bits(O.mc_math("cos", G["mc_angles"])), bits(G["mc_cos"]))
    assert np.array_equal(bits(O.mc_math("log2", G["mc_unit"])), bits(G["mc_log2"]))
    assert np.array_equal(bits(O.mc_math("pow045", G["mc_unit"])), bits(G["mc_pow045"]))
    assert np.array_equal(bits(O.pathtrace(32, 24, 8, math_mode=O.MATH_MC)), bits(G["pt_mc_32x24_spp8"]))
    # libm results may move by an ulp between glibc versions: values, not bits
    assert np.allclose(O.pathtrace(32, 24, 8, math_mode=O.MATH_LIBM), G["pt_libm_32x24_spp8"], atol=0.51)


def test_pathtrace_progressive_and_tiles_in_the_oracle(O, G):
    whole = G["pt_mc_32x24_spp8"]
    part = O.pathtrace(32, 24, 8, math_mode=O.MATH_MC, sample_end=3)
    assert np.array_equal(bits(part), bits(G["pt_mc_32x24_spp8_first3"]))
    part = O.pathtrace(32, 24, 8, math_mode=O.MATH_MC, sample_begin=3, sample_end=8, acc=part)
    assert np.array_equal(bits(part), bits(whole))
    tile = O.pathtrace(32, 24, 8, math_mode=O.MATH_MC, row_begin=5, row_end=9)
    assert np.array_equal(bits(tile), bits(whole[5:9]))
    assert np.all(whole[..., 3] == 0.0)          # alpha stays 0 in the buffer (pathTracer.comp:452-453)
    assert whole[..., :3].min() >= 0.5 and whole[..., :3].max() <= 255.5


def test_pathtrace_op_counters(O):
    """The algorithmic work per sample frozen in DESIGN.md/bench.py comes from these counters."""
    _, c = O.pathtrace(90, 60, 8, counts=True)
    s = c["samples"]
    assert s == 90 * 60 * 8
    flops = (c["add"] + c["mul"] + c["div"] + c["sqrt"] + c["trig"] + c["pow"]) / s
    assert 3600 < flops < 4000
    assert 17.0 < c["intersect_calls"] / s < 18.5
    assert 9.3 < c["bounces"] / s < 10.3
    assert abs(c["iop"] / s - 27 * (1 + c["bounces"] / s)) < 1e-6     # one rand01 per sample + one per bounce


def test_pathtrace_statistics_match_the_reference_image(O):
    """imageForReadme.png (900x600, the reference's only artefact) as 30x30-pixel block means vs the oracle at
    16 spp: correct orientation (the mirrored image is far off) and matching brightness up to the low-spp
    clamp/gamma bias."""
    blocks = np.load(os.path.join(GOLDEN, "readme_image_block_means.npy")).astype(np.float64)
    buf = O.pathtrace(900, 600, 16)
    img = O.rotate180(O.float_to_rgba8(buf, 1.0).reshape(600, 900, 4), 900, 600)[..., :3].astype(np.float64)
    mine = img.reshape(20, 30, 30, 30, 3).mean(axis=(1, 3))
    mirrored = img[:, ::-1].reshape(20, 30, 30, 30, 3).mean(axis=(1, 3))
    rmse = np.sqrt(((mine - blocks) ** 2).mean())
    rmse_m = np.sqrt(((mirrored - blocks) ** 2).mean())
    assert rmse < 8.0 and rmse_m > 30.0, (rmse, rmse_m)
    assert np.all(np.abs(mine.mean(axis=(0, 1)) - blocks.mean(axis=(0, 1))) < 9.0)


@pytest.mark.skipif(not os.path.exists(REFERENCE), reason="reference checkout absent (GPU box)")
def test_reference_lodepng_reencodes_its_own_artefact(O):
    """oracle/_ref (the reference's lodepng compiled where it lies): encode(decode(imageForReadme.png)) is the
    file itself, byte for byte — pins the PNG stage of the output contract."""
    if O.ref_lodepng() is None:
        pytest.skip("oracle/_ref not built")
    data = open(os.path.join(REFERENCE, "imageForReadme.png"), "rb").read()
    img = O.ref_png_decode(data)
    assert img.shape == (600, 900, 4) and np.all(img[..., 3] == 255)
    assert O.ref_png_encode(img, 900, 600) == data


def test_oracle_reproduces_the_reference_image_per_pixel(O):
    """The reference's README image is its default run (900x600, 500 spp, `./pocketpt-mac`, README.md:24).  A band
    of 24 storage rows rendered by the oracle at 500 spp, pushed through the host post-process
    (pathtracerApp.h:202-243), must reproduce the corresponding image rows: >90 % of the pixels exactly and
    >98 % within +-1 (forks come from the rendering GPU's implementation-defined transcendental precision)."""
    rgb = np.load(os.path.join(GOLDEN, "readme_image_rgb.npz"))["rgb"].astype(np.int32)   # (600, 900, 3) final image
    W, H, spp = 900, 600, 500
    for mode in (O.MATH_LIBM, O.MATH_MC):
        r0, r1 = 300, 312 if mode == O.MATH_MC else 324
        buf = O.pathtrace(W, H, spp, math_mode=mode, row_begin=r0, row_end=r1)
        u8 = O.float_to_rgba8(buf, 1.0).reshape(r1 - r0, W, 4)[..., :3].astype(np.int32)
        # final image pixel (R, C) = storage pixel (row H-1-R, column W-1-C): the 180-degree rotation
        exp = rgb[H - r1:H - r0][::-1, ::-1]
        d = np.abs(u8 - exp).max(-1)
        assert (d == 0).mean() > 0.90 and (d <= 1).mean() > 0.98, ((d == 0).mean(), (d <= 1).mean())
