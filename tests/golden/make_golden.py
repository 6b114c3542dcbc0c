#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference (pjhusky/vulkan-compute-tests) has no tests and no golden vectors, and its
compute path (GLSL + Vulkan) cannot be built in this environment, so these vectors come from
  (a) the CPU oracle (oracle/, a literal restatement of the shaders) — inputs AND expected outputs, and
  (b) the reference's only artefact, imageForReadme.png (the README's 900x600 render), decoded to RGB and
      also reduced to 20x30 block means; needs /root/reference at generation time only.  It turns out to be
      a 500-spp render of the default scene with the shader's hash RNG: the oracle and the HIP kernels
      reproduce >93 % of its pixels exactly and >99 % within +-1 (the rest fork on the rendering GPU's
      implementation-defined sin/cos/sqrt precision), which pins the path tracer per pixel.
They pin the oracle against regressions on any machine and let the GPU box (which has no /root/reference)
compare the HIP path with committed data.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DEEP = dict(centre=(-0.7436438870371587, 0.13182590420531198), scale=(1e-8, 1e-8 * 2.0 / 3.0))


def main():
    rng = np.random.default_rng(20261003)
    g = {}
    # Mandelbrot fp32: reference view 64x64 M=128 (mandelbrot.comp:38-40) and a zoomed view
    g["mandel_ref_64x64_M128"] = O.mandelbrot_iters(64, 64, 128)
    g["mandel_zoom_view"] = O.make_view(-0.75, 0.1, 0.01, 0.0075)
    g["mandel_zoom_80x60_M300"] = O.mandelbrot_iters(80, 60, 300, view=g["mandel_zoom_view"])
    # two-float variant, deep view (BASELINE K4 view at reduced size)
    g["mandel_ds_view"] = O.make_view(*DEEP["centre"], *DEEP["scale"])
    g["mandel_ds_48x32_M2000"] = O.mandelbrot_iters(48, 32, 2000, view=g["mandel_ds_view"], precision=1)
    # colour LUT, M = 128, kColor {0.1,0.7,0.6,0}
    lut_f, lut_u8 = O.mandel_lut(128)
    g["lut_M128_f32"], g["lut_M128_u8"] = lut_f, lut_u8
    # rand01
    keys = np.concatenate([rng.integers(0, 2**32, size=(256, 3), dtype=np.uint64).astype(np.uint32),
                           np.array([[0, 0, 0], [1, 2, 3], [899, 599, 5999], [0xffffffff] * 3], np.uint32)])
    g["rand01_keys"], g["rand01_out"] = keys, O.rand01(keys)
    # ds primitives incl. cancellation
    n = 512
    hi = (rng.standard_normal(n) * 10.0 ** rng.integers(-4, 4, n)).astype(np.float32)
    lo = (hi * rng.uniform(-1, 1, n) * 2.0 ** -24).astype(np.float32)
    hi2 = (rng.standard_normal(n) * 10.0 ** rng.integers(-4, 4, n)).astype(np.float32)
    lo2 = (hi2 * rng.uniform(-1, 1, n) * 2.0 ** -24).astype(np.float32)
    hi2[:64] = -hi[:64]
    hi2[64:96], lo2[64:96] = hi[64:96], lo[64:96]
    a, b = np.stack([hi, lo], 1), np.stack([hi2, lo2], 1)
    g["ds_a"], g["ds_b"] = a, b
    for op in ("add", "sub", "mul", "compare"):
        g["ds_" + op] = O.ds_op(op, a, b)
    # mc math
    ang = rng.uniform(0, 6.2831855, 512).astype(np.float32)
    unit = rng.uniform(0, 1, 512).astype(np.float32)
    unit[:4] = [0.0, 1.0, 0.5, 1e-30]
    g["mc_angles"], g["mc_unit"] = ang, unit
    g["mc_sin"], g["mc_cos"] = O.mc_math("sin", ang), O.mc_math("cos", ang)
    g["mc_log2"], g["mc_pow045"] = O.mc_math("log2", unit), O.mc_math("pow045", unit)
    # path tracer, default scene, 32x24 @ 8 spp, both math back-ends; plus one progressive half
    g["pt_mc_32x24_spp8"] = O.pathtrace(32, 24, 8, math_mode=O.MATH_MC)
    g["pt_libm_32x24_spp8"] = O.pathtrace(32, 24, 8, math_mode=O.MATH_LIBM)
    g["pt_mc_32x24_spp8_first3"] = O.pathtrace(32, 24, 8, math_mode=O.MATH_MC, sample_end=3)
    np.savez_compressed(os.path.join(OUT, "oracle_vectors.npz"), **g)
    print("wrote oracle_vectors.npz:", {k: v.shape for k, v in g.items()})

    ref_png = "/root/reference/imageForReadme.png"
    if os.path.exists(ref_png):
        from PIL import Image
        img = np.asarray(Image.open(ref_png).convert("RGB")).astype(np.float64)
        assert img.shape == (600, 900, 3)
        blocks = img.reshape(20, 30, 30, 30, 3).mean(axis=(1, 3)).astype(np.float32)
        np.save(os.path.join(OUT, "readme_image_block_means.npy"), blocks)
        # the decoded pixels themselves (expected OUTPUT of the reference for 900x600 @ 500 spp, its default run):
        # a per-pixel known-answer for the whole path-tracer pipeline (sample keys, accumulation order, gamma, u8
        # conversion, 180-degree rotation).  Stored decoded + deflated, not as the reference's PNG file.
        np.savez_compressed(os.path.join(OUT, "readme_image_rgb.npz"), rgb=img.astype(np.uint8))
        print("wrote readme_image_block_means.npy", blocks.shape, "mean RGB", img.mean(axis=(0, 1)))
    else:
        print("reference checkout absent: readme_image_block_means.npy not regenerated")

    # the reference's own lodepng (oracle/_ref, compiled from /root/reference where it lies) on the oracle's default
    # Mandelbrot RGBA8 image: the byte stream INTEGRATION.md route B produces; only its SHA-256 is committed
    if O.ref_lodepng() is not None:
        import hashlib
        _, lut_u8 = O.mandel_lut(128)
        img = np.ascontiguousarray(lut_u8[O.mandelbrot_iters(256, 256, 128)])
        png = O.ref_png_encode(img, 256, 256)
        with open(os.path.join(OUT, "mandelbrot_256_M128_lodepng.sha256"), "w") as f:
            f.write(hashlib.sha256(png).hexdigest() + "  lodepng::encode(RGBA8 of the oracle's 256x256 M=128 Mandelbrot), %d bytes; "
                    "made by tests/golden/make_golden.py\n" % len(png))
        print("wrote mandelbrot_256_M128_lodepng.sha256")


if __name__ == "__main__":
    main()
