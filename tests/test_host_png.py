"""CPU tests of the parallel PNG writer (host/pngWriter.cpp): row stripes deflated by several
threads and concatenated into one zlib stream must decode to the same pixels as the serial encoding, for every
thread count, stripe layout and colour type."""
import ctypes as C
import io
import os
import zlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def H(B):
    lib = C.CDLL(os.environ.get("MC_HOSTUTIL_LIB_PATH") or os.path.join(os.path.dirname(B.LIB_PATH), "libmc_hostutil.so"))
    lib.mcu_png_encode_mt.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t)]
    lib.mcu_free.argtypes = [C.c_void_p]
    return lib


def encode(H, img, threads):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    out, n = C.POINTER(C.c_ubyte)(), C.c_size_t(0)
    assert H.mcu_png_encode_mt(img.ctypes.data_as(C.c_void_p), w, h, threads, C.byref(out), C.byref(n)) == 0
    data = C.string_at(out, n.value)
    H.mcu_free(out)
    return data


def decode(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))


def idat_payload(data):
    pos, out = 8, b""
    while pos < len(data):
        ln = int.from_bytes(data[pos:pos + 4], "big")
        typ = data[pos + 4:pos + 8]
        assert zlib.crc32(data[pos + 4:pos + 8 + ln]) == int.from_bytes(data[pos + 8 + ln:pos + 12 + ln], "big")
        if typ == b"IDAT":
            out += data[pos + 8:pos + 8 + ln]
        pos += 12 + ln
    return out


@pytest.mark.parametrize("shape", [(600, 900), (1, 1), (3, 5000), (2000, 17), (257, 263)])
def test_parallel_png_equals_serial_pixels(H, O, shape):
    h, w = shape
    rng = np.random.default_rng(h * 7 + w)
    smooth = (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8)
    img = np.stack([smooth, smooth[::-1], rng.integers(0, 256, (h, w), dtype=np.uint8), np.full((h, w), 255, np.uint8)], -1)
    serial = encode(H, img, 1)
    assert np.array_equal(decode(serial), img)
    for threads in (2, 3, 8, 0):
        data = encode(H, img, threads)
        assert np.array_equal(decode(data), img), threads
        # the concatenated stripes form ONE valid zlib stream with a correct combined Adler-32
        raw = zlib.decompress(idat_payload(data))
        assert len(raw) == h * (w * 3 + 1)
        if O.ref_lodepng() is not None:       # and the reference's own decoder accepts it
            assert np.array_equal(O.ref_png_decode(data), img)


def test_parallel_png_with_alpha(H):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (300, 200, 4), dtype=np.uint8)
    for threads in (1, 4):
        assert np.array_equal(decode(encode(H, img, threads)), img)


def test_parallel_png_large_image_is_faster_or_equal_size_sane(H, O):
    """A Mandelbrot-like 3200x2400 image: stripe-parallel output stays within 2 % of the serial size."""
    _, lut_u8 = O.mandel_lut(256)
    img = np.ascontiguousarray(lut_u8[O.mandelbrot_iters(1600, 1200, 256)])
    a, b = encode(H, img, 1), encode(H, img, 8)
    assert np.array_equal(decode(b), img)
    assert len(b) < len(a) * 1.02 + 4096


# ---- the reference-compatible encoder (host/pngReference.cpp): the BYTES the reference's codec writes -----------------------
@pytest.fixture(scope="module")
def HR(H):
    H.mcu_png_encode_reference.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t)]
    return H


def encode_reference(H, img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    out, n = C.POINTER(C.c_ubyte)(), C.c_size_t(0)
    rc = H.mcu_png_encode_reference(img.ctypes.data_as(C.c_void_p), w, h, C.byref(out), C.byref(n))
    if rc:
        return None
    data = C.string_at(out, n.value)
    H.mcu_free(out)
    return data


def opaque(rgb):
    return np.ascontiguousarray(np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], -1))


def reference_encoder_cases(O):
    rng = np.random.default_rng(1)
    cases = {"noise": opaque(rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)),
             "1x1": opaque(np.array([[[10, 20, 30]]], np.uint8)),
             "grey ramp (8-bit grey)": opaque(np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, 2).repeat(40, 0)),
             "black / white (1-bit grey)": opaque((rng.integers(0, 2, (33, 45, 1), dtype=np.uint8) * 255).repeat(3, 2)),
             "4-bit grey": opaque((rng.integers(0, 16, (20, 31, 1), dtype=np.uint8) * 17).repeat(3, 2)),
             "2-bit grey": opaque((rng.integers(0, 4, (20, 31, 1), dtype=np.uint8) * 85).repeat(3, 2)),
             "all zero": opaque(np.zeros((100, 300, 3), np.uint8))}
    for n_col, shape in ((2, (17, 23)), (5, (50, 70)), (16, (40, 40)), (17, (40, 40)), (100, (80, 90)), (256, (64, 64)), (257, (64, 64))):
        pal = rng.integers(0, 256, (n_col, 3), dtype=np.uint8)
        idx = rng.integers(0, n_col, shape)
        idx.flat[:n_col] = np.arange(n_col)               # every colour present
        cases[f"{n_col} colours"] = opaque(pal[idx])
    cases["3 colours, 5 pixels (too few for a palette)"] = opaque(np.array([[[1, 2, 3], [4, 5, 6], [7, 8, 9], [1, 2, 3], [4, 5, 6]]], np.uint8))
    g = (np.outer(np.linspace(0, 1, 200), np.linspace(0, 1, 300)) * 255).astype(np.uint8)
    cases["smooth rgb"] = opaque(np.stack([g, g[::-1], 255 - g], -1))
    _, lut_u8 = O.mandel_lut(128)
    cases["mandelbrot 256x256 M=128 (palette)"] = np.ascontiguousarray(lut_u8[O.mandelbrot_iters(256, 256, 128)])
    _, lut_u8 = O.mandel_lut(1000)
    cases["mandelbrot 800x600 M=1000 (several deflate blocks)"] = np.ascontiguousarray(lut_u8[O.mandelbrot_iters(800, 600, 1000)])
    return cases


def test_reference_compatible_encoder_matches_lodepng(HR, O):
    """Byte for byte against the reference's own codec (oracle/_ref, built from the reference sources where they lie) over the
    colour models, bit depths, filter choices and block counts the encoder can produce."""
    if O.ref_lodepng() is None:
        pytest.skip("oracle/_ref/liblodepng_ref.so not built (needs the reference checkout)")
    for name, img in reference_encoder_cases(O).items():
        mine = encode_reference(HR, img)
        assert mine is not None and mine == O.ref_png_encode(img, img.shape[1], img.shape[0]), name
        assert np.array_equal(decode(mine), img), name


def test_reference_compatible_encoder_golden_bytes(HR, O):
    """Without the reference at hand: the SHA-256 pinned from the reference's codec for the oracle's default Mandelbrot image
    (tests/golden/make_golden.py), and the reference's README image — 900 x 600, 1 057 515 bytes — reproduced from its pixels."""
    import hashlib
    from conftest import GOLDEN
    _, lut_u8 = O.mandel_lut(128)
    img = np.ascontiguousarray(lut_u8[O.mandelbrot_iters(256, 256, 128)])
    golden = open(os.path.join(GOLDEN, "mandelbrot_256_M128_lodepng.sha256")).read().split()[0]
    assert hashlib.sha256(encode_reference(HR, img)).hexdigest() == golden
    rgb = np.load(os.path.join(GOLDEN, "readme_image_rgb.npz"))
    rgb = rgb[list(rgb.keys())[0]]
    png = encode_reference(HR, opaque(rgb))
    assert len(png) == 1057515
    assert hashlib.sha256(png).hexdigest() == open(os.path.join(GOLDEN, "readme_image_png.sha256")).read().split()[0]


def test_reference_compatible_encoder_refuses_alpha(HR):
    img = np.full((4, 4, 4), 255, np.uint8)
    img[2, 3, 3] = 254
    assert encode_reference(HR, img) is None              # the apps then take the parallel writer (ComputeApp::writePng)
