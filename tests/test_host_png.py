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


# ---- north_star's "bit-identical PNG": identical RGBA8 pixels + the REFERENCE'S OWN codec (INTEGRATION.md route B) ----------------
# The apps write standard PNGs of exactly the reference's pixels through their own parallel writer; they do not re-implement the
# reference's vendored third-party codec (round 3 carried a source-derived port of it, removed in round 4).  The byte contract is
# met where the reference tree keeps its lodepng::encode call: the codec built from the reference sources where they lie
# (oracle/_ref) turns the pixels this library produces into the pinned bytes.
def opaque(rgb):
    return np.ascontiguousarray(np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], -1))


def test_reference_codec_on_these_pixels_gives_the_pinned_bytes(H, O):
    """oracle/_ref (the reference's lodepng, compiled where it lies) on the oracle's 256 x 256 M = 128 Mandelbrot pixels and on the
    decoded README image reproduces the SHA-256 pins / the README file's 1 057 515 bytes; the apps' own writer round-trips the same
    pixels (another valid byte stream)."""
    import hashlib
    from conftest import GOLDEN
    if O.ref_lodepng() is None:
        pytest.skip("oracle/_ref/liblodepng_ref.so not built (needs the reference checkout)")
    _, lut_u8 = O.mandel_lut(128)
    img = np.ascontiguousarray(lut_u8[O.mandelbrot_iters(256, 256, 128)])
    golden = open(os.path.join(GOLDEN, "mandelbrot_256_M128_lodepng.sha256")).read().split()[0]
    assert hashlib.sha256(O.ref_png_encode(img, 256, 256)).hexdigest() == golden
    assert np.array_equal(decode(encode(H, img, 4)), img)
    rgb = np.load(os.path.join(GOLDEN, "readme_image_rgb.npz"))
    rgb = rgb[list(rgb.keys())[0]]
    png = O.ref_png_encode(opaque(rgb), rgb.shape[1], rgb.shape[0])
    assert len(png) == 1057515
    assert hashlib.sha256(png).hexdigest() == open(os.path.join(GOLDEN, "readme_image_png.sha256")).read().split()[0]


@pytest.mark.parametrize("shape", [(40, 60), (34, 51), (1, 1), (3, 7), (600, 900), (257, 263)])
def test_storage_buffer_straight_to_png_is_the_two_step_file(H, O, shape):
    """Round 6: the apps' storage-buffer route hands the fp32 vec4 buffer to the PNG writer, whose stripe workers convert the rows they
    filter (getRenderedImage's x86 cast + the path tracer's point reflection, odd-width middle column included).  For every thread
    count the file is BYTE FOR BYTE the one the two-step way produces (convertStorage, then encode), and its pixels are the oracle's."""
    h, w = shape
    rng = np.random.default_rng(h * 31 + w)
    buf = (rng.random((h, w, 4), dtype=np.float32) * 300.0 - 20.0).astype(np.float32)     # out-of-range values: the wrap-around cast
    buf[..., 3] = 0.0
    H.mcu_png_encode_storage.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_ubyte)),
                                         C.POINTER(C.c_size_t)]
    H.mcu_convert_storage.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_int]
    for scale, rotate in ((1.0, 1), (255.0 / 280.0, 0), (1.0, 0)):
        exp = O.float_to_rgba8(buf, scale).reshape(h, w, 4)
        if rotate:
            exp = O.rotate180(exp, w, h)
        for threads in (1, 3, 0):
            out, n = C.POINTER(C.c_ubyte)(), C.c_size_t(0)
            assert H.mcu_png_encode_storage(buf.ctypes.data_as(C.c_void_p), w, h, scale, rotate, threads, C.byref(out), C.byref(n)) == 0
            fused = C.string_at(out, n.value)
            H.mcu_free(out)
            assert np.array_equal(decode(fused), exp), (scale, rotate, threads)
            two = np.empty((h, w, 4), np.uint8)
            H.mcu_convert_storage(buf.ctypes.data_as(C.c_void_p), two.ctypes.data_as(C.c_void_p), w, h, scale, rotate, threads)
            assert fused == encode(H, two, threads), (scale, rotate, threads)


@pytest.mark.parametrize("w,h", [(97, 61), (640, 480), (1500, 1100)])
def test_progressive_encoder_gives_the_one_shot_file(H, O, w, h):
    """Round 6: the Mandelbrot app renders its image in row bands and feeds pngwriter::Progressive — the stripe workers filter and deflate
    band k while the device renders band k + 1.  Same stripes, same bytes as the one-shot encoders, however the rows arrived: here the
    buffer the workers read holds garbage until a band is copied in and declared ready (a worker that ran ahead would encode it)."""
    rng = np.random.default_rng(h * 17 + w)
    buf = (rng.random((h, w, 4), dtype=np.float32) * 300.0 - 20.0).astype(np.float32)
    buf[..., 3] = 0.0
    H.mcu_png_encode_storage.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_ubyte)),
                                         C.POINTER(C.c_size_t)]
    H.mcu_png_encode_progressive.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t)]
    rgba = O.float_to_rgba8(buf, 1.0).reshape(h, w, 4).copy()
    for threads in (1, 3, 0):
        out, n = C.POINTER(C.c_ubyte)(), C.c_size_t(0)
        assert H.mcu_png_encode_storage(buf.ctypes.data_as(C.c_void_p), w, h, 1.0, 0, threads, C.byref(out), C.byref(n)) == 0
        one_shot = C.string_at(out, n.value)
        H.mcu_free(out)
        assert one_shot == encode(H, rgba, threads)
        for bands in (1, 2, 7, h):
            for route, image in ((0, buf), (1, rgba)):
                out, n = C.POINTER(C.c_ubyte)(), C.c_size_t(0)
                assert H.mcu_png_encode_progressive(image.ctypes.data_as(C.c_void_p), w, h, 1.0, route, threads, bands, C.byref(out), C.byref(n)) == 0
                got = C.string_at(out, n.value)
                H.mcu_free(out)
                assert got == one_shot, (threads, bands, route)


@pytest.mark.parametrize("route", [0, 1])
def test_an_abandoned_progressive_save_never_reads_rows_that_did_not_arrive(B, route):
    """A streamed save whose render fails (run() throws, the application object is torn down): pngwriter::Progressive must stop its
    workers WITHOUT reading the rows that never arrived — its source may already be gone by then (the apps' storage buffer is a later
    member than the encoder, so it is destroyed first).  The helper maps the image with every row from `ready` on inaccessible, reports the
    first rows and destroys the encoder: a read beyond them is a fault, so the call runs in a child process."""
    import subprocess
    import sys
    lib = os.environ.get("MC_HOSTUTIL_LIB_PATH") or os.path.join(os.path.dirname(B.LIB_PATH), "libmc_hostutil.so")
    code = ("import ctypes as C, sys; H = C.CDLL(sys.argv[1]); "
            "H.mcu_png_progressive_abandon.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32]; "
            "rcs = [H.mcu_png_progressive_abandon(1500, 1100, int(sys.argv[2]), t, ready) for t in (1, 3, 0) for ready in (0, 1, 300, 1099, 1100)]; "
            "sys.exit(1 if any(rcs) else 0)")
    p = subprocess.run([sys.executable, "-c", code, lib, str(route)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, (p.returncode, p.stderr[-500:])
