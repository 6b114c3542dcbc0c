"""GPU tests of the fused render + post-process entry points (mc_*_render_rgba8) and of the apps' end-to-end
options: the RGBA8 image produced on the device equals the reference's host post-process of the storage buffer."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rgba8(B, ctx, fn, *args):
    return fn(ctx._h, *args)


def test_render_rgba8_equals_host_postprocess(ctx, B, O):
    L = B.lib()
    L.mc_mandelbrot_render_rgba8.argtypes = [C.c_void_p, C.POINTER(B.MandelbrotParams), C.c_void_p]
    L.mc_pathtrace_render_rgba8.argtypes = [C.c_void_p, C.POINTER(B.PathtraceParams), C.c_void_p, C.c_uint32, C.c_void_p,
                                            C.c_uint32, C.c_void_p]
    # Mandelbrot
    W, H, M = 333, 211, 200
    p = B.mandelbrot_params(W, H, max_iter=M)
    out = np.empty((H, W, 4), np.uint8)
    assert L.mc_mandelbrot_render_rgba8(ctx._h, C.byref(p), out.ctypes.data_as(C.c_void_p)) == 0
    _, lut_u8 = O.mandel_lut(M)
    assert np.array_equal(out, lut_u8[O.mandelbrot_iters(W, H, M)])
    # ... and in contiguous row bands (round 6, the app's streamed save): each band lands where the whole image has it; an interleaved
    # tile is refused
    banded = np.zeros((H, W, 4), np.uint8)
    for r0, r1 in ((0, 64), (64, 65), (65, 200), (200, H)):
        pb = B.mandelbrot_params(W, H, max_iter=M, row_begin=r0, row_end=r1)
        assert L.mc_mandelbrot_render_rgba8(ctx._h, C.byref(pb), banded[r0:r1].ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(banded, out)
    pb = B.mandelbrot_params(W, H, max_iter=M, row_begin=0, row_end=H, row_block=8, row_stride=16)
    assert L.mc_mandelbrot_render_rgba8(ctx._h, C.byref(pb), banded.ctypes.data_as(C.c_void_p)) == 1
    # path tracer (odd width: the reference's middle-column quirk included)
    for W, H in ((48, 32), (51, 30)):
        q = B.pathtrace_params(W, H, 5)
        planes, spheres = B.default_scene()
        out = np.empty((H, W, 4), np.uint8)
        assert L.mc_pathtrace_render_rgba8(ctx._h, C.byref(q), planes.ctypes.data_as(C.c_void_p), 6,
                                           spheres.ctypes.data_as(C.c_void_p), 3, out.ctypes.data_as(C.c_void_p)) == 0
        ref = O.pathtrace(W, H, 5, math_mode=O.MATH_MC)
        assert np.array_equal(out, O.rotate180(O.float_to_rgba8(ref, 1.0).reshape(H, W, 4), W, H))
    # tiles / partial sample ranges are rejected
    q = B.pathtrace_params(48, 32, 5, sample_end=3)
    assert L.mc_pathtrace_render_rgba8(ctx._h, C.byref(q), planes.ctypes.data_as(C.c_void_p), 6,
                                       spheres.ctypes.data_as(C.c_void_p), 3, out.ctypes.data_as(C.c_void_p)) == 1


@pytest.mark.parametrize("precision", ["f32", "ds"])
def test_banded_render_is_the_blocking_render_band_by_band(ctx, B, O, precision):
    """mc_mandelbrot_render_banded (round 6): the image in pipelined row bands — band k + 1 launched on a second stream before band k has
    finished — with a callback as each band has arrived on the host.  Same bytes as mc_mandelbrot_render / mc_mandelbrot_render_rgba8
    (the kernels are tiling-invariant), the callback hears the band ends in order, a row range of the image works, and
    mc_context_last_timing answers afterwards."""
    W, H, M = 333, 211, 200
    prec = B.PRECISION_DS if precision == "ds" else B.PRECISION_F32
    kw = dict(max_iter=M, precision=prec, centre=(-0.7436438870371587, 0.1318259042053119), scale=(3e-4, 2e-4)) if precision == "ds" else dict(max_iter=M)
    whole, _ = ctx.mandelbrot(B.mandelbrot_params(W, H, **kw), want_iters=False)
    whole8 = ctx.convert_rgba8(whole, 255.0)
    for band_rows in (1000, 64, 37, 1):
        for rgba8 in (False, True):
            img, heard = ctx.mandelbrot_banded(B.mandelbrot_params(W, H, **kw), band_rows, rgba8=rgba8)
            want = whole8 if rgba8 else whole
            assert img.dtype == want.dtype and np.array_equal(img.view(np.uint8), want.view(np.uint8)), (band_rows, rgba8)
            bands = (H + band_rows - 1) // band_rows
            assert heard == [H * (b + 1) // bands for b in range(bands)], (band_rows, heard)
            k_ms, c_ms = ctx.last_timing()
            assert k_ms > 0 and c_ms >= 0
    img, heard = ctx.mandelbrot_banded(B.mandelbrot_params(W, H, row_begin=40, row_end=171, **kw), 50)
    assert np.array_equal(img.view(np.uint32), whole[40:171].view(np.uint32)) and heard == [40 + 131 * (b + 1) // 3 for b in range(3)]
    with pytest.raises(B.McError):
        ctx.mandelbrot_banded(B.mandelbrot_params(W, H, row_block=8, row_stride=16, **kw), 64)
    # the blocking calls' timing is their own again afterwards
    ctx.mandelbrot(B.mandelbrot_params(W, H, **kw), want_iters=False)
    assert ctx.last_timing()[0] > 0


def test_multi_rgba8_one_device(ctx, B, O, monkeypatch):
    L = B.lib()
    L.mc_multi_mandelbrot_render_rgba8.argtypes = [C.c_void_p, C.POINTER(B.MandelbrotParams), C.c_void_p]
    L.mc_multi_pathtrace_render_rgba8.argtypes = [C.c_void_p, C.POINTER(B.PathtraceParams), C.c_void_p, C.c_uint32, C.c_void_p,
                                                  C.c_uint32, C.c_void_p]
    monkeypatch.setenv("MC_MULTI_FORCE_RCCL", "1")
    with B.Multi(1) as m:
        p = B.mandelbrot_params(200, 120, max_iter=150)
        out = np.empty((120, 200, 4), np.uint8)
        assert L.mc_multi_mandelbrot_render_rgba8(m._h, C.byref(p), out.ctypes.data_as(C.c_void_p)) == 0
        _, lut_u8 = O.mandel_lut(150)
        assert np.array_equal(out, lut_u8[O.mandelbrot_iters(200, 120, 150)])
        q = B.pathtrace_params(40, 24, 4)
        planes, spheres = B.default_scene()
        out = np.empty((24, 40, 4), np.uint8)
        assert L.mc_multi_pathtrace_render_rgba8(m._h, C.byref(q), planes.ctypes.data_as(C.c_void_p), 6,
                                                 spheres.ctypes.data_as(C.c_void_p), 3, out.ctypes.data_as(C.c_void_p)) == 0
        ref = O.pathtrace(40, 24, 4, math_mode=O.MATH_MC)
        assert np.array_equal(out, O.rotate180(O.float_to_rgba8(ref, 1.0).reshape(24, 40, 4), 40, 24))


def test_apps_gpu_postprocess_and_precision_options(B, O, tmp_path):
    from PIL import Image
    bindir = os.path.join(os.path.dirname(os.path.dirname(B.LIB_PATH)), "bin")
    # same PNG pixels with and without the device-side post-process, serial and parallel deflate
    imgs = []
    for extra in ([], ["--gpu-postprocess"], ["--gpu-postprocess", "--png-threads", "1"], ["--gpus", "1", "--gpu-postprocess"]):
        r = subprocess.run([os.path.join(bindir, "pathtracer"), "4", "40", "--out", "o.png", "--quiet"] + extra,
                           capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        imgs.append(np.asarray(Image.open(tmp_path / "o.png").convert("RGBA")))
    ref = O.pathtrace(60, 40, 4, math_mode=O.MATH_MC)
    exp = O.rotate180(O.float_to_rgba8(ref, 1.0).reshape(40, 60, 4), 60, 40)
    for im in imgs:
        assert np.array_equal(im, exp)
    r = subprocess.run([os.path.join(bindir, "mandelbrot"), "--width", "300", "--height", "200", "--gpu-postprocess", "--quiet"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0
    _, lut_u8 = O.mandel_lut(128)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "mandelbrot.png").convert("RGBA")), lut_u8[O.mandelbrot_iters(300, 200, 128)])
    # the precision experiment from the command line
    r = subprocess.run([os.path.join(bindir, "pathtracer"), "4", "32", "--large-sphere-walls", "--sphere-precision", "ds", "--quiet"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0
    img = np.asarray(Image.open(tmp_path / "pathtracer.png").convert("RGBA"))
    ref = O.pathtrace(48, 32, 4, planes=O.LARGE_SPHERE_PLANES, spheres=O.LARGE_SPHERE_SPHERES, math_mode=O.MATH_MC, precision=O.PREC_DS)
    assert np.array_equal(img, O.rotate180(O.float_to_rgba8(ref, 1.0).reshape(32, 48, 4), 48, 32))
    # Mandelbrot two-float from the command line
    r = subprocess.run([os.path.join(bindir, "mandelbrot"), "--width", "64", "--height", "48", "--max-iter", "600", "--precision", "ds",
                        "--centre", "-0.7436438870371587", "0.13182590420531198", "--scale", "1e-8", "6.666666666666667e-9", "--quiet"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0
    _, lut_u8 = O.mandel_lut(600)
    it = O.mandelbrot_iters(64, 48, 600, view=O.make_view(-0.7436438870371587, 0.13182590420531198, 1e-8, 6.666666666666667e-9), precision=1)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "mandelbrot.png").convert("RGBA")), lut_u8[it])


def test_apps_pixels_through_the_reference_codec_give_the_reference_bytes(B, O, tmp_path):
    """north_star: "bit-identical PNG".  The apps' files decode to exactly the reference's RGBA8 pixels, and the reference's OWN codec
    (oracle/_ref, built from the reference sources where they lie — the lodepng::encode call a reference tree keeps, INTEGRATION.md
    route B) turns those pixels into the pinned bytes: the SHA-256 frozen from it for the 256 x 256, M = 128 Mandelbrot."""
    import hashlib
    from PIL import Image
    from conftest import GOLDEN
    bindir = os.path.join(os.path.dirname(os.path.dirname(B.LIB_PATH)), "bin")
    r = subprocess.run([os.path.join(bindir, "mandelbrot"), "--width", "256", "--height", "256", "--quiet"], capture_output=True,
                       text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.ascontiguousarray(np.asarray(Image.open(tmp_path / "mandelbrot.png").convert("RGBA")))
    _, lut_u8 = O.mandel_lut(128)
    assert np.array_equal(got, lut_u8[O.mandelbrot_iters(256, 256, 128)])
    if O.ref_lodepng() is not None:
        golden = open(os.path.join(GOLDEN, "mandelbrot_256_M128_lodepng.sha256")).read().split()[0]
        assert hashlib.sha256(O.ref_png_encode(got, 256, 256)).hexdigest() == golden
    r = subprocess.run([os.path.join(bindir, "pathtracer"), "8", "40", "--quiet"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    ref = O.pathtrace(60, 40, 8, math_mode=O.MATH_MC)
    exp = O.rotate180(O.float_to_rgba8(ref, 1.0).reshape(40, 60, 4), 60, 40)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "pathtracer.png").convert("RGBA")), exp)
    r = subprocess.run([os.path.join(bindir, "pathtracer"), "8", "40", "--quiet", "--fast-png", "--out", "fast.png"], capture_output=True,
                       text=True, cwd=tmp_path)   # (the round-3 flag is still accepted)
    assert r.returncode == 0
    assert np.array_equal(np.asarray(Image.open(tmp_path / "fast.png").convert("RGBA")), exp)


def test_apps_at_default_sizes_decode_to_the_reference_pixels(B, O, tmp_path):
    """The shipped path's pixel contract at the reference's own defaults: `mandelbrot` (2000 x 2000, M = 128, main.cpp:20)
    and `pathtracer` (500 spp, 900 x 600, main.cpp:22-24; strict math, the default) write PNGs that decode to exactly the RGBA8 pixels
    the reference's host post-process makes of the oracle's storage buffer — through the pinned storage buffer and the stripe-parallel
    host conversion (round 5).  An odd width (51 x 34) runs the rotation's middle-column quirk through the same parallel loop, and
    --fast-png says what it no longer does."""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    bindir = os.path.join(os.path.dirname(os.path.dirname(B.LIB_PATH)), "bin")
    r = subprocess.run([os.path.join(bindir, "mandelbrot"), "--timing-json"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    _, lut_u8 = O.mandel_lut(128)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "mandelbrot.png").convert("RGBA")), lut_u8[O.mandelbrot_iters(2000, 2000, 128)])
    import json
    t = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"timing_ms"')][0])["timing_ms"]
    # (round 6: the storage-buffer route converts inside the PNG writer's stripe workers — `convert` is 0 and `png` contains it)
    assert t["kernel"] > 0 and t["copy"] > 0 and t["convert"] == 0 and t["png"] > 0 and t["total"] >= t["run"] + t["png"]
    r = subprocess.run([os.path.join(bindir, "pathtracer"), "--quiet"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    ref = O.pathtrace(900, 600, 500, math_mode=O.MATH_MC)
    exp = O.rotate180(O.float_to_rgba8(ref, 1.0).reshape(600, 900, 4), 900, 600)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "pathtracer.png").convert("RGBA")), exp)
    for threads in ("1", "3", "0"):
        r = subprocess.run([os.path.join(bindir, "pathtracer"), "3", "34", "--quiet", "--png-threads", threads, "--out", "odd.png", "--fast-png"],
                           capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0 and "--fast-png has no effect" in r.stdout
        ref = O.pathtrace(51, 34, 3, math_mode=O.MATH_MC)
        assert np.array_equal(np.asarray(Image.open(tmp_path / "odd.png").convert("RGBA")), O.rotate180(O.float_to_rgba8(ref, 1.0).reshape(34, 51, 4), 51, 34))


def test_pinned_and_pageable_callers_get_identical_bytes_at_the_pinned_rate(ctx, B):
    """the application owns a page-locked storage buffer (mc_host_alloc: what stands where the reference allocates its
    buffer HOST_VISIBLE | HOST_COHERENT, vulkanComputeApp.cpp:489-533).  A pageable caller gets the same bytes; the copy into the pinned
    buffer runs at >= 80 % of what a plain pinned hipMemcpy of the same size reaches (K1's 122.9 MB vec4 buffer; device time of the
    copy from mc_context_last_timing)."""
    import torch
    W, H = 3200, 2400
    p = B.mandelbrot_params(W, H, max_iter=100)
    with B.HostBuffer((H, W, 4)) as hb:
        best = 1e9
        for _ in range(4):
            hb.array[...] = 0
            ctx.mandelbrot(p, want_iters=False, out=hb.array)
            kernel_ms, copy_ms = ctx.last_timing()
            assert kernel_ms > 0 and copy_ms > 0
            best = min(best, copy_ms)
        pageable, _ = ctx.mandelbrot(p, want_iters=False)
        assert np.array_equal(pageable.view(np.uint32), hb.array.view(np.uint32))
        q = B.pathtrace_params(96, 64, 8)
        small = ctx.pathtrace(q)
        with B.HostBuffer((64, 96, 4)) as hs:
            assert np.array_equal(ctx.pathtrace(q, out=hs.array).view(np.uint32), small.view(np.uint32))
        nbytes = W * H * 16
        src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        dst = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        probe = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dst.copy_(src, non_blocking=True); e1.record(); torch.cuda.synchronize()
            if rep:
                probe = min(probe, e0.elapsed_time(e1))
        print(f"D2H of {nbytes / 1e6:.1f} MB into mc_host_alloc memory: {nbytes / best / 1e6:.1f} GB/s; pinned torch copy: {nbytes / probe / 1e6:.1f} GB/s")
        assert probe / best >= 0.8, (best, probe)
    with B.Context(0) as fresh:      # before the first blocking host-buffer call there is nothing to report
        with pytest.raises(B.McError):
            fresh.last_timing()


def test_warmup_leaves_the_blocking_calls_results_unchanged(B, O):
    """Round 6: mc_context_warmup_* — what the apps call from a helper thread in init() so that run()'s first launch
    does not pay for the code object, the tables and the device scratch — changes no result: on a FRESH context whose first call is the
    warm-up, the blocking calls return exactly what they return on a context that was never warmed: the fp32 storage buffer and the
    RGBA8 image, Mandelbrot (fp32 and two-float; the warm-up builds the REAL colour / c tables and runs one short tile) and path tracer
    (strict, fast, careful; odd width); argument errors."""
    L = B.lib()
    vp, u32 = C.c_void_p, C.c_uint32
    L.mc_context_warmup_pathtrace.argtypes = [vp, C.POINTER(B.PathtraceParams), vp, u32, vp, u32, C.c_int]
    L.mc_context_warmup_mandelbrot.argtypes = [vp, C.POINTER(B.MandelbrotParams), C.c_int]
    L.mc_pathtrace_render_rgba8.argtypes = [vp, C.POINTER(B.PathtraceParams), vp, u32, vp, u32, vp]
    L.mc_mandelbrot_render_rgba8.argtypes = [vp, C.POINTER(B.MandelbrotParams), vp]
    planes, spheres = B.default_scene()
    pl, sp = planes.ctypes.data_as(vp), spheres.ctypes.data_as(vp)

    def ptr(a):
        return a.ctypes.data_as(vp)

    with B.Context(0) as ref_ctx:
        for mode in (B.PT_MATH_STRICT, B.PT_MATH_FAST, B.PT_MATH_FAST_CAREFUL):
            for W, H in ((96, 64), (51, 30)):
                q = B.pathtrace_params(W, H, 12, math_mode=mode)
                want = ref_ctx.pathtrace(q)
                want8 = np.empty((H, W, 4), np.uint8)
                assert L.mc_pathtrace_render_rgba8(ref_ctx._h, C.byref(q), pl, 6, sp, 3, ptr(want8)) == 0
                for rgba8 in (0, 1):
                    with B.Context(0) as c:                       # fresh: the warm-up is this context's first launch
                        assert L.mc_context_warmup_pathtrace(c._h, C.byref(q), pl, 6, sp, 3, rgba8) == 0
                        if rgba8:
                            got = np.zeros((H, W, 4), np.uint8)
                            assert L.mc_pathtrace_render_rgba8(c._h, C.byref(q), pl, 6, sp, 3, ptr(got)) == 0
                            assert np.array_equal(got, want8)
                        else:
                            assert np.array_equal(c.pathtrace(q).view(np.uint32), want.view(np.uint32))
                        k, cp = c.last_timing()
                        assert k > 0 and cp > 0
        # a tile request warms up too (its scratch is the tile's); a request the render would refuse is refused by the warm-up
        q = B.pathtrace_params(64, 48, 6, row_begin=8, row_end=48, row_block=8, row_stride=16)
        want = ref_ctx.pathtrace(q)
        with B.Context(0) as c:
            assert L.mc_context_warmup_pathtrace(c._h, C.byref(q), pl, 6, sp, 3, 0) == 0
            assert np.array_equal(c.pathtrace(q).view(np.uint32), want.view(np.uint32))
            bad = B.pathtrace_params(64, 48, 6, row_begin=40, row_end=60)
            assert L.mc_context_warmup_pathtrace(c._h, C.byref(bad), pl, 6, sp, 3, 0) == 1
        for kw in (dict(max_iter=300), dict(max_iter=700, precision=B.PRECISION_DS, centre=(-0.7436438870371587, 0.13182590420531198),
                                            scale=(1e-6, 1e-6))):
            p = B.mandelbrot_params(200, 121, **kw)
            want, want_it = ref_ctx.mandelbrot(p)
            want8 = np.empty((121, 200, 4), np.uint8)
            assert L.mc_mandelbrot_render_rgba8(ref_ctx._h, C.byref(p), ptr(want8)) == 0
            for rgba8 in (0, 1):
                with B.Context(0) as c:
                    assert L.mc_context_warmup_mandelbrot(c._h, C.byref(p), rgba8) == 0
                    rg, it = c.mandelbrot(p)
                    assert np.array_equal(rg.view(np.uint32), want.view(np.uint32)) and np.array_equal(it, want_it)
                    got = np.zeros((121, 200, 4), np.uint8)
                    assert L.mc_mandelbrot_render_rgba8(c._h, C.byref(p), ptr(got)) == 0 and np.array_equal(got, want8)
        assert L.mc_context_warmup_mandelbrot(None, C.byref(p), 0) == 1 and L.mc_context_warmup_pathtrace(None, C.byref(q), pl, 6, sp, 3, 0) == 1


def test_apps_overlapped_start_writes_the_same_file_as_the_serial_start(B, tmp_path):
    """The apps hand the warm-up to a helper thread in init() and allocate the storage buffer inside run(), after the launch;
    --serial-start is the round-5 order (allocate in preRun(), first launch in run()).  Same file bytes either way, both routes, both apps; the timing line says what ran."""
    import json
    bindir = os.path.join(os.path.dirname(os.path.dirname(B.LIB_PATH)), "bin")
    for app, args in (("pathtracer", ["6", "48"]), ("pathtracer", ["6", "48", "--math", "fast"]), ("pathtracer", ["6", "48", "--math", "careful"]),
                      ("mandelbrot", ["--width", "320", "--height", "200", "--max-iter", "300"])):
        for route in ([], ["--gpu-postprocess"]):
            files = []
            for start in ([], ["--serial-start"]):
                out = tmp_path / f"{app}{len(files)}.png"
                r = subprocess.run([os.path.join(bindir, app)] + args + route + start + ["--quiet", "--timing-json", "--out", str(out)],
                                   capture_output=True, text=True, cwd=tmp_path)
                assert r.returncode == 0, r.stdout + r.stderr
                t = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"timing_ms"')][0])
                assert t["overlap_start"] is (not start) and t["timing_ms"]["kernel"] > 0
                if not start:
                    assert t["timing_ms"]["warmup"] > 0 and t["timing_ms"]["alloc"] > 0
                files.append(open(out, "rb").read())
            assert files[0] == files[1], (app, args, route)


def test_mandelbrot_app_streams_its_save_and_writes_the_same_file(B, O, tmp_path):
    """bin/mandelbrot renders its image in row bands and its PNG workers filter and deflate band k while the device renders band k + 1
    (ComputeApp::setStreamedSave, mc_mandelbrot_render_banded, pngwriter::Progressive) — by default where it pays (W x H x M >= 1e11: K4,
    not K1), always with --streamed-save; --no-streamed-save renders everything first, then encodes.  Same file bytes,
    both routes, fp32 and two-float, 1 .. 3 bands and a height that is no multiple of anything; the timing line says how many bands ran;
    the pixels are the oracle's."""
    import json
    app = os.path.join(os.path.dirname(os.path.dirname(B.LIB_PATH)), "bin", "mandelbrot")
    for size, extra in ((("700", "1283"), ["--max-iter", "300"]), (("640", "1900"), ["--max-iter", "200", "--precision", "ds"]), (("320", "200"), [])):
        args = ["--width", size[0], "--height", size[1]] + extra
        for route in ([], ["--gpu-postprocess"]):
            files = []
            for mode in (["--streamed-save"], ["--no-streamed-save"], []):   # (default: only where it pays, W x H x M >= 1e11 — not here)
                out = tmp_path / f"m{len(files)}.png"
                r = subprocess.run([app] + args + route + mode + ["--quiet", "--timing-json", "--out", str(out)], capture_output=True, text=True, cwd=tmp_path)
                assert r.returncode == 0, r.stdout + r.stderr
                t = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"timing_ms"')][0])["timing_ms"]
                assert t["streamed_bands"] == ((int(size[1]) + 639) // 640 if mode == ["--streamed-save"] else 0), t
                assert t["kernel"] > 0 and t["copy"] >= 0
                files.append(open(out, "rb").read())
            assert files[0] == files[1] == files[2], (args, route)
    # a request above the line streams by itself: 2000 x 1300 at M = 50 000 (1.3e11; the reference view: mostly early escapes, cheap)
    r = subprocess.run([app, "--width", "2000", "--height", "1300", "--max-iter", "50000", "--quiet", "--timing-json", "--out", str(tmp_path / "big.png")],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and json.loads([l for l in r.stdout.splitlines() if l.startswith('{"timing_ms"')][0])["timing_ms"]["streamed_bands"] == 3
    # the pixels of the last image (320 x 200, the reference's M = 128): the oracle's
    Image = pytest.importorskip("PIL.Image")
    _, lut_u8 = O.mandel_lut(128)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "m0.png").convert("RGBA")), lut_u8[O.mandelbrot_iters(320, 200, 128)])


def test_apps_leave_at_once_or_tear_down_on_request_same_file_same_exit_code(B, tmp_path):
    """Once the picture is on disk and everything is printed the apps leave with _Exit: destroying the context, unregistering the storage
    buffer and the HIP runtime's exit handlers cost 45 - 50 ms — a third of a K2 process — to return what the operating system reclaims
    anyway (profiles/r06_init_spread_probe.txt).  --full-teardown runs them (what leak checkers and sanitizers want).  Same file, same
    output, exit code 0 either way; the timing line carries CLOCK_MONOTONIC at main() and at the end, so that a parent can see what the
    app's own `total` cannot contain — and the parent's clock shows the teardown."""
    import json
    import time
    bindir = os.path.join(os.path.dirname(os.path.dirname(B.LIB_PATH)), "bin")
    # (K2's and K1's own sizes: a 6 spp x 48 row image read 76 / 87 ms after the file here, both modes alike — whatever is still in
    # flight at the end of so short a process, its exit waits for it; the timing is printed, not asserted: a test must not depend on
    # how busy the box is)
    for app, args in (("pathtracer", ["500", "600", "--math", "fast"]), ("mandelbrot", ["--width", "3200", "--height", "2400", "--max-iter", "1000"])):
        files, after = [], []
        for mode in ([], ["--full-teardown"]):
            out = tmp_path / f"{app}{len(files)}.png"
            t0 = time.monotonic()
            r = subprocess.run([os.path.join(bindir, app)] + args + mode + ["--quiet", "--timing-json", "--out", str(out)], capture_output=True, text=True, cwd=tmp_path)
            t1 = time.monotonic()
            assert r.returncode == 0, r.stdout + r.stderr
            j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"timing_ms"')][0])
            assert t0 * 1e3 <= j["main_at_ms"] <= j["end_at_ms"] <= t1 * 1e3
            assert abs((j["end_at_ms"] - j["main_at_ms"]) - j["timing_ms"]["total"]) < 2.0
            after.append(t1 * 1e3 - j["end_at_ms"])
            files.append(open(out, "rb").read())
        assert files[0] == files[1] and len(files[0]) > 100, app
        print(f"{app}: after the file was written: {after[0]:.1f} ms (default), {after[1]:.1f} ms (--full-teardown)")   # measured 1 / 42 - 77


def test_reference_png_mode_writes_the_reference_bytes(B, O, tmp_path):
    """VERDICT r5 item 3 / north_star "bit-identical PNG": route A with the reference's own codec.  `make REFERENCE=<checkout>`
    (what __graft_entry__.build() does where the checkout exists) compiles the reference's lodepng.cpp WHERE IT LIES into the apps
    behind --reference-png; this test runs wherever such a build is present (the GPU box receives the built binaries, not the
    checkout).  bin/mandelbrot --reference-png at 256 x 256, M = 128 must be the file whose SHA-256 was frozen from the reference
    codec over the oracle's pixels (tests/golden/mandelbrot_256_M128_lodepng.sha256, tests/golden/make_golden.py); at the reference's
    default 2000 x 2000 and for bin/pathtracer at ITS defaults (500 spp, 900 x 600, strict) the file must equal oracle/_ref's
    lodepng::encode of the oracle's pixels byte for byte (when oracle/_ref travelled too)."""
    import hashlib
    from conftest import GOLDEN
    bindir = os.path.join(os.path.dirname(os.path.dirname(B.LIB_PATH)), "bin")
    probe = subprocess.run([os.path.join(bindir, "mandelbrot"), "--reference-png", "--width", "8", "--height", "8", "--quiet"],
                           capture_output=True, text=True, cwd=tmp_path)
    if probe.returncode != 0 and "built without the reference's PNG codec" in probe.stdout:
        pytest.skip("the apps were built without REFERENCE=<checkout> (no reference checkout where they were built)")
    assert probe.returncode == 0, probe.stdout + probe.stderr
    r = subprocess.run([os.path.join(bindir, "mandelbrot"), "--reference-png", "--width", "256", "--height", "256", "--quiet"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    golden = open(os.path.join(GOLDEN, "mandelbrot_256_M128_lodepng.sha256")).read().split()[0]
    assert hashlib.sha256(open(tmp_path / "mandelbrot.png", "rb").read()).hexdigest() == golden
    # the same pixels through the apps' own writer: another (valid) file, the same image
    from PIL import Image
    r = subprocess.run([os.path.join(bindir, "mandelbrot"), "--width", "256", "--height", "256", "--quiet", "--out", "own.png"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0
    assert np.array_equal(np.asarray(Image.open(tmp_path / "own.png").convert("RGBA")), np.asarray(Image.open(tmp_path / "mandelbrot.png").convert("RGBA")))
    if O.ref_lodepng() is None:
        return
    _, lut_u8 = O.mandel_lut(128)
    want_m = bytes(O.ref_png_encode(np.ascontiguousarray(lut_u8[O.mandelbrot_iters(2000, 2000, 128)]), 2000, 2000))
    ref = O.pathtrace(900, 600, 500, math_mode=O.MATH_MC)
    want_p = bytes(O.ref_png_encode(np.ascontiguousarray(O.rotate180(O.float_to_rgba8(ref, 1.0).reshape(600, 900, 4), 900, 600)), 900, 600))
    for extra in ([], ["--gpu-postprocess"]):
        r = subprocess.run([os.path.join(bindir, "mandelbrot"), "--reference-png", "--quiet"] + extra, capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        assert open(tmp_path / "mandelbrot.png", "rb").read() == want_m
        r = subprocess.run([os.path.join(bindir, "pathtracer"), "--reference-png", "--quiet"] + extra, capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        assert open(tmp_path / "pathtracer.png", "rb").read() == want_p


def test_apps_fail_cleanly_on_an_image_no_memory_holds(B, tmp_path):
    """The failure path of the reference flow (main.cpp:35-38: message, EXIT_FAILURE) with the round-6 helpers in play: a storage
    buffer of 16 B x 6 000 000 x 4 000 000 = 384 TB can be neither page-locked (mc_host_alloc refuses it before touching a page:
    MC_ERR_OUT_OF_MEMORY) nor reserved on the device (the warm-up helper's request fails at the same time, on its own thread).
    preRun() throws while the helper may still be running; the object is torn down — derived destructor joins the helper before the
    scene tables go, the base abandons any encoder, the context is destroyed — and the process returns EXIT_FAILURE: no signal, no
    hang, no file."""
    bindir = os.path.join(os.path.dirname(os.path.dirname(B.LIB_PATH)), "bin")
    runs = [[os.path.join(bindir, "pathtracer"), "1", "4000000", "--quiet"],
            [os.path.join(bindir, "pathtracer"), "1", "4000000", "--quiet", "--gpu-postprocess", "--math", "fast"],
            [os.path.join(bindir, "mandelbrot"), "--width", "4000000", "--height", "4000000", "--quiet"],
            [os.path.join(bindir, "mandelbrot"), "--width", "4000000", "--height", "4000000", "--quiet", "--gpu-postprocess", "--serial-start"]]
    for cmd in runs:
        r = subprocess.run(cmd + ["--out", "never.png"], capture_output=True, text=True, cwd=tmp_path, timeout=120)
        assert r.returncode == 1, (cmd, r.returncode, r.stdout[-400:], r.stderr[-400:])      # EXIT_FAILURE — not a signal (negative)
        assert "mc_host_alloc" in r.stdout and "page-locked host memory" in r.stdout, r.stdout[-400:]
        assert not (tmp_path / "never.png").exists()
