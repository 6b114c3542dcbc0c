"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liboracle.so (CPU restatement of the
reference's shaders) and, when present, oracle/_ref/liblodepng_ref.so (the reference's own vendored
lodepng, compiled from /root/reference by oracle/Makefile).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

# Default scene: src/pathtracerApp.h:14-39 (TEST_PRECISION_WITH_LARGE_SPHERE_WALLS == 0).  DATA, 12 floats/object.
DEFAULT_PLANES = np.array([
    -1.0, 0.0, 0.0, 2.6, 0, 0, 0, 0, .85, .25, .25, 1,   # Left
    +1.0, 0.0, 0.0, 2.6, 0, 0, 0, 0, .25, .35, .85, 1,   # Right
    0.0, +1.0, 0.0, 2.0, 0, 0, 0, 0, .75, .75, .75, 1,   # Top
    0.0, -1.0, 0.0, 2.0, 0, 0, 0, 0, .75, .75, .75, 1,   # Bottom
    0.0, 0.0, -1.0, 2.8, 0, 0, 0, 0, .85, .85, .25, 1,   # Back
    0.0, 0.0, +1.0, 7.9, 0, 0, 0, 0, 0.1, 0.7, 0.7, 1,   # Front
], dtype=np.float64).astype(np.float32)
DEFAULT_SPHERES = np.array([
    -1.3, -1.2, -1.3, 0.8, 0, 0, 0, 0, .999, .999, .999, 2,   # mirror
    1.3, -1.2, -0.2, 0.8, 0, 0, 0, 0, .999, .999, .999, 3,    # glass
    0, 2 * 0.8, 0, 0.2, 100, 100, 100, 0, 0, 0, 0, 1,         # light
], dtype=np.float64).astype(np.float32)

# Reference Mandelbrot view (shaders/mandelbrot.comp:38) and push constant (src/mandelbrotApp.h:139)
REF_VIEW = np.array([-0.445, 0.0, 0.0, 0.0, 2.34, 0.0, 2.34, 0.0], dtype=np.float32)
REF_KCOLOR = np.array([0.1, 0.7, 0.6, 0.0], dtype=np.float32)

MATH_LIBM, MATH_MC = 0, 1
PREC_F32, PREC_FP64, PREC_DS, PREC_DF64 = 0, 1, 2, 3   # sphere-test branch (emulateDouble.h.glsl:13-26)

# Scene of TEST_PRECISION_WITH_LARGE_SPHERE_WALLS != 0 (src/pathtracerApp.h:22-23,28-38): one dummy plane, six
# radius-1e5 wall spheres (the classic smallpt walls), then the mirror / glass / light spheres.  DATA.
LARGE_SPHERE_PLANES = np.array([1, 0, 0, 1000, 0, 0, 0, 0, 1, 1, 1, 1], dtype=np.float64).astype(np.float32)
LARGE_SPHERE_SPHERES = np.array([
    1e5 - 2.6, 0, 0, 1e5, 0, 0, 0, 0, .85, .25, .25, 1,     # Left
    1e5 + 2.6, 0, 0, 1e5, 0, 0, 0, 0, .25, .35, .85, 1,     # Right
    0, 1e5 + 2, 0, 1e5, 0, 0, 0, 0, .75, .75, .75, 1,       # Top
    0, -1e5 - 2, 0, 1e5, 0, 0, 0, 0, .75, .75, .75, 1,      # Bottom
    0, 0, -1e5 - 2.8, 1e5, 0, 0, 0, 0, .85, .85, .25, 1,    # Back
    0, 0, 1e5 + 7.9, 1e5, 0, 0, 0, 0, 0.1, 0.7, 0.7, 1,     # Front
    -1.3, -1.2, -1.3, 0.8, 0, 0, 0, 0, .999, .999, .999, 2,
    1.3, -1.2, -0.2, 0.8, 0, 0, 0, 0, .999, .999, .999, 3,
    0, 2 * 0.8, 0, 0.2, 100, 100, 100, 0, 0, 0, 0, 1,
], dtype=np.float64).astype(np.float32)

_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("MC_ORACLE_LIB_PATH") or os.path.join(_HERE, "liboracle.so")   # (sanitizer build: oracle/_san/)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_mandelbrot_iters.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, _f32p, C.c_int, C.c_uint32,
                                              C.c_uint32, _u32p, C.c_int]
        L.oracle_mandelbrot_iters.restype = C.c_int
        L.oracle_mandel_lut.argtypes = [C.c_uint32, _f32p, _f32p, _u8p]
        L.oracle_mandel_pixel_iters.argtypes = [_u32p, C.c_uint64, C.c_uint32]
        L.oracle_mandel_pixel_iters.restype = C.c_uint64
        L.oracle_float_to_rgba8.argtypes = [C.c_uint64, C.c_float, _f32p, _u8p]
        L.oracle_rotate180_rgba8.argtypes = [C.c_uint32, C.c_uint32, _u8p]
        L.oracle_pathtrace.argtypes = [C.c_uint32] * 6 + [_f32p, C.c_uint32, _f32p, C.c_uint32, C.c_int, C.c_uint32,
                                                          C.c_uint32, _f32p, C.c_int, C.c_void_p]
        L.oracle_pathtrace.restype = C.c_int
        L.oracle_pathtrace_sample.argtypes = [C.c_uint32] * 6 + [_f32p, C.c_uint32, _f32p, C.c_uint32, C.c_int, _f32p]
        L.oracle_rand01.argtypes = [C.c_uint64, _u32p, _f32p]
        L.oracle_ds_op.argtypes = [C.c_int, C.c_uint64, _f32p, _f32p, _f32p]
        L.oracle_mc_math.argtypes = [C.c_int, C.c_uint64, _f32p, _f32p]
        L.oracle_hardware_threads.restype = C.c_int
        _LIB = L
    return _LIB


def split_double(d):
    """hi = (float)d ; lo = (float)(d - (double)hi) — how the host feeds the ds variant."""
    hi = np.float32(d)
    lo = np.float32(np.float64(d) - np.float64(hi))
    return hi, lo


def make_view(cx, cy, sx, sy):
    v = np.zeros(8, np.float32)
    v[0], v[1] = split_double(cx)
    v[2], v[3] = split_double(cy)
    v[4], v[5] = split_double(sx)
    v[6], v[7] = split_double(sy)
    return v


def mandelbrot_iters(W, H, max_iter, view=REF_VIEW, precision=0, row_begin=0, row_end=None, nthreads=0):
    nthreads = nthreads or hardware_threads()
    row_end = H if row_end is None else row_end
    out = np.empty((row_end - row_begin, W), np.uint32)
    rc = lib().oracle_mandelbrot_iters(W, H, max_iter, np.ascontiguousarray(view, np.float32), precision, row_begin,
                                       row_end, out, nthreads)
    if rc:
        raise ValueError("oracle_mandelbrot_iters: bad arguments")
    return out


def mandel_lut(max_iter, kcolor=REF_KCOLOR):
    f = np.empty((max_iter + 1, 4), np.float32)
    u = np.empty((max_iter + 1, 4), np.uint8)
    lib().oracle_mandel_lut(max_iter, np.ascontiguousarray(kcolor, np.float32), f, u)
    return f, u


def mandel_pixel_iters(iters, max_iter):
    it = np.ascontiguousarray(iters, np.uint32).reshape(-1)
    return int(lib().oracle_mandel_pixel_iters(it, it.size, max_iter))


def float_to_rgba8(buf, scale):
    b = np.ascontiguousarray(buf, np.float32).reshape(-1, 4)
    out = np.empty((b.shape[0], 4), np.uint8)
    lib().oracle_float_to_rgba8(b.shape[0], scale, b, out)
    return out


def rotate180(rgba8, W, H):
    a = np.ascontiguousarray(rgba8, np.uint8).reshape(H, W, 4).copy()
    lib().oracle_rotate180_rgba8(W, H, a)
    return a


def pathtrace(W, H, spp, planes=DEFAULT_PLANES, spheres=DEFAULT_SPHERES, math_mode=MATH_LIBM, max_depth=12,
              sample_begin=0, sample_end=None, row_begin=0, row_end=None, acc=None, nthreads=0, counts=False,
              precision=PREC_F32):
    math_mode = math_mode | (precision << 8)
    nthreads = nthreads or hardware_threads()
    sample_end = spp if sample_end is None else sample_end
    row_end = H if row_end is None else row_end
    out = np.zeros((row_end - row_begin, W, 4), np.float32) if acc is None else np.ascontiguousarray(acc, np.float32).copy()
    planes = np.ascontiguousarray(planes, np.float32)
    spheres = np.ascontiguousarray(spheres, np.float32)
    cnt = (C.c_uint64 * 12)() if counts else None
    rc = lib().oracle_pathtrace(W, H, spp, sample_begin, sample_end, max_depth, planes, planes.size // 12, spheres,
                                spheres.size // 12, math_mode, row_begin, row_end, out.reshape(-1), nthreads,
                                C.cast(cnt, C.c_void_p) if counts else None)
    if rc:
        raise ValueError("oracle_pathtrace: bad arguments")
    if counts:
        names = ["add", "mul", "div", "sqrt", "trig", "pow", "cmp", "iop", "cvt", "intersect_calls", "bounces", "samples"]
        return out, dict(zip(names, [int(x) for x in cnt]))
    return out


def pathtrace_sample(gx, gy, W, H, samp, planes=DEFAULT_PLANES, spheres=DEFAULT_SPHERES, math_mode=MATH_LIBM,
                     max_depth=12):
    rgb = np.zeros(3, np.float32)
    planes = np.ascontiguousarray(planes, np.float32)
    spheres = np.ascontiguousarray(spheres, np.float32)
    lib().oracle_pathtrace_sample(gx, gy, W, H, samp, max_depth, planes, planes.size // 12, spheres, spheres.size // 12,
                                  math_mode, rgb)
    return rgb


def rand01(xyz):
    k = np.ascontiguousarray(xyz, np.uint32).reshape(-1, 3)
    out = np.empty(k.shape, np.float32)
    lib().oracle_rand01(k.shape[0], k.reshape(-1), out.reshape(-1))
    return out


DS_OPS = {"add": 0, "sub": 1, "mul": 2, "compare": 3, "sqrt": 4, "df64_add": 5, "df64_mult": 6, "df64_sqrt": 7,
          "twoprod": 8, "div": 9, "twodiff": 10, "df64_eqneq": 11}


def ds_op(op, a, b):
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(b, np.float32).reshape(-1, 2)
    out = np.empty(a.shape, np.float32)
    lib().oracle_ds_op(DS_OPS[op], a.shape[0], a.reshape(-1), b.reshape(-1), out.reshape(-1))
    return out


def mc_math(fn, x):
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    out = np.empty_like(x)
    lib().oracle_mc_math({"sin": 0, "cos": 1, "log2": 2, "exp2": 3, "pow045": 4}[fn], x.size, x, out)
    return out


def _cgroup_cpu_quota():
    """CPUs granted by the cgroup (v2 cpu.max or v1 cfs quota), or None when unlimited / unreadable."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = float(f.read())
        if quota > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def hardware_threads():
    """Worker threads the oracle should use: the CPUs this process may actually run on — the smaller of the
    hardware/affinity count and the cgroup CPU quota (a GPU box shows 256 CPUs but grants 16: 256 threads there only
    add throttling and context switches, measured 1.06e7 vs 1.54e7 samples/s)."""
    n = int(lib().oracle_hardware_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    q = _cgroup_cpu_quota()
    if q:
        n = min(n, max(1, int(q + 0.5)))
    return max(1, n)


# ---- oracle/_ref: the reference's own lodepng -------------------------------------------------------
def ref_lodepng():
    """Returns the ctypes handle of the reference's lodepng build, or None when it is unavailable."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "liblodepng_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.lodepng_encode32.argtypes = [C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t), _u8p, C.c_uint, C.c_uint]
        R.lodepng_encode32.restype = C.c_uint
        R.lodepng_decode32.argtypes = [C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                       C.c_char_p, C.c_size_t]
        R.lodepng_decode32.restype = C.c_uint
        _REF = R
    return _REF


def ref_png_encode(rgba8, W, H):
    """lodepng::encode(filename, image, w, h) of the reference (mandelbrotApp.h:181, pathtracerApp.h:245) → bytes."""
    R = ref_lodepng()
    if R is None:
        raise RuntimeError("oracle/_ref/liblodepng_ref.so not built (needs /root/reference)")
    img = np.ascontiguousarray(rgba8, np.uint8).reshape(-1)
    assert img.size == W * H * 4
    out = C.POINTER(C.c_ubyte)()
    n = C.c_size_t(0)
    err = R.lodepng_encode32(C.byref(out), C.byref(n), img, W, H)
    if err:
        raise RuntimeError(f"lodepng_encode32 error {err}")
    data = C.string_at(out, n.value)
    C.CDLL(None).free(out)
    return data


def ref_png_decode(png_bytes):
    R = ref_lodepng()
    if R is None:
        raise RuntimeError("oracle/_ref/liblodepng_ref.so not built (needs /root/reference)")
    out = C.POINTER(C.c_ubyte)()
    w, h = C.c_uint(0), C.c_uint(0)
    err = R.lodepng_decode32(C.byref(out), C.byref(w), C.byref(h), png_bytes, len(png_bytes))
    if err:
        raise RuntimeError(f"lodepng_decode32 error {err}")
    arr = np.ctypeslib.as_array(out, shape=(h.value * w.value * 4,)).copy().reshape(h.value, w.value, 4)
    C.CDLL(None).free(out)
    return arr
