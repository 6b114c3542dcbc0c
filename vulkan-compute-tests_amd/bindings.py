"""ctypes binding of lib/libmc_compute.so — the C ABI declared in include/mc_compute.h.

This is plumbing for tests/ and bench.py (Python drives torch.distributed and owns device tensors);
the product's host layer is the C++ code in host/.  There is NO CPU fallback: if the HIP library is
missing or no GPU is visible, the calls fail loudly.
"""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MC_LIB_PATH") or os.path.join(_HERE, "lib", "libmc_compute.so")   # MC_LIB_PATH: diagnostic builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mc_compute.h")

MC_OK = 0
ABI_VERSION = 3   # MC_ABI_VERSION of include/mc_compute.h this binding is written against
PRECISION_F32, PRECISION_DS = 0, 1
PT_MATH_STRICT, PT_MATH_FAST, PT_MATH_FAST_CAREFUL = 0, 1, 2
MANDEL_FMA = 1
MANDEL_ITERS_U16 = 2   # device form: d_iters is a uint16 plane (max_iter <= 65535): the multi-GPU exchange format
PT_GENERIC_KERNEL = 1
PT_NO_BOX_KERNEL = 4    # fast math: the general slab kernel instead of the closed-box ones
PT_NO_POOL_KERNEL = 8   # fast math: the round-synchronous closed-box kernel instead of the sample-pool kernel
PT_SCENE_IN_LDS = 16    # generic scenes: records staged into LDS by every block (automatic for small scenes) ...
PT_SCENE_IN_MEMORY = 32  # ... or read where they lie (automatic for large ones); same results
PT_NO_FAST_GUARD = 64    # measurements only: fast math even on a scene classified PT_SCENE_LIGHT_ENCLOSED
PT_PREC_F32, PT_PREC_FP64, PT_PREC_DS, PT_PREC_DF64 = 0, 1, 2, 3
DS_OPS = {"add": 0, "sub": 1, "mul": 2, "compare": 3, "sqrt": 4, "df64_add": 5, "df64_mult": 6, "df64_sqrt": 7, "twoprod": 8,
          "div": 9, "twodiff": 10, "df64_eqneq": 11, "mul_fma": 12}


def pt_precision(x):
    return int(x) << 16


def pt_force_s(s):
    return int(s) << 8


class McError(RuntimeError):
    def __init__(self, status, what, detail):
        super().__init__(f"{what}: {detail[0]} (status {status}){': ' + detail[1] if detail[1] else ''}")
        self.status = status


class MandelbrotParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_iter", C.c_uint32), ("precision", C.c_uint32),
                ("centre_x_hi", C.c_float), ("centre_x_lo", C.c_float), ("centre_y_hi", C.c_float),
                ("centre_y_lo", C.c_float), ("scale_x_hi", C.c_float), ("scale_x_lo", C.c_float),
                ("scale_y_hi", C.c_float), ("scale_y_lo", C.c_float), ("k_color", C.c_float * 4),
                ("row_begin", C.c_uint32), ("row_end", C.c_uint32), ("row_block", C.c_uint32),
                ("row_stride", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class PathtraceParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("sample_begin", C.c_uint32),
                ("sample_end", C.c_uint32), ("max_depth", C.c_uint32), ("row_begin", C.c_uint32),
                ("row_end", C.c_uint32), ("row_block", C.c_uint32), ("row_stride", C.c_uint32),
                ("math_mode", C.c_uint32), ("flags", C.c_uint32)]


def declared_symbols():
    """Every function name include/mc_compute.h declares (used by the export test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib():
    """Loads libmc_compute.so (raises if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int, C.c_float
        # The binding below is written against ABI_VERSION of include/mc_compute.h.  A diagnostic build loaded through MC_LIB_PATH may
        # be older (it then lacks the entry points added since, bound conditionally below); the shipped library must match.
        have = int(L.mc_abi_version())
        if have != ABI_VERSION and not os.environ.get("MC_LIB_PATH"):
            raise RuntimeError(f"{LIB_PATH} reports ABI version {have}, this binding is written against {ABI_VERSION}: rebuild "
                               f"(python -c 'import __graft_entry__ as g; g.build()')")
        L.mc_error_string.restype = C.c_char_p
        L.mc_error_string.argtypes = [i32]
        L.mc_last_error_detail.restype = C.c_char_p
        L.mc_device_count.argtypes = [C.POINTER(i32)]
        L.mc_context_create.argtypes = [i32, C.POINTER(vp)]
        L.mc_context_destroy.argtypes = [vp]
        L.mc_context_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(i32), C.POINTER(i32)]
        L.mc_context_synchronize.argtypes = [vp]
        L.mc_context_measure_clock.argtypes = [vp, C.POINTER(C.c_double)]
        L.mc_row_block.argtypes = []
        L.mc_row_block.restype = u32
        L.mc_tile_rows.argtypes = [u32, u32, u32, u32]
        L.mc_tile_rows.restype = u32
        L.mc_deinterleave_rows_device_async.argtypes = [vp, vp, u32, u32, u32, u32, u32, u32, vp, vp]
        L.mc_mandelbrot_assemble_device_async.argtypes = [vp, vp, vp, u32, u32, u32, u32, vp, vp, vp]
        L.mc_mandelbrot_default_params.argtypes = [u32, u32, C.POINTER(MandelbrotParams)]
        L.mc_mandelbrot_render.argtypes = [vp, C.POINTER(MandelbrotParams), vp, vp]
        L.mc_mandelbrot_render_device_async.argtypes = [vp, C.POINTER(MandelbrotParams), vp, vp, vp]
        L.mc_mandelbrot_colour_lut.argtypes = [u32, C.POINTER(f32), vp]
        L.mc_pathtrace_default_params.argtypes = [u32, u32, u32, C.POINTER(PathtraceParams)]
        L.mc_pathtrace_default_scene.argtypes = [C.POINTER(C.POINTER(f32)), C.POINTER(u32), C.POINTER(C.POINTER(f32)),
                                                 C.POINTER(u32)]
        L.mc_pathtrace_render.argtypes = [vp, C.POINTER(PathtraceParams), vp, u32, vp, u32, vp]
        L.mc_pathtrace_render_device_async.argtypes = [vp, C.POINTER(PathtraceParams), vp, u32, vp, u32, vp, vp]
        L.mc_convert_rgba8_device_async.argtypes = [vp, vp, u32, u32, f32, i32, vp, vp]
        L.mc_convert_rgba8.argtypes = [vp, vp, u32, u32, f32, i32, vp]
        if have >= 2 or hasattr(L, "mc_build_id"):   # (ABI 1 libraries older than round 5 lack these four)
            L.mc_build_id.restype = C.c_char_p
            L.mc_build_id.argtypes = []
            L.mc_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
            L.mc_host_free.argtypes = [vp]
            L.mc_context_last_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        if have >= 2:   # round 6
            L.mc_assemble_rgba8_device_async.argtypes = [vp, vp, u32, u32, u32, u32, u32, i32, vp, vp]
            L.mc_multi_pathtrace_render_rgba8.argtypes = [vp, C.POINTER(PathtraceParams), vp, u32, vp, u32, vp]
            L.mc_multi_mandelbrot_render_rgba8.argtypes = [vp, C.POINTER(MandelbrotParams), vp]
        L.mc_multi_create.argtypes = [i32, C.POINTER(vp)]
        L.mc_multi_destroy.argtypes = [vp]
        L.mc_multi_mandelbrot_render.argtypes = [vp, C.POINTER(MandelbrotParams), vp, vp]
        L.mc_multi_pathtrace_render.argtypes = [vp, C.POINTER(PathtraceParams), vp, u32, vp, u32, vp]
        _lib = L
    return _lib


_test_lib = None
TEST_LIB_PATH = os.path.join(_HERE, "lib", "libmc_compute_test.so")
TEST_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mc_compute_test.h")


def test_lib():
    """Loads libmc_compute_test.so — the device self-test hooks of the parity suite (include/mc_compute_test.h).  Test
    infrastructure: not part of the drop-in boundary, never linked by the apps."""
    global _test_lib
    if _test_lib is None:
        lib()   # the product library first (the hooks link against it; with MC_LIB_PATH the diagnostic build must be the one bound)
        if not os.path.exists(TEST_LIB_PATH):
            raise FileNotFoundError(f"{TEST_LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(TEST_LIB_PATH)
        vp, i32 = C.c_void_p, C.c_int
        L.mc_test_math.argtypes = [vp, i32, i32, vp, vp, C.c_size_t]
        L.mc_test_div3.argtypes = [vp, i32, vp, vp, vp, C.c_size_t]
        L.mc_test_math_sweep.argtypes = [vp, i32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint32)]
        L.mc_test_rand01.argtypes = [vp, vp, vp, C.c_size_t]
        L.mc_test_ds_op.argtypes = [vp, i32, vp, vp, vp, C.c_size_t]
        _test_lib = L
    return _test_lib


def declared_test_symbols():
    """Every function name include/mc_compute_test.h declares."""
    text = re.sub(r"/\*.*?\*/", "", open(TEST_HEADER_PATH).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(mc_test_[a-z0-9_]+)\s*\(", text)))


def _check(status, what):
    if status != MC_OK:
        L = lib()
        raise McError(status, what, (L.mc_error_string(status).decode(), L.mc_last_error_detail().decode()))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    _check(lib().mc_device_count(C.byref(n)), "mc_device_count")
    return n.value


def default_scene():
    pp, sp = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    np_, ns_ = C.c_uint32(0), C.c_uint32(0)
    _check(lib().mc_pathtrace_default_scene(C.byref(pp), C.byref(np_), C.byref(sp), C.byref(ns_)), "default_scene")
    planes = np.ctypeslib.as_array(pp, shape=(np_.value * 12,)).copy()
    spheres = np.ctypeslib.as_array(sp, shape=(ns_.value * 12,)).copy()
    return planes, spheres


def split_double(d):
    hi = np.float32(d)
    lo = np.float32(np.float64(d) - np.float64(hi))
    return float(hi), float(lo)


def mandelbrot_params(width, height, max_iter=128, precision=PRECISION_F32, centre=(-0.445, 0.0), scale=(2.34, 2.34),
                      k_color=(0.1, 0.7, 0.6, 0.0), row_begin=0, row_end=None, row_block=0, row_stride=0, flags=0):
    p = MandelbrotParams()
    _check(lib().mc_mandelbrot_default_params(width, height, C.byref(p)), "mc_mandelbrot_default_params")
    p.max_iter, p.precision, p.flags = max_iter, precision, flags
    p.centre_x_hi, p.centre_x_lo = split_double(centre[0])
    p.centre_y_hi, p.centre_y_lo = split_double(centre[1])
    p.scale_x_hi, p.scale_x_lo = split_double(scale[0])
    p.scale_y_hi, p.scale_y_lo = split_double(scale[1])
    for i in range(4):
        p.k_color[i] = k_color[i]
    p.row_begin, p.row_end = row_begin, height if row_end is None else row_end
    p.row_block, p.row_stride = row_block, row_stride
    return p


def pathtrace_params(width, height, spp, math_mode=PT_MATH_STRICT, sample_begin=0, sample_end=None, max_depth=12,
                     row_begin=0, row_end=None, row_block=0, row_stride=0, flags=0):
    p = PathtraceParams()
    _check(lib().mc_pathtrace_default_params(width, height, spp, C.byref(p)), "mc_pathtrace_default_params")
    p.math_mode, p.sample_begin, p.max_depth = math_mode, sample_begin, max_depth
    p.sample_end = spp if sample_end is None else sample_end
    p.row_begin, p.row_end = row_begin, height if row_end is None else row_end
    p.row_block, p.row_stride = row_block, row_stride
    p.flags = flags
    return p


def tile_rows(p):
    return int(lib().mc_tile_rows(p.row_begin, p.row_end, p.row_block, p.row_stride))


PT_SCENE_SLAB, PT_SCENE_LIGHTS_INSIDE, PT_SCENE_SPHERES_DISJOINT, PT_SCENE_LIGHT_ENCLOSED, PT_SCENE_MANY_SPHERES = 1, 2, 4, 8, 16
PT_SCENE_SPECULAR = 32
PT_KERNEL_GENERIC, PT_KERNEL_SLAB, PT_KERNEL_BOX, PT_KERNEL_POOL, PT_KERNEL_GENERIC_MEMORY = 0, 1, 3, 4, 5
PT_KERNEL_NAMES = {0: "generic", 1: "slab", 3: "box", 4: "pool", 5: "generic_memory"}


class PathtraceKernelInfo(C.Structure):
    _fields_ = [("kernel", C.c_uint32), ("lanes_per_pixel", C.c_uint32), ("math_mode", C.c_uint32), ("launches", C.c_uint32)]


def pathtrace_select_kernel(p, planes=None, spheres=None):
    """mc_pathtrace_select_kernel: the kernel a render call with these parameters and this scene runs (no device needed)."""
    if planes is None or spheres is None:
        planes, spheres = default_scene()
    planes = np.ascontiguousarray(planes, np.float32).reshape(-1, 12)
    spheres = np.ascontiguousarray(spheres, np.float32).reshape(-1, 12)
    out = PathtraceKernelInfo()
    fn = lib().mc_pathtrace_select_kernel
    fn.argtypes = [C.POINTER(PathtraceParams), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(PathtraceKernelInfo)]
    _check(fn(C.byref(p), _ptr(planes), planes.shape[0], _ptr(spheres), spheres.shape[0], C.byref(out)), "mc_pathtrace_select_kernel")
    return out


def build_id():
    """mc_build_id as a dict: {"pt": ..., "mandel": ..., "lib": ...} (source hashes of the loaded library's kernel families)."""
    return dict(kv.split("=") for kv in lib().mc_build_id().decode().split())


class HostBuffer:
    """Page-locked host memory from mc_host_alloc, exposed as a numpy array (the storage buffer an application owns)."""

    def __init__(self, shape, dtype=np.float32):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = C.c_void_p()
        _check(lib().mc_host_alloc(self.nbytes, C.byref(self._p)), "mc_host_alloc")
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self._p.value), dtype=dtype).reshape(shape)

    def free(self):
        if self._p:
            self.array = None
            _check(lib().mc_host_free(self._p), "mc_host_free")
            self._p = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()

    def __del__(self):   # (views handed out earlier dangle once the buffer is freed: keep the HostBuffer alive while `array` is in use)
        try:
            self.free()
        except Exception:
            pass


def _check_out(out, shape, what):
    """The caller's own output array goes to the library as a bare pointer: it must be exactly the buffer the call fills."""
    if not isinstance(out, np.ndarray) or out.dtype != np.float32 or tuple(out.shape) != tuple(shape):
        raise ValueError(f"{what}: out must be a float32 array of shape {tuple(shape)}, got "
                         f"{getattr(out, 'dtype', type(out))} {getattr(out, 'shape', '')}")
    if not out.flags.c_contiguous or not out.flags.writeable:
        raise ValueError(f"{what}: out must be C-contiguous and writeable")


def pathtrace_scene_class(planes, spheres):
    """mc_pathtrace_scene_class: which kernel specialisations the host would select for this scene (no device needed)."""
    planes = np.ascontiguousarray(planes, np.float32).reshape(-1, 12)
    spheres = np.ascontiguousarray(spheres, np.float32).reshape(-1, 12)
    out = C.c_uint32(0)
    fn = lib().mc_pathtrace_scene_class
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    _check(fn(_ptr(planes), planes.shape[0], _ptr(spheres), spheres.shape[0], C.byref(out)), "mc_pathtrace_scene_class")
    return int(out.value)


class Context:
    """mc_context wrapper (replaces VulkanComputeApp::init / cleanup)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().mc_context_create(device, C.byref(self._h)), "mc_context_create")
        self.device = device

    def close(self):
        if self._h:
            lib().mc_context_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, clk = C.c_int(0), C.c_int(0)
        _check(lib().mc_context_device_info(self._h, name, 256, C.byref(cu), C.byref(clk)), "mc_context_device_info")
        return name.value.decode(), cu.value, clk.value

    def measure_clock(self):
        """Shader clock (MHz) under full fp32 VALU load, measured in-kernel."""
        mhz = C.c_double(0.0)
        _check(lib().mc_context_measure_clock(self._h, C.byref(mhz)), "mc_context_measure_clock")
        return mhz.value

    def synchronize(self):
        _check(lib().mc_context_synchronize(self._h), "mc_context_synchronize")

    def last_timing(self):
        """(kernel_ms, copy_ms) of the last blocking host-buffer call on this context (mc_context_last_timing)."""
        k, c = C.c_double(0.0), C.c_double(0.0)
        _check(lib().mc_context_last_timing(self._h, C.byref(k), C.byref(c)), "mc_context_last_timing")
        return k.value, c.value

    # ---- host-buffer forms -------------------------------------------------------------------------
    def mandelbrot(self, p, want_rgba=True, want_iters=True, out=None):
        rows = tile_rows(p)
        if out is not None:
            if not want_rgba:
                raise ValueError("Context.mandelbrot: out= is the colour buffer; it needs want_rgba=True")
            _check_out(out, (rows, p.width, 4), "Context.mandelbrot")
        rgba = (out if out is not None else np.empty((rows, p.width, 4), np.float32)) if want_rgba else None
        iters = np.empty((rows, p.width), np.uint32) if want_iters else None
        _check(lib().mc_mandelbrot_render(self._h, C.byref(p), _ptr(rgba), _ptr(iters)), "mc_mandelbrot_render")
        return rgba, iters

    def mandelbrot_banded(self, p, band_rows, rgba8=False):
        """mc_mandelbrot_render_banded: the image (rows [row_begin, row_end)) rendered in pipelined row bands; returns the image — the fp32
        storage buffer, or RGBA8 converted on the device — and the rows_done values the callback heard, in the order it heard them."""
        rows = p.row_end - p.row_begin
        out = np.empty((rows, p.width, 4), np.uint8 if rgba8 else np.float32)
        heard = []
        cb_t = C.CFUNCTYPE(None, C.c_uint32, C.c_void_p)
        cb = cb_t(lambda done, user: heard.append(int(done)))
        fn = lib().mc_mandelbrot_render_banded
        fn.argtypes = [C.c_void_p, C.POINTER(MandelbrotParams), C.c_void_p, C.c_void_p, C.c_uint32, cb_t, C.c_void_p]
        _check(fn(self._h, C.byref(p), None if rgba8 else _ptr(out), _ptr(out) if rgba8 else None, band_rows, cb, None),
               "mc_mandelbrot_render_banded")
        return out, heard

    def pathtrace(self, p, planes=None, spheres=None, acc=None, out=None):
        if planes is None or spheres is None:
            planes, spheres = default_scene()
        planes = np.ascontiguousarray(planes, np.float32).reshape(-1)
        spheres = np.ascontiguousarray(spheres, np.float32).reshape(-1)
        rows = tile_rows(p)
        if out is not None:   # (out: the caller's own buffer, e.g. HostBuffer.array — written in place)
            _check_out(out, (rows, p.width, 4), "Context.pathtrace")
            if acc is not None:   # a progressive continuation starts from the accumulator: it goes into the caller's buffer first
                out[...] = np.asarray(acc, np.float32).reshape(out.shape)
        else:
            out = np.zeros((rows, p.width, 4), np.float32) if acc is None else np.ascontiguousarray(acc, np.float32).copy()
        _check(lib().mc_pathtrace_render(self._h, C.byref(p), _ptr(planes), planes.size // 12, _ptr(spheres),
                                         spheres.size // 12, _ptr(out)), "mc_pathtrace_render")
        return out

    def convert_rgba8(self, rgba_f32, scale, rotate180=False):
        a = np.ascontiguousarray(rgba_f32, np.float32)
        H, W = a.shape[0], a.shape[1]
        out = np.empty((H, W, 4), np.uint8)
        _check(lib().mc_convert_rgba8(self._h, _ptr(a), W, H, scale, int(rotate180), _ptr(out)), "mc_convert_rgba8")
        return out

    # ---- device-buffer forms (pointers are ints, e.g. torch.Tensor.data_ptr()) ------------------------
    def mandelbrot_device(self, p, d_rgba=0, d_iters=0, stream=0):
        _check(lib().mc_mandelbrot_render_device_async(self._h, C.byref(p), d_rgba or None, d_iters or None, stream or None),
               "mc_mandelbrot_render_device_async")

    def pathtrace_device(self, p, d_rgba, planes=None, spheres=None, stream=0):
        if planes is None or spheres is None:
            planes, spheres = default_scene()
        planes = np.ascontiguousarray(planes, np.float32).reshape(-1)
        spheres = np.ascontiguousarray(spheres, np.float32).reshape(-1)
        _check(lib().mc_pathtrace_render_device_async(self._h, C.byref(p), _ptr(planes), planes.size // 12, _ptr(spheres),
                                                      spheres.size // 12, d_rgba, stream or None),
               "mc_pathtrace_render_device_async")

    def convert_rgba8_device(self, d_rgba_f32, W, H, scale, rotate180, d_rgba8, stream=0):
        _check(lib().mc_convert_rgba8_device_async(self._h, d_rgba_f32, W, H, scale, int(rotate180), d_rgba8, stream or None),
               "mc_convert_rgba8_device_async")

    def mandelbrot_assemble_device(self, p, d_tiles, iters_bytes, n_tiles, row_block, tile_rows_padded, d_rgba=0, d_iters=0, stream=0):
        """Root side of the Mandelbrot exchange: gathered interleaved tiles of iteration counts (2 or 4 B/pixel) -> the whole
        image's vec4 storage buffer (lut[n]) and / or its uint32 count plane."""
        _check(lib().mc_mandelbrot_assemble_device_async(self._h, C.byref(p), d_tiles, iters_bytes, n_tiles, row_block,
                                                         tile_rows_padded, d_rgba or None, d_iters or None, stream or None),
               "mc_mandelbrot_assemble_device_async")

    def assemble_rgba8_device(self, d_tiles_u8, W, H, n_tiles, row_block, tile_rows_padded, rotate180, d_rgba8, stream=0):
        """Root side of the path tracer's RGBA8 exchange: gathered interleaved byte tiles -> the whole RGBA8 image, point-reflected
        as saveRenderedImage leaves it when rotate180 is set."""
        _check(lib().mc_assemble_rgba8_device_async(self._h, d_tiles_u8, W, H, n_tiles, row_block, tile_rows_padded, int(rotate180),
                                                    d_rgba8, stream or None), "mc_assemble_rgba8_device_async")

    def deinterleave_rows_device(self, d_tiles, W, H, n_tiles, row_block, tile_rows_padded, bpp, d_out, stream=0):
        _check(lib().mc_deinterleave_rows_device_async(self._h, d_tiles, W, H, n_tiles, row_block, tile_rows_padded, bpp,
                                                       d_out, stream or None), "mc_deinterleave_rows_device_async")

    # ---- device self-tests ---------------------------------------------------------------------------
    def test_math(self, fn, x, fast=False):
        code = {"sin": 0, "cos": 1, "log2": 2, "exp2": 3, "pow045": 4, "rsqrt": 5, "sqrt": 6, "rcp": 7, "sincos_s": 8,
                "sincos_c": 9}[fn]
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        out = np.empty_like(x)
        _check(test_lib().mc_test_math(self._h, code, int(fast), _ptr(x), _ptr(out), x.size), "mc_test_math")
        return out

    def test_div3(self, a, s, with_y=False):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 3)
        s = np.ascontiguousarray(s, np.float32).reshape(-1)
        out = np.empty_like(a)
        _check(test_lib().mc_test_div3(self._h, int(with_y), _ptr(a), _ptr(s), _ptr(out), s.size), "mc_test_div3")
        return out

    def test_math_sweep(self, fn, first_bits, count):
        """(mismatches vs the IEEE expansion, checksum, first mismatching pattern) over `count` consecutive bit patterns."""
        code = {"rsqrt": 5, "sqrt": 6, "rcp": 7}[fn]
        bad, chk, first = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        _check(test_lib().mc_test_math_sweep(self._h, code, first_bits, count, C.byref(bad), C.byref(chk), C.byref(first)),
               "mc_test_math_sweep")
        return bad.value, chk.value, first.value

    def test_rand01(self, xyz):
        k = np.ascontiguousarray(xyz, np.uint32).reshape(-1, 3)
        out = np.empty(k.shape, np.float32)
        _check(test_lib().mc_test_rand01(self._h, _ptr(k), _ptr(out), k.shape[0]), "mc_test_rand01")
        return out

    def test_ds_op(self, op, a, b):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(b, np.float32).reshape(-1, 2)
        out = np.empty(a.shape, np.float32)
        _check(test_lib().mc_test_ds_op(self._h, DS_OPS[op], _ptr(a), _ptr(b), _ptr(out),
                                   a.shape[0]), "mc_test_ds_op")
        return out


class Multi:
    """mc_multi wrapper: single-process multi-GPU render with an RCCL gather to device 0."""

    def __init__(self, n_devices):
        self._h = C.c_void_p()
        _check(lib().mc_multi_create(n_devices, C.byref(self._h)), "mc_multi_create")
        self.n = n_devices

    def close(self):
        if self._h:
            lib().mc_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def mandelbrot(self, p, want_rgba=True, want_iters=True):
        rgba = np.empty((p.height, p.width, 4), np.float32) if want_rgba else None
        iters = np.empty((p.height, p.width), np.uint32) if want_iters else None
        _check(lib().mc_multi_mandelbrot_render(self._h, C.byref(p), _ptr(rgba), _ptr(iters)), "mc_multi_mandelbrot_render")
        return rgba, iters

    def pathtrace(self, p, planes=None, spheres=None):
        if planes is None or spheres is None:
            planes, spheres = default_scene()
        planes = np.ascontiguousarray(planes, np.float32).reshape(-1)
        spheres = np.ascontiguousarray(spheres, np.float32).reshape(-1)
        out = np.empty((p.height, p.width, 4), np.float32)
        _check(lib().mc_multi_pathtrace_render(self._h, C.byref(p), _ptr(planes), planes.size // 12, _ptr(spheres),
                                               spheres.size // 12, _ptr(out)), "mc_multi_pathtrace_render")
        return out

    def pathtrace_rgba8(self, p, planes=None, spheres=None):
        """mc_multi_pathtrace_render_rgba8: the finished image as saveRenderedImage converts and rotates it, 4 B/pixel exchanged."""
        if planes is None or spheres is None:
            planes, spheres = default_scene()
        planes = np.ascontiguousarray(planes, np.float32).reshape(-1)
        spheres = np.ascontiguousarray(spheres, np.float32).reshape(-1)
        out = np.empty((p.height, p.width, 4), np.uint8)
        _check(lib().mc_multi_pathtrace_render_rgba8(self._h, C.byref(p), _ptr(planes), planes.size // 12, _ptr(spheres),
                                                     spheres.size // 12, _ptr(out)), "mc_multi_pathtrace_render_rgba8")
        return out

    def mandelbrot_rgba8(self, p):
        out = np.empty((p.height, p.width, 4), np.uint8)
        _check(lib().mc_multi_mandelbrot_render_rgba8(self._h, C.byref(p), _ptr(out)), "mc_multi_mandelbrot_render_rgba8")
        return out
