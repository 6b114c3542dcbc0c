"""vulkan-compute-tests_amd — MI355X-native implementation of the two compute hot paths of
pjhusky/vulkan-compute-tests (Mandelbrot escape-time, smallpt-style path tracer).

Layout:  csrc/  hand-written HIP kernels for gfx950 + the C ABI (include/mc_compute.h)
         host/  C++ mirror of the reference's app surface (MandelbrotApp / PathtracerApp / main)
         bindings.py  ctypes plumbing over the C ABI for tests/ and bench.py

The directory name contains a hyphen, so import it through `__graft_entry__.load_package()`.
"""
from . import bindings, sharding  # noqa: F401
