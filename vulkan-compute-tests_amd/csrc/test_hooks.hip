// Device self-test hooks of the parity suite (include/mc_compute_test.h) — libmc_compute_test.so, NOT part of the product
// library: evaluates the device functions of mc_math.h / ds_arith.h / the path tracer RNG over arrays so that tests/ can compare
// them with the oracle bit for bit.  Links against libmc_compute.so (context, error detail, DeviceBuffer).
#include <algorithm>
#include <cstring>

#include "ds_arith.h"
#include "mc_internal.h"
#include "mc_math.h"
#include "../../include/mc_compute_test.h"

namespace mc {

// ---- device self-test kernels -----------------------------------------------------------------------
__global__ void test_math_kernel(int fn, int fast, const float* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = in[i], r = 0.0f;
    if (!fast) {
        switch (fn) {
            case 0: r = dm::mc_sin(x); break;
            case 1: r = dm::mc_cos(x); break;
            case 2: r = dm::mc_log2(x); break;
            case 3: r = dm::mc_exp2(x); break;
            case 4: r = dm::fpow<false>(x, 0.45f); break;
            case 5: r = dm::inversesqrt<false>(x); break;
            case 6: r = dm::fsqrt<false>(x); break;
            case 7: r = dm::fdiv<false>(1.0f, x); break;
            case 8: { float s, c; dm::mc_sincos(x, s, c); r = s; } break;
            case 9: { float s, c; dm::mc_sincos(x, s, c); r = c; } break;
            default: break;
        }
    } else {
        const float two_pi = 2.0f * 3.141592653589793f;
        switch (fn) {
            case 0: case 8: { float s, c; dm::sincos_angle<true>(x, x / two_pi, s, c); r = s; } break;
            case 1: case 9: { float s, c; dm::sincos_angle<true>(x, x / two_pi, s, c); r = c; } break;
            case 2: r = __builtin_amdgcn_logf(x); break;
            case 3: r = __builtin_amdgcn_exp2f(x); break;
            case 4: r = dm::fpow<true>(x, 0.45f); break;
            case 5: r = dm::inversesqrt<true>(x); break;
            case 6: r = dm::fsqrt<true>(x); break;
            case 7: r = dm::fdiv<true>(1.0f, x); break;
            default: break;
        }
    }
    out[i] = r;
}

__global__ void test_rand01_kernel(const uint32_t* __restrict__ xyz, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    for (int k = 0; k < 3; k++) {   // pathTracer.comp:107-110
        uint32_t nx = ((x >> 8) ^ y) * 1103515245u, ny = ((y >> 8) ^ z) * 1103515245u, nz = ((z >> 8) ^ x) * 1103515245u;
        x = nx; y = ny; z = nz;
    }
    const float s = 2.3283064365386963e-10f;
    out[3 * i] = (float)x * s; out[3 * i + 1] = (float)y * s; out[3 * i + 2] = (float)z * s;
}

__global__ void test_ds_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                               size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ds2 x{a[2 * i], a[2 * i + 1]}, y{b[2 * i], b[2 * i + 1]}, r{0.0f, 0.0f};
    auto rsq = [](float v) { return dm::inversesqrt<false>(v); };
    switch (op) {
        case 0: r = ds_add(x, y); break;
        case 1: r = ds_sub(x, y); break;
        case 2: r = ds_mul(x, y); break;
        case 4: r = ds_sqrt(x, rsq); break;
        case 5: r = df64_add(x, y); break;
        case 6: r = df64_mult(x, y); break;
        case 7: r = df64_sqrt(x, rsq); break;
        case 8: r = ds_twoProd(x.hi, y.hi); break;
        case 9: r = ds_div(x, y); break;
        case 10: r = twoDiff(x.hi, y.hi); break;
        case 11: r = ds2{df64_eq(x, y) ? 1.0f : 0.0f, df64_neq(x, y) ? 1.0f : 0.0f}; break;
        case 12: r = ds_mul_fma(x, y); break;
        default: r = ds2{ds_compare(x, y), 0.0f}; break;
    }
    out[2 * i] = r.hi; out[2 * i + 1] = r.lo;
}

}  // namespace mc

using namespace mc;

extern "C" {

// ---- device self-tests ------------------------------------------------------------------------------
static int run_test(mc_context* ctx, const void* in_a, size_t bytes_a, const void* in_b, size_t bytes_b, void* out,
                    size_t bytes_out, void (*launch)(void*, void*, void*, size_t, hipStream_t, int, int), size_t n, int p0,
                    int p1) {
    MC_HIP_TRY(hipSetDevice(ctx->device));
    DeviceBuffer da, db, dout;   // released on every exit path
    struct Release {
        DeviceBuffer &a, &b, &c;
        ~Release() { a.release(); b.release(); c.release(); }
    } release{da, db, dout};
    int rc;
    if ((rc = da.reserve(bytes_a))) return rc;
    if (bytes_b && (rc = db.reserve(bytes_b))) return rc;
    if ((rc = dout.reserve(bytes_out))) return rc;
    MC_HIP_TRY(hipMemcpy(da.ptr, in_a, bytes_a, hipMemcpyHostToDevice));
    if (bytes_b) MC_HIP_TRY(hipMemcpy(db.ptr, in_b, bytes_b, hipMemcpyHostToDevice));
    launch(da.ptr, db.ptr, dout.ptr, n, ctx->stream, p0, p1);
    MC_HIP_TRY(hipGetLastError());
    MC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    MC_HIP_TRY(hipMemcpy(out, dout.ptr, bytes_out, hipMemcpyDeviceToHost));
    return MC_OK;
}

int mc_test_math(mc_context* ctx, int fn, int fast, const float* in, float* out, size_t n) {
    if (!ctx || !in || !out || !n) return MC_ERR_INVALID_ARGUMENT;
    return run_test(ctx, in, n * 4, nullptr, 0, out, n * 4,
                    [](void* a, void*, void* o, size_t n_, hipStream_t s, int fn_, int fast_) {
                        hipLaunchKernelGGL(test_math_kernel, dim3((unsigned)((n_ + 255) / 256)), dim3(256), 0, s, fn_, fast_,
                                           (const float*)a, (float*)o, n_);
                    },
                    n, fn, fast);
}

// strict (a0, a1, a2) / s through dm::div3 (short division inside its window, IEEE expansion outside); with_y: y = RN(1/s) supplied
__global__ void test_div3_kernel(int with_y, const float* __restrict__ a, const float* __restrict__ sv, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a0 = a[3 * i], a1 = a[3 * i + 1], a2 = a[3 * i + 2];
    const float s = sv[i];
    if (with_y) dm::div3<false, true>(a0, a1, a2, s, dm::ieee_div(1.0f, s));
    else dm::div3<false, false>(a0, a1, a2, s, 0.0f);
    out[3 * i] = a0; out[3 * i + 1] = a1; out[3 * i + 2] = a2;
}

int mc_test_div3(mc_context* ctx, int with_y, const float* a, const float* s, float* out, size_t n) {
    if (!ctx || !a || !s || !out || !n) return MC_ERR_INVALID_ARGUMENT;
    return run_test(ctx, a, n * 12, s, n * 4, out, n * 12,
                    [](void* x, void* y, void* o, size_t n_, hipStream_t st, int with_y_, int) {
                        hipLaunchKernelGGL(test_div3_kernel, dim3((unsigned)((n_ + 255) / 256)), dim3(256), 0, st, with_y_,
                                           (const float*)x, (const float*)y, (float*)o, n_);
                    },
                    n, with_y, 0);
}

// Every bit pattern first_bits .. first_bits + count - 1: strict device function vs the compiler's IEEE expansion.
__global__ void test_math_sweep_kernel(int fn, uint32_t first_bits, unsigned long long count, unsigned long long* res) {
    unsigned long long bad = 0, sum = 0;
    uint32_t first_bad = 0xffffffffu;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t u = first_bits + (uint32_t)i;
        const float x = dm::as_float(u);
        float got, ref;
        switch (fn) {
            case 5: got = dm::inversesqrt<false>(x); ref = dm::ieee_div(1.0f, dm::ieee_sqrt(x)); break;
            case 6: got = dm::fsqrt<false>(x); ref = dm::ieee_sqrt(x); break;
            default: got = dm::in_short_window(x) ? dm::rcp_short(x) : dm::ieee_div(1.0f, x); ref = dm::ieee_div(1.0f, x); break;
        }
        const uint32_t gb = dm::as_uint(got), rb = dm::as_uint(ref);
        if (gb != rb && !(got != got && ref != ref)) { bad++; first_bad = first_bad < u ? first_bad : u; }
        sum += (unsigned long long)(gb ^ (u * 0x9E3779B1u));
    }
    atomicAdd(&res[0], bad);
    atomicAdd(&res[1], sum);
    atomicMin(&res[2], (unsigned long long)first_bad);
}

int mc_test_math_sweep(mc_context* ctx, int fn, uint32_t first_bits, uint64_t count, uint64_t* mismatches, uint64_t* checksum,
                       uint32_t* first_mismatch) {
    if (!ctx || (fn != 5 && fn != 6 && fn != 7) || !count || count > (1ull << 32) || !mismatches) return MC_ERR_INVALID_ARGUMENT;
    MC_HIP_TRY(hipSetDevice(ctx->device));
    DeviceBuffer res;
    struct Release { DeviceBuffer& b; ~Release() { b.release(); } } release{res};
    if (int rc = res.reserve(3 * sizeof(unsigned long long))) return rc;
    const unsigned long long init[3] = {0ull, 0ull, ~0ull};
    MC_HIP_TRY(hipMemcpy(res.ptr, init, sizeof init, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(test_math_sweep_kernel, dim3(256 * 16), dim3(256), 0, ctx->stream, fn, first_bits,
                       (unsigned long long)count, (unsigned long long*)res.ptr);
    MC_HIP_TRY(hipGetLastError());
    unsigned long long out[3];
    MC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    MC_HIP_TRY(hipMemcpy(out, res.ptr, sizeof out, hipMemcpyDeviceToHost));
    *mismatches = out[0];
    if (checksum) *checksum = out[1];
    if (first_mismatch) *first_mismatch = (uint32_t)out[2];
    return MC_OK;
}

int mc_test_rand01(mc_context* ctx, const uint32_t* xyz, float* out, size_t n) {
    if (!ctx || !xyz || !out || !n) return MC_ERR_INVALID_ARGUMENT;
    return run_test(ctx, xyz, n * 12, nullptr, 0, out, n * 12,
                    [](void* a, void*, void* o, size_t n_, hipStream_t s, int, int) {
                        hipLaunchKernelGGL(test_rand01_kernel, dim3((unsigned)((n_ + 255) / 256)), dim3(256), 0, s,
                                           (const uint32_t*)a, (float*)o, n_);
                    },
                    n, 0, 0);
}

int mc_test_ds_op(mc_context* ctx, int op, const float* a, const float* b, float* out, size_t n) {
    if (!ctx || !a || !b || !out || !n) return MC_ERR_INVALID_ARGUMENT;
    return run_test(ctx, a, n * 8, b, n * 8, out, n * 8,
                    [](void* x, void* y, void* o, size_t n_, hipStream_t s, int op_, int) {
                        hipLaunchKernelGGL(test_ds_kernel, dim3((unsigned)((n_ + 255) / 256)), dim3(256), 0, s, op_,
                                           (const float*)x, (const float*)y, (float*)o, n_);
                    },
                    n, op, 0);
}

}  // extern "C"
