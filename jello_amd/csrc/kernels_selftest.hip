// kernels_selftest.hip -- diagnostics: evaluates the dmath.h scalar routines on the device so that
// tests can compare them bit-for-bit with the CPU checker (tests/test_gpu_math.py).  Not part of
// the render path.
#include "kcommon.h"

using namespace jd;

__global__ void k_selftest_math(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
    switch (op) {
        case 0: r = sin_(x); break;
        case 1: r = cos_(x); break;
        case 2: r = atan2_(x, y); break;
        case 3: r = acos_(x); break;
        case 4: r = asin_(x); break;
        case 5: r = pow23_abs_(x); break;
        case 6: r = x / y; break;
        case 7: r = sqrt_(x); break;
        case 8: r = round_(x); break;
        case 9: r = u2f(to_u32(x)); break;
        case 10: r = u2f((uint32_t)to_i32(x)); break;
        case 11: r = u2f((uint32_t)f32_to_f16(x)); break;
        case 12: r = x * y + x; break;      // must NOT be contracted to an FMA
        case 13: r = floor_(x * y + 0.5f); break;
        case 14: r = fmin_(x, y); break;
        case 15: r = fmax_(x, y); break;
        case 16: r = clamp_(x, 0.0f, 1.0f); break;
        case 17: r = clamp_(x * y, 0.0f, 1.0f); break;
        default: break;
    }
    out[i] = r;
}

extern "C" int jh_selftest_math_launch(hipStream_t stream, int op, const float* a, const float* b, float* out, uint32_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_selftest_math, dim3((n + 255) / 256), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
