// pk_probe.hip -- issue-rate probe: scalar v_fma_f32 vs v_pk_fma_f32 vs IEEE f32 division on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/pk_probe.hip -o tools/pk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f;
    float x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    if (MODE == 0) {
        for (int i = 0; i < iters; i++) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        }
    } else if (MODE == 1) {
        f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
        f2 av = {a, a}, bv = {b, b};
        for (int i = 0; i < iters; i++) {
            p0 = __builtin_elementwise_fma(p0, av, bv); p1 = __builtin_elementwise_fma(p1, av, bv);
            p2 = __builtin_elementwise_fma(p2, av, bv); p3 = __builtin_elementwise_fma(p3, av, bv);
        }
        x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
    } else if (MODE == 2) {
        for (int i = 0; i < iters; i++) {
            x0 = x0 / a + b; x1 = x1 / a + b; x2 = x2 / a + b; x3 = x3 / a + b;
            x4 = x4 / a + b; x5 = x5 / a + b; x6 = x6 / a + b; x7 = x7 / a + b;
        }
    } else if (MODE == 3) {
        f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
        f2 av = {a, a}, bv = {b, b};
        for (int i = 0; i < iters; i++) {
            p0 = p0 * av + bv; p1 = p1 * av + bv; p2 = p2 * av + bv; p3 = p3 * av + bv;
        }
        x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
    } else {
        for (int i = 0; i < iters; i++) {
            x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
            x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
static void run(const char* name, float* d, int opsPerIter) {
    const int iters = 4096, blocks = 256 * 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 0.999f, 0.001f, iters);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, 0.999f, 0.001f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double lane_ops = (double)blocks * 256 * iters * opsPerIter;
    printf("%-28s %.3f ms  %.2f T lane-ops/s\n", name, ms, lane_ops / ms * 1e-9);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 16 * 256 * 4);
    run<0>("fma scalar (8 fma)", d, 8);
    run<1>("pk_fma (4 pk = 8 fma)", d, 8);
    run<4>("mul+add scalar (8+8)", d, 16);
    run<3>("pk mul+add (4+4 pk)", d, 16);
    run<2>("div+add scalar (8 div)", d, 8);
    return 0;
}
