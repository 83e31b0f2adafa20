// Issue-rate microbenchmark for gfx950 (run on the GPU box): how many cycles one SIMD needs per wave64 instruction of a
// given kind when 1..8 waves share it.  Each wave runs REPS trips of a loop of 64 instructions of the kind under test on
// 8 independent registers (no dependency closer than 8 instructions).  Prints cycles per instruction per SIMD, taking the
// clock from s_memrealtime-free timing: wall time * (reported clock) -- so the absolute numbers carry the clock's
// uncertainty, the RATIOS between kinds do not.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu.hip -o tools/ubench/valu && tools/ubench/valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REPS 4096

#define BODY8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define BODY64(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I) BODY8(I)

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, float seed, uint64_t* clk, uint32_t iseed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b0 = 1, b1 = 2, b2 = 3, b3 = 4, b4 = 5, b5 = 6, b6 = 7, b7 = 8;
    const float m = 1.0000001f, c = 1e-9f;
    uint32_t s0 = iseed, s1 = 3;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REPS; r++) {
        if constexpr (KIND == 0) {  // v_fma_f32
#define I(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##n) : "v"(m), "v"(c));
            BODY64(I)
#undef I
        } else if constexpr (KIND == 1) {  // v_add_f32
#define I(n) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a##n) : "v"(c));
            BODY64(I)
#undef I
        } else if constexpr (KIND == 2) {  // v_pk_add_f32 (two registers each)
            asm volatile(
                ".rept 16\n"
                "v_pk_add_f32 %[p0], %[p0], %[q]\n v_pk_add_f32 %[p1], %[p1], %[q]\n v_pk_add_f32 %[p2], %[p2], %[q]\n v_pk_add_f32 %[p3], %[p3], %[q]\n"
                ".endr\n"
                : [p0] "+v"(*(double*)&a0), [p1] "+v"(*(double*)&a2), [p2] "+v"(*(double*)&a4), [p3] "+v"(*(double*)&a6)
                : [q] "v"(*(double*)&b0));
        } else if constexpr (KIND == 3) {  // v_max_f32 / v_min_f32 (clamp style)
#define I(n) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a##n) : "v"(c));
            BODY64(I)
#undef I
        } else if constexpr (KIND == 4) {  // s_add_u32 only
            asm volatile(".rept 64\n s_add_u32 %0, %0, %1\n .endr\n" : "+s"(s0) : "s"(s1) : "scc");
        } else if constexpr (KIND == 5) {  // VALU : SALU 1 : 1 interleaved (64 + 64)
#define I(n) asm volatile("v_add_f32_e32 %0, %2, %0\n s_add_u32 %1, %1, %3" : "+v"(a##n), "+s"(s0) : "v"(c), "s"(s1) : "scc");
            BODY64(I)
#undef I
        } else if constexpr (KIND == 6) {  // v_readlane_b32 (VALU -> SGPR)
#define I(n) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(a##n));
            BODY64(I)
#undef I
        } else if constexpr (KIND == 7) {  // v_rcp_f32 (transcendental)
#define I(n) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a##n));
            BODY64(I)
#undef I
        } else if constexpr (KIND == 8) {  // DPP add (row_shr:1)
#define I(n) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##n));
            BODY64(I)
#undef I
        } else if constexpr (KIND == 9) {  // v_cmp + v_cndmask pairs (32 + 32)
#define I(n) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a##n) : "v"(c) : "vcc");
            BODY8(I) BODY8(I) BODY8(I) BODY8(I)
#undef I
        } else if constexpr (KIND == 10) {  // ds_bpermute_b32 (LDS crossbar), 64 in flight then wait
#define I(n) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a##n) : "v"(b##n));
            BODY64(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (KIND == 11) {  // v_mul_f32 with an SGPR operand
#define I(n) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a##n) : "s"(m));
            BODY64(I)
#undef I
        } else if constexpr (KIND == 12) {  // v_cvt_i32_f32 + v_cvt_f32_i32
#define I(n) asm volatile("v_cvt_i32_f32_e32 %0, %0\n v_cvt_f32_i32_e32 %0, %0" : "+v"(a##n));
            BODY8(I) BODY8(I) BODY8(I) BODY8(I)
#undef I
        } else if constexpr (KIND == 13) {  // dependent v_add_f32 chain on ONE register
            asm volatile(".rept 64\n v_add_f32_e32 %0, %1, %0\n .endr\n" : "+v"(a0) : "v"(c));
        } else if constexpr (KIND == 14) {  // s_and_b64 / s_bcnt style 64-bit scalar
            uint64_t q = s0, q2 = ~0ull;
            asm volatile(".rept 32\n s_and_b64 %0, %0, %1\n s_bcnt1_i32_b64 %2, %0\n .endr\n" : "+s"(q), "+s"(q2), "+s"(s1) : : "scc");
            s0 += (uint32_t)q;
        } else if constexpr (KIND == 15) {  // v_mbcnt + v_bcnt (integer VALU)
#define I(n) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a##n) : "v"(b##n));
            BODY64(I)
#undef I
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    float sum = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)s0 + b0;
    if (sum == 123.456f) out[0] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

typedef void (*KF)(float*, float, uint64_t*, uint32_t);
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, mhz);
    float* out; hipMalloc(&out, 4); uint64_t* clk; hipMalloc(&clk, 8);
    const char* names[] = {"v_fma_f32", "v_add_f32", "v_pk_add_f32", "v_max_f32", "s_add_u32", "v_add+s_add pairs", "v_readlane_b32", "v_rcp_f32",
                           "v_add_u32 dpp", "v_cmp+v_cndmask", "ds_bpermute_b32", "v_mul_f32 sgpr", "v_cvt i32<->f32", "v_add_f32 dependent", "s_and_b64+s_bcnt1", "v_mbcnt_lo"};
    KF fs[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>};
    const int insts[] = {64, 64, 64, 64, 64, 128, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-22s", "waves/SIMD:");
    const int occ[] = {1, 2, 3, 4, 6, 8};
    for (int o : occ) printf("%8d", o);
    printf("   (cycles per instruction per SIMD at the nominal clock)\n");
    for (int kind = 0; kind < 16; kind++) {
        printf("%-22s", names[kind]);
        for (int o : occ) {
            const int blocks = cus * 4 * o;  // one-wave workgroups: the dispatcher spreads them evenly over SIMDs
            fs[kind]<<<blocks, 64>>>(out, 1.0f, clk, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            fs[kind]<<<blocks, 64>>>(out, 1.0f, clk, 1u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double cycles = ms * 1e-3 * mhz * 1e6;
            printf("%8.2f", cycles / ((double)REPS * insts[kind] * o));
        }
        printf("\n");
    }
    return 0;
}
