// f64 / division / LDS issue rates for gfx950 (companion of valu3.hip; run on the GPU box).  One asm block of 64 instances per trip over 4
// independent register pairs (the f32 kinds: over ONE register, i.e. a dependent chain -- read them at 4+ waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REPS 1024
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, float seed, uint32_t iseed) {
    double p0 = seed, p1 = seed + 1, p2 = seed + 2, p3 = seed + 3, q = seed * 5;
    float f = seed * 3.0f + (float)threadIdx.x, f2 = seed * 0.5f;
    double sq = (double)__builtin_bit_cast(float, iseed);
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(f, f, f, f);
    uint32_t addr = threadIdx.x * 16u, addr4 = threadIdx.x * 4u;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f rv4 = {0, 0, 0, 0};
    float rs = (float)threadIdx.x * 4.0f;
    double rd = 0.0;
    unsigned long long m64 = 1ull << threadIdx.x;
    for (int r = 0; r < REPS; r++) {
        if constexpr (KIND == 0) asm volatile(".rept 16\n"
"v_fma_f64 %[p0], %[p0], %[q], %[q]\n"
"v_fma_f64 %[p1], %[p1], %[q], %[q]\n"
"v_fma_f64 %[p2], %[p2], %[q], %[q]\n"
"v_fma_f64 %[p3], %[p3], %[q], %[q]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 1) asm volatile(".rept 16\n"
"v_add_f64 %[p0], %[p0], %[q]\n"
"v_add_f64 %[p1], %[p1], %[q]\n"
"v_add_f64 %[p2], %[p2], %[q]\n"
"v_add_f64 %[p3], %[p3], %[q]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 2) asm volatile(".rept 16\n"
"v_mul_f64 %[p0], %[p0], %[q]\n"
"v_mul_f64 %[p1], %[p1], %[q]\n"
"v_mul_f64 %[p2], %[p2], %[q]\n"
"v_mul_f64 %[p3], %[p3], %[q]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 3) asm volatile(".rept 16\n"
"v_fma_f64 %[p0], %[p0], %[sq], %[q]\n"
"v_fma_f64 %[p1], %[p1], %[sq], %[q]\n"
"v_fma_f64 %[p2], %[p2], %[sq], %[q]\n"
"v_fma_f64 %[p3], %[p3], %[sq], %[q]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 4) asm volatile(".rept 16\n"
"v_rcp_f64_e32 %[p0], %[p0]\n"
"v_rcp_f64_e32 %[p1], %[p1]\n"
"v_rcp_f64_e32 %[p2], %[p2]\n"
"v_rcp_f64_e32 %[p3], %[p3]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 5) asm volatile(".rept 16\n"
"v_rsq_f64_e32 %[p0], %[p0]\n"
"v_rsq_f64_e32 %[p1], %[p1]\n"
"v_rsq_f64_e32 %[p2], %[p2]\n"
"v_rsq_f64_e32 %[p3], %[p3]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 6) asm volatile(".rept 16\n"
"v_sqrt_f64_e32 %[p0], %[p0]\n"
"v_sqrt_f64_e32 %[p1], %[p1]\n"
"v_sqrt_f64_e32 %[p2], %[p2]\n"
"v_sqrt_f64_e32 %[p3], %[p3]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 7) asm volatile(".rept 16\n"
"v_cvt_f64_f32_e32 %[p0], %[f]\n"
"v_cvt_f64_f32_e32 %[p1], %[f]\n"
"v_cvt_f64_f32_e32 %[p2], %[f]\n"
"v_cvt_f64_f32_e32 %[p3], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 8) asm volatile(".rept 16\n"
"v_cvt_f32_f64_e32 %[f], %[p0]\n"
"v_cvt_f32_f64_e32 %[f], %[p1]\n"
"v_cvt_f32_f64_e32 %[f], %[p2]\n"
"v_cvt_f32_f64_e32 %[f], %[p3]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 9) asm volatile(".rept 16\n"
"v_cmp_lt_f64_e32 vcc, %[p0], %[q]\n"
"v_cmp_lt_f64_e32 vcc, %[p1], %[q]\n"
"v_cmp_lt_f64_e32 vcc, %[p2], %[q]\n"
"v_cmp_lt_f64_e32 vcc, %[p3], %[q]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 10) asm volatile(".rept 16\n"
"v_max_f64 %[p0], %[p0], %[q]\n"
"v_max_f64 %[p1], %[p1], %[q]\n"
"v_max_f64 %[p2], %[p2], %[q]\n"
"v_max_f64 %[p3], %[p3], %[q]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 11) asm volatile(".rept 16\n"
"v_ldexp_f64 %[p0], %[p0], 1\n"
"v_ldexp_f64 %[p1], %[p1], 1\n"
"v_ldexp_f64 %[p2], %[p2], 1\n"
"v_ldexp_f64 %[p3], %[p3], 1\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 12) asm volatile(".rept 16\n"
"v_fract_f64_e32 %[p0], %[p0]\n"
"v_fract_f64_e32 %[p1], %[p1]\n"
"v_fract_f64_e32 %[p2], %[p2]\n"
"v_fract_f64_e32 %[p3], %[p3]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 13) asm volatile(".rept 16\n"
"v_rndne_f64_e32 %[p0], %[p0]\n"
"v_rndne_f64_e32 %[p1], %[p1]\n"
"v_rndne_f64_e32 %[p2], %[p2]\n"
"v_rndne_f64_e32 %[p3], %[p3]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 14) asm volatile(".rept 16\n"
"v_cvt_i32_f64_e32 %[f], %[p0]\n"
"v_cvt_i32_f64_e32 %[f], %[p1]\n"
"v_cvt_i32_f64_e32 %[f], %[p2]\n"
"v_cvt_i32_f64_e32 %[f], %[p3]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 15) asm volatile(".rept 16\n"
"v_cvt_f64_i32_e32 %[p0], %[f]\n"
"v_cvt_f64_i32_e32 %[p1], %[f]\n"
"v_cvt_f64_i32_e32 %[p2], %[f]\n"
"v_cvt_f64_i32_e32 %[p3], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 16) asm volatile(".rept 16\n"
"v_div_scale_f64 %[p0], vcc, %[p0], %[q], %[q]\n"
"v_div_scale_f64 %[p1], vcc, %[p1], %[q], %[q]\n"
"v_div_scale_f64 %[p2], vcc, %[p2], %[q], %[q]\n"
"v_div_scale_f64 %[p3], vcc, %[p3], %[q], %[q]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 17) asm volatile(".rept 16\n"
"v_div_fmas_f64 %[p0], %[p0], %[q], %[q]\n"
"v_div_fmas_f64 %[p1], %[p1], %[q], %[q]\n"
"v_div_fmas_f64 %[p2], %[p2], %[q], %[q]\n"
"v_div_fmas_f64 %[p3], %[p3], %[q], %[q]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 18) asm volatile(".rept 16\n"
"v_div_fixup_f64 %[p0], %[p0], %[q], %[q]\n"
"v_div_fixup_f64 %[p1], %[p1], %[q], %[q]\n"
"v_div_fixup_f64 %[p2], %[p2], %[q], %[q]\n"
"v_div_fixup_f64 %[p3], %[p3], %[q], %[q]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 19) asm volatile(".rept 16\n"
"v_mov_b64_e32 %[p0], %[q]\n"
"v_mov_b64_e32 %[p1], %[q]\n"
"v_mov_b64_e32 %[p2], %[q]\n"
"v_mov_b64_e32 %[p3], %[q]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 20) asm volatile(".rept 16\n"
"v_cndmask_b32_e64 %[f], %[f], %[f], s[20:21]\n"
"v_cndmask_b32_e64 %[f], %[f], %[f], s[20:21]\n"
"v_cndmask_b32_e64 %[f], %[f], %[f], s[20:21]\n"
"v_cndmask_b32_e64 %[f], %[f], %[f], s[20:21]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 21) asm volatile(".rept 16\n"
"v_div_scale_f32 %[f], vcc, %[f], %[f], 1.0\n"
"v_div_scale_f32 %[f], vcc, %[f], %[f], 1.0\n"
"v_div_scale_f32 %[f], vcc, %[f], %[f], 1.0\n"
"v_div_scale_f32 %[f], vcc, %[f], %[f], 1.0\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 22) asm volatile(".rept 16\n"
"v_div_fmas_f32 %[f], %[f], %[f], %[f]\n"
"v_div_fmas_f32 %[f], %[f], %[f], %[f]\n"
"v_div_fmas_f32 %[f], %[f], %[f], %[f]\n"
"v_div_fmas_f32 %[f], %[f], %[f], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 23) asm volatile(".rept 16\n"
"v_div_fixup_f32 %[f], %[f], %[f], %[f]\n"
"v_div_fixup_f32 %[f], %[f], %[f], %[f]\n"
"v_div_fixup_f32 %[f], %[f], %[f], %[f]\n"
"v_div_fixup_f32 %[f], %[f], %[f], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 24) asm volatile(".rept 16\n"
"v_fmac_f32_e32 %[f], %[f2], %[f2]\n"
"v_fmac_f32_e32 %[f], %[f2], %[f2]\n"
"v_fmac_f32_e32 %[f], %[f2], %[f2]\n"
"v_fmac_f32_e32 %[f], %[f2], %[f2]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 25) asm volatile(".rept 16\n"
"v_fma_f32 %[f], %[f2], %[f2], %[f]\n"
"v_fma_f32 %[f], %[f2], %[f2], %[f]\n"
"v_fma_f32 %[f], %[f2], %[f2], %[f]\n"
"v_fma_f32 %[f], %[f2], %[f2], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 26) asm volatile(".rept 16\n"
"v_exp_f32_e32 %[f], %[f]\n"
"v_exp_f32_e32 %[f], %[f]\n"
"v_exp_f32_e32 %[f], %[f]\n"
"v_exp_f32_e32 %[f], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 27) asm volatile(".rept 16\n"
"v_log_f32_e32 %[f], %[f]\n"
"v_log_f32_e32 %[f], %[f]\n"
"v_log_f32_e32 %[f], %[f]\n"
"v_log_f32_e32 %[f], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 28) asm volatile(".rept 16\n"
"v_sin_f32_e32 %[f], %[f]\n"
"v_sin_f32_e32 %[f], %[f]\n"
"v_sin_f32_e32 %[f], %[f]\n"
"v_sin_f32_e32 %[f], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 29) asm volatile(".rept 16\n"
"v_frexp_mant_f32_e32 %[f], %[f]\n"
"v_frexp_mant_f32_e32 %[f], %[f]\n"
"v_frexp_mant_f32_e32 %[f], %[f]\n"
"v_frexp_mant_f32_e32 %[f], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 30) asm volatile(".rept 16\n"
"v_ldexp_f32 %[f], %[f], 1\n"
"v_ldexp_f32 %[f], %[f], 1\n"
"v_ldexp_f32 %[f], %[f], 1\n"
"v_ldexp_f32 %[f], %[f], 1\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 31) asm volatile(".rept 16\n"
"v_rndne_f32_e32 %[f], %[f]\n"
"v_rndne_f32_e32 %[f], %[f]\n"
"v_rndne_f32_e32 %[f], %[f]\n"
"v_rndne_f32_e32 %[f], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 32) asm volatile(".rept 16\n"
"v_trunc_f32_e32 %[f], %[f]\n"
"v_trunc_f32_e32 %[f], %[f]\n"
"v_trunc_f32_e32 %[f], %[f]\n"
"v_trunc_f32_e32 %[f], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 33) asm volatile(".rept 16\n"
"v_cvt_f32_i32_e32 %[f], %[f]\n"
"v_cvt_f32_i32_e32 %[f], %[f]\n"
"v_cvt_f32_i32_e32 %[f], %[f]\n"
"v_cvt_f32_i32_e32 %[f], %[f]\n"
".endr\n" : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [f] "+v"(f) : [q] "v"(q), [sq] "s"(sq), [f2] "v"(f2) : "vcc", "s20", "s21");
        if constexpr (KIND == 34) { asm volatile(".rept 64\n" "ds_read_b32 %[r0], %[a4]\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
        if constexpr (KIND == 35) { asm volatile(".rept 64\n" "ds_read_b64 %[r01], %[a8]\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
        if constexpr (KIND == 36) { asm volatile(".rept 64\n" "ds_read_b128 %[r], %[a]\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
        if constexpr (KIND == 37) { asm volatile(".rept 64\n" "ds_read_b128 %[r], %[z]\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
        if constexpr (KIND == 38) { asm volatile(".rept 64\n" "ds_write_b32 %[a4], %[r0]\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
        if constexpr (KIND == 39) { asm volatile(".rept 64\n" "ds_write_b128 %[a], %[r]\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
        if constexpr (KIND == 40) { asm volatile(".rept 64\n" "ds_write_b8 %[a4], %[r0]\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
        if constexpr (KIND == 41) { asm volatile(".rept 64\n" "ds_read_u8 %[r0], %[a4]\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
        if constexpr (KIND == 42) { asm volatile(".rept 64\n" "ds_or_b64 %[a8r], %[m]\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
        if constexpr (KIND == 43) { asm volatile(".rept 64\n" "ds_bpermute_b32 %[r0], %[a4], %[r0]\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
        if constexpr (KIND == 44) { asm volatile(".rept 64\n" "ds_swizzle_b32 %[r0], %[r0] offset:0x041F\n" ".endr\n s_waitcnt lgkmcnt(0)\n" : [r] "+v"(rv4), [r0] "+v"(rs), [r01] "+v"(rd) : [a] "v"(addr), [a4] "v"(addr4), [a8] "v"(addr4 * 2u), [a8r] "v"((threadIdx.x & 15u) * 8u), [z] "v"(0u), [m] "v"(m64) : "memory"); }
    }
    float sum = f + rv4.x + rv4.w + rs + (float)rd + (float)(p0 + p1 + p2 + p3) + lds[(threadIdx.x + 1) & 63].x;
    if (sum == 123.456f) out[0] = sum;
}
typedef void (*KF)(float*, float, uint32_t);
int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    float* out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_fma_f64 sgpr", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_cvt_f64_f32 (lo)", "v_cvt_f32_f64", "v_cmp_lt_f64", "v_max_f64", "v_ldexp_f64", "v_fract_f64", "v_rndne_f64", "v_cvt_i32_f64", "v_cvt_f64_i32", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_mov_b64", "v_cndmask x2 (f64 select)", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_fmac_f32", "v_fma_f32 (2 srcs same)", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_frexp_mant_f32", "v_ldexp_f32", "v_rndne_f32", "v_trunc_f32", "v_cvt_f32_i32", "ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_read_b128 broadcast (one address)", "ds_write_b32", "ds_write_b128", "ds_write_b8", "ds_read_u8", "ds_or_b64 (16 addresses)", "ds_bpermute_b32", "ds_swizzle_b32"};
    KF fs[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>, k<17>, k<18>, k<19>, k<20>, k<21>, k<22>, k<23>, k<24>, k<25>, k<26>, k<27>, k<28>, k<29>, k<30>, k<31>, k<32>, k<33>, k<34>, k<35>, k<36>, k<37>, k<38>, k<39>, k<40>, k<41>, k<42>, k<43>, k<44>};
    const int nk = sizeof(fs) / sizeof(fs[0]);
    printf("%d CUs, nominal %.0f MHz; cycles per instruction per SIMD\n", cus, mhz);
    printf("%-40s%8s%8s%8s%8s\n", "waves/SIMD:", "1", "2", "4", "6");
    const int occ[] = {1, 2, 4, 6};
    for (int kind = 0; kind < nk; kind++) {
        printf("%-40s", names[kind]);
        for (int o : occ) {
            const int blocks = cus * 4 * o;
            fs[kind]<<<blocks, 64>>>(out, 1.0f, 0x3f800001u);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            fs[kind]<<<blocks, 64>>>(out, 1.0f, 0x3f800001u);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%8.2f", ms * 1e-3 * mhz * 1e6 / ((double)REPS * 64 * o));
        }
        printf("\n");
    }
    return 0;
}
