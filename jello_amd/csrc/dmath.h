// dmath.h -- device arithmetic rules shared by every jello_amd kernel (gfx950).
//
// The kernels restate the reference WGSL (engine/wgpu_engine/shaders/original/*.wgsl).  WGSL
// leaves the precision of transcendentals implementation-defined; to make results reproducible
// bit-for-bit (run to run, and against the CPU checker) every kernel uses:
//   * IEEE binary32 + - * / sqrt with no contraction (build flag -ffp-contract=off, correctly
//     rounded divide/sqrt are hipcc's default);
//   * min/max = IEEE minNum/maxNum (v_min_f32/v_max_f32), clamp = min(max(x,lo),hi), WGSL-spec sign/mix/fract;
//     round() = ties-to-even; saturating u32()/i32();
//   * sin/cos/atan2/acos/asin/|x|^(2/3) evaluated in binary64 by the fixed operation sequences
//     below (Cody-Waite + Taylor, table-split atan, Halley cbrt) and rounded once to binary32 --
//     the same policy as the reference's Go twin, which rounds float64 libm (jmath/jmath.go:48-87).
//     FP64 vector rate on MI355X is ample; flatten is not the bandwidth-bound stage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define JD __device__ __forceinline__

namespace jd {

JD uint32_t f2u(float f) { return __float_as_uint(f); }
JD float u2f(uint32_t u) { return __uint_as_float(u); }

// min/max = IEEE-754 minNum/maxNum as implemented by v_min_f32/v_max_f32: a NaN operand yields the other
// operand, and -0 orders below +0.  (WGSL leaves NaN handling of min/max implementation-defined.)
JD float fmin_(float a, float b) { return __builtin_fminf(a, b); }
JD float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
JD float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
JD int32_t imin_(int32_t a, int32_t b) { return (b < a) ? b : a; }
JD int32_t imax_(int32_t a, int32_t b) { return (a < b) ? b : a; }
JD int32_t iclamp_(int32_t x, int32_t lo, int32_t hi) { return imin_(imax_(x, lo), hi); }
JD uint32_t umin_(uint32_t a, uint32_t b) { return (b < a) ? b : a; }
JD uint32_t umax_(uint32_t a, uint32_t b) { return (a < b) ? b : a; }
JD float sign_(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
JD float abs_(float x) { return u2f(f2u(x) & 0x7fffffffu); }
JD float floor_(float x) { return floorf(x); }
JD float ceil_(float x) { return ceilf(x); }
JD float round_(float x) { return rintf(x); }
JD float sqrt_(float x) { return sqrtf(x); }
JD float fract_(float x) { return x - floorf(x); }
JD float mix_(float a, float b, float t) { return a * (1.0f - t) + b * t; }

// WGSL's saturating conversions (NaN -> 0).  On the device ONE instruction each: v_cvt_u32_f32 / v_cvt_i32_f32 truncate, clamp out-of-range
// magnitudes to 0 / 0xffffffff resp. to +-2^31 (min_int for the negative side) and turn NaN into 0 (CDNA ISA, "V_CVT_U32_F32"); written as
// assembly because the C++ cast is undefined out of range, so the compiler is free to assume the range; tests/test_gpu_math.py holds
// both to the table of edge cases.  (Round 6: five instructions each as compare + select chains before; line_setup of path_count /
// path_tiling and the multisampled fine kernel convert per line / per touched pixel.)
JD uint32_t to_u32(float f) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JD_NO_ASM_CVT)
    uint32_t r;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#else
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
#endif
}
JD int32_t to_i32(float f) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JD_NO_ASM_CVT)
    int32_t r;
    asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#else
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)f;
#endif
}

// ---- binary64 kernels ----
// Every polynomial is evaluated with explicit fused multiply-adds in Estrin form (a fixed tree, written out below):
// half the operations of mul+add Horner and a dependency chain of 4-5 instead of 14-18 -- the flatten kernels that
// use these are bound by exactly that latency.  The CPU oracle executes the identical sequence (oracle/omath.h).
JD double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
JD double reduce_pio2(double x, int* q) {
    const double TWO_OVER_PI = 0.6366197723675814;
    const double PIO2_1 = 1.57079632673412561417e+00;
    const double PIO2_1T = 6.07710050650619224932e-11;
    double k = rint(x * TWO_OVER_PI);
    double r = dfma(-k, PIO2_1T, dfma(-k, PIO2_1, x));
    *q = (int)((long long)k & 3);
    return r;
}
JD double sin_poly(double r) {  // r + r z (S1 + S2 z + ... + S7 z^6), z = r^2
    const double S1 = -1.0 / 6.0, S2 = 1.0 / 120.0, S3 = -1.0 / 5040.0, S4 = 1.0 / 362880.0, S5 = -1.0 / 39916800.0,
                 S6 = 1.0 / 6227020800.0, S7 = -1.0 / 1307674368000.0;
    double z = r * r;
    double z2 = z * z;
    double z4 = z2 * z2;
    double a = dfma(z, S2, S1), b = dfma(z, S4, S3), c = dfma(z, S6, S5);
    double ab = dfma(z2, b, a), cd = dfma(z2, S7, c);
    double p = dfma(z4, cd, ab);
    return dfma(r * z, p, r);
}
JD double cos_poly(double r) {  // 1 + z (C1 + C2 z + ... + C8 z^7)
    const double C1 = -0.5, C2 = 1.0 / 24.0, C3 = -1.0 / 720.0, C4 = 1.0 / 40320.0, C5 = -1.0 / 3628800.0,
                 C6 = 1.0 / 479001600.0, C7 = -1.0 / 87178291200.0, C8 = 1.0 / 20922789888000.0;
    double z = r * r;
    double z2 = z * z;
    double z4 = z2 * z2;
    double a = dfma(z, C2, C1), b = dfma(z, C4, C3), c = dfma(z, C6, C5), d = dfma(z, C8, C7);
    double ab = dfma(z2, b, a), cd = dfma(z2, d, c);
    double p = dfma(z4, cd, ab);
    return dfma(z, p, 1.0);
}
JD double dsin(double x) {
    int q;
    double r = reduce_pio2(x, &q);
    double s = (q & 1) ? cos_poly(r) : sin_poly(r);
    return (q & 2) ? -s : s;
}
JD double dcos(double x) {
    int q;
    double r = reduce_pio2(x, &q);
    double c = (q & 1) ? sin_poly(r) : cos_poly(r);
    return ((q + 1) & 2) ? -c : c;
}
// atan(k/8), k = 0..8.  The index differs per lane, so a `switch` compiles to a tree of divergent branches (~50 scalar
// instructions and their pipeline bubbles per call) and a chain of selects to ~35 vector instructions (every 32-bit half of
// every constant needs a move).  Kernels that call atan2_ / acos_ / asin_ in a hot loop therefore keep the nine constants in
// LDS: define JD_ATAN_TAB_LDS as a `__shared__ double[9]` before including this header and fill it with atan_tab_fill()
// at the top of the kernel -- one ds_read_b64 per call.  Same constants, same result either way.
JD double atan_tab_const(int k) {
    // (a tree over the bits of k; any k outside 0..8 reads as 8, like a switch's default)
    const bool b0 = (k & 1) != 0, b1 = (k & 2) != 0, b2 = (k & 4) != 0, hi = (unsigned)k >= 8u;
    const double t01 = b0 ? 0.12435499454676144 : 0.0, t23 = b0 ? 0.35877067027057225 : 0.24497866312686414;
    const double t45 = b0 ? 0.5585993153435624 : 0.4636476090008061, t67 = b0 ? 0.7188299996216245 : 0.6435011087932844;
    const double t03 = b1 ? t23 : t01, t47 = b1 ? t67 : t45;
    const double t07 = b2 ? t47 : t03;
    return hi ? 0.7853981633974483 : t07;
}
#ifdef JD_ATAN_TAB_LDS
JD void atan_tab_fill() {  // every thread of the workgroup must call it (before its first atan2_ / acos_ / asin_)
    if (threadIdx.x < 9u) JD_ATAN_TAB_LDS[threadIdx.x] = atan_tab_const((int)threadIdx.x);
    __syncthreads();
}
JD double atan_tab(int k) { return JD_ATAN_TAB_LDS[(unsigned)k > 8u ? 8u : (unsigned)k]; }
#else
JD double atan_tab(int k) { return atan_tab_const(k); }
#endif
// atan(n/d) for 0 <= n <= d, d > 0: split at k/8, atan(n/d) = atan(k/8) + atan(t), t = (n - c d) / (d + c n), c = k/8.
// k is picked from the binary32 quotient (any k with |n/d - k/8| <= 1/16 + 2^-24 keeps |t| < 0.07), so the only
// binary64 division is the one of t.
JD double datan_frac(double n, double d) {
    float af = (float)n / (float)d;  // IEEE binary32 division of the operands rounded to binary32 (exact for atan2_'s)
    if (!(af >= 0.0f && af <= 1.0f)) return (double)af;  // NaN passes through
    float kf = rintf(af * 8.0f);
    int k = (int)kf;
    double c = (double)kf * 0.125;
    double t = dfma(-c, d, n) / dfma(c, n, d);
    const double A1 = -1.0 / 3.0, A2 = 1.0 / 5.0, A3 = -1.0 / 7.0, A4 = 1.0 / 9.0, A5 = -1.0 / 11.0, A6 = 1.0 / 13.0;
    double z = t * t;
    double z2 = z * z;
    double z4 = z2 * z2;
    double a = dfma(z, A2, A1), b = dfma(z, A4, A3), cc = dfma(z, A6, A5);
    double ab = dfma(z2, b, a);
    double p = dfma(z4, cc, ab);
    return atan_tab(k) + dfma(t * z, p, t);
}
JD bool dsignbit(double x) { return (__double_as_longlong(x) >> 63) != 0; }
// (operands: binary32 values widened to binary64, or sqrt(1 - x^2) next to such an x -- all well inside binary32's range)
JD double datan2(double y, double x) {
    const double PI = 3.141592653589793;
    const double PIO2 = 1.5707963267948966;
    double ax = fabs(x), ay = fabs(y);
    // one evaluation on (smaller, larger) instead of one per branch (on the GPU both branches of a divergent wave run)
    const bool swap = !(ay <= ax);
    double f = datan_frac(swap ? ax : ay, swap ? ay : ax);
    double r = swap ? PIO2 - f : f;
    if (ax == 0.0 && ay == 0.0) r = 0.0;
    if (dsignbit(x)) r = PI - r;
    return dsignbit(y) ? -r : r;
}
JD double dsqrt1mx2(double x) { return sqrt(dfma(-x, x, 1.0)); }  // sqrt(1 - x^2), the product not rounded
// |x|^(2/3) = x * x^(-1/3): bit-trick seed for the inverse cube root (relative error < 6 %), four division-free Newton
// steps r <- r (4 - x r^3) / 3 (error -> 2 e^2: 6e-2, 7e-3, 1e-4, 2e-8, 1e-15), then one more for the rounding.
JD double dpow23(double ax) {
    if (ax == 0.0) return 0.0;
    uint64_t hx = (uint64_t)__double_as_longlong(ax) >> 32;
    double r = __longlong_as_double((long long)((uint64_t)(0x553EF0FFu - (uint32_t)(hx / 3u)) << 32));
    const double THIRD = 1.0 / 3.0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        double r3 = (r * r) * r;
        double h = dfma(-ax, r3, 4.0);
        r = (r * h) * THIRD;
    }
    return ax * r;
}

JD float sin_(float x) { return (float)dsin((double)x); }
JD float cos_(float x) { return (float)dcos((double)x); }
JD float atan2_(float y, float x) { return (float)datan2((double)y, (double)x); }
JD float acos_(float x) { return (float)datan2(dsqrt1mx2((double)x), (double)x); }
JD float asin_(float x) { return (float)datan2((double)x, dsqrt1mx2((double)x)); }
JD float pow23_abs_(float x) { return (float)dpow23((double)abs_(x)); }

// binary16 <-> binary32 (hardware v_cvt, RTNE)
JD float f16_to_f32(uint16_t h) {
    union { uint16_t u; _Float16 f; } c;
    c.u = h;
    return (float)c.f;
}
JD uint16_t f32_to_f16(float f) {
    union { uint16_t u; _Float16 h; } c;
    c.h = (_Float16)f;
    return c.u;
}

struct V2 {
    float x, y;
};
JD V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
JD V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
JD V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
JD V2 operator*(V2 a, float s) { return v2(a.x * s, a.y * s); }
JD V2 operator*(float s, V2 a) { return v2(s * a.x, s * a.y); }
JD V2 operator-(V2 a) { return v2(-a.x, -a.y); }
JD float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
JD float length(V2 a) { return sqrt_(a.x * a.x + a.y * a.y); }
JD V2 normalize(V2 a) { float l = length(a); return v2(a.x / l, a.y / l); }
JD V2 vmix(V2 a, V2 b, float t) { return v2(mix_(a.x, b.x, t), mix_(a.y, b.y, t)); }
JD bool veq(V2 a, V2 b) { return a.x == b.x && a.y == b.y; }

struct Xf {  // shared/transform.wgsl Transform
    float m0, m1, m2, m3, t0, t1;
};
JD V2 xf_apply(const Xf& t, V2 p) { return v2(t.m0 * p.x + t.m2 * p.y + t.t0, t.m1 * p.x + t.m3 * p.y + t.t1); }
JD Xf xf_identity() { Xf r; r.m0 = 1.0f; r.m1 = 0.0f; r.m2 = 0.0f; r.m3 = 1.0f; r.t0 = 0.0f; r.t1 = 0.0f; return r; }
JD Xf xf_inverse(const Xf& t) {
    float inv_det = 1.0f / (t.m0 * t.m3 - t.m1 * t.m2);
    Xf r;
    r.m0 = inv_det * t.m3; r.m1 = inv_det * -t.m1; r.m2 = inv_det * -t.m2; r.m3 = inv_det * t.m0;
    float ntx = -t.t0, nty = -t.t1;
    r.t0 = r.m0 * ntx + r.m2 * nty;
    r.t1 = r.m1 * ntx + r.m3 * nty;
    return r;
}
JD Xf xf_mul(const Xf& a, const Xf& b) {
    Xf r;
    r.m0 = a.m0 * b.m0 + a.m2 * b.m1;
    r.m1 = a.m1 * b.m0 + a.m3 * b.m1;
    r.m2 = a.m0 * b.m2 + a.m2 * b.m3;
    r.m3 = a.m1 * b.m2 + a.m3 * b.m3;
    r.t0 = a.m0 * b.t0 + a.m2 * b.t1 + a.t0;
    r.t1 = a.m1 * b.t0 + a.m3 * b.t1 + a.t1;
    return r;
}
JD Xf xf_read(const uint32_t* scene, uint32_t transform_base, uint32_t ix) {
    const uint32_t* p = scene + transform_base + ix * 6u;
    Xf r;
    r.m0 = u2f(p[0]); r.m1 = u2f(p[1]); r.m2 = u2f(p[2]); r.m3 = u2f(p[3]); r.t0 = u2f(p[4]); r.t1 = u2f(p[5]);
    return r;
}

}  // namespace jd
