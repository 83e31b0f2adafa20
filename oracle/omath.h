// oracle/omath.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Scalar arithmetic rules of the oracle.  The reference's ground truth is the WGSL
// text (engine/wgpu_engine/shaders/original/*.wgsl); WGSL leaves the precision of
// `/`, sqrt, and every transcendental implementation-defined, so "the WGSL path" has
// no single bit pattern.  The oracle pins one:
//
//   * + - * / sqrt floor ceil are IEEE-754 binary32, no contraction (-ffp-contract=off);
//   * min/max are IEEE minNum/maxNum (-0 < +0); clamp/sign/select/mix/fract follow the WGSL spec formulas;
//   * round() is ties-to-even (WGSL), NOT Go's math.Round;
//   * u32(f)/i32(f) saturate (WGSL), NaN -> 0;
//   * sin cos atan2 acos asin pow(x,2/3) are evaluated in binary64 with the fixed
//     sequences of IEEE operations below and rounded ONCE to binary32.  This follows
//     the reference's own CPU twin, which evaluates float64 libm and rounds
//     (jmath/jmath.go:48-87), but replaces libm with explicit polynomials so the HIP
//     kernels can execute the identical operation sequence.  tests/test_oracle_math.py
//     checks these against libm (they agree with correctly-rounded f32 results).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace om {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint64_t d2u(double f) { uint64_t u; std::memcpy(&u, &f, 8); return u; }
static inline double u2d(uint64_t u) { double f; std::memcpy(&f, &u, 8); return f; }

// min/max: IEEE-754 minNum/maxNum with -0 < +0 (what gfx950's v_min_f32/v_max_f32 compute; WGSL leaves the
// NaN behaviour of min/max implementation-defined): a NaN operand yields the other operand.
static inline float fmin_(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return std::signbit(a) ? a : b;
    return (b < a) ? b : a;
}
static inline float fmax_(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return std::signbit(a) ? b : a;
    return (a < b) ? b : a;
}
static inline float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
static inline int32_t imin_(int32_t a, int32_t b) { return (b < a) ? b : a; }
static inline int32_t imax_(int32_t a, int32_t b) { return (a < b) ? b : a; }
static inline int32_t iclamp_(int32_t x, int32_t lo, int32_t hi) { return imin_(imax_(x, lo), hi); }
static inline uint32_t umin_(uint32_t a, uint32_t b) { return (b < a) ? b : a; }
static inline uint32_t umax_(uint32_t a, uint32_t b) { return (a < b) ? b : a; }
static inline float sign_(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
static inline float abs_(float x) { return u2f(f2u(x) & 0x7fffffffu); }
static inline float floor_(float x) { return std::floor(x); }
static inline float ceil_(float x) { return std::ceil(x); }
static inline float round_(float x) { return std::nearbyintf(x); }  // default mode: ties-to-even
static inline float sqrt_(float x) { return std::sqrt(x); }
static inline float fract_(float x) { return x - std::floor(x); }
static inline float mix_(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline float length_(float x, float y) { return sqrt_(x * x + y * y); }

// WGSL scalar conversions saturate.
static inline uint32_t to_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
static inline int32_t to_i32(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)f;
}

// ---------------------------------------------------------------- binary64 kernels
// Every polynomial uses explicit fused multiply-adds (std::fma is exact whether it compiles to vfmadd or calls libm) in
// Estrin form, a fixed tree of operations written out below; jello_amd/csrc/dmath.h executes the identical sequence.
static inline double dfma(double a, double b, double c) { return std::fma(a, b, c); }
// sin/cos: Cody-Waite reduction by pi/2 (fdlibm split), Taylor polynomials on [-pi/4, pi/4].
static inline double reduce_pio2(double x, int* q) {
    const double TWO_OVER_PI = 0.6366197723675814;
    const double PIO2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double PIO2_1T = 6.07710050650619224932e-11;  // pi/2 - PIO2_1
    double k = std::nearbyint(x * TWO_OVER_PI);
    double r = dfma(-k, PIO2_1T, dfma(-k, PIO2_1, x));
    *q = (int)((long long)k & 3);
    return r;
}
static inline double sin_poly(double r) {  // r + r z (S1 + S2 z + ... + S7 z^6), z = r^2
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
static inline double cos_poly(double r) {  // 1 + z (C1 + C2 z + ... + C8 z^7)
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
static inline double dsin(double x) {
    int q;
    double r = reduce_pio2(x, &q);
    double s = (q & 1) ? cos_poly(r) : sin_poly(r);
    return (q & 2) ? -s : s;
}
static inline double dcos(double x) {
    int q;
    double r = reduce_pio2(x, &q);
    double c = (q & 1) ? sin_poly(r) : cos_poly(r);
    return ((q + 1) & 2) ? -c : c;
}

// atan(n/d) for 0 <= n <= d, d > 0: split at k/8, atan(n/d) = atan(k/8) + atan(t), t = (n - c d) / (d + c n), c = k/8.
// k is picked from the binary32 quotient of the operands rounded to binary32 (any k with |n/d - k/8| <= 1/16 + 2^-22 keeps
// |t| < 0.07), so the only binary64 division is the one of t.
static inline double datan_frac(double n, double d) {
    static const double T[9] = {0.0,
                                0.12435499454676144,
                                0.24497866312686414,
                                0.35877067027057225,
                                0.4636476090008061,
                                0.5585993153435624,
                                0.6435011087932844,
                                0.7188299996216245,
                                0.7853981633974483};
    float af = (float)n / (float)d;
    if (!(af >= 0.0f && af <= 1.0f)) return (double)af;  // NaN passes through
    float kf = std::nearbyintf(af * 8.0f);
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
    return T[k] + dfma(t * z, p, t);
}
// (operands: binary32 values widened to binary64, or sqrt(1 - x^2) next to such an x -- all well inside binary32's range)
static inline double datan2(double y, double x) {
    const double PI = 3.141592653589793;
    const double PIO2 = 1.5707963267948966;
    double ax = std::fabs(x), ay = std::fabs(y);
    // one evaluation on (smaller, larger) instead of one per branch (on the GPU both branches of a divergent wave run)
    const bool swap = !(ay <= ax);
    double f = datan_frac(swap ? ax : ay, swap ? ay : ax);
    double r = swap ? PIO2 - f : f;
    if (ax == 0.0 && ay == 0.0) r = 0.0;
    if (std::signbit(x)) r = PI - r;
    return std::signbit(y) ? -r : r;
}
static inline double dsqrt1mx2(double x) { return std::sqrt(dfma(-x, x, 1.0)); }  // sqrt(1 - x^2), the product not rounded
static inline double dacos(double x) { return datan2(dsqrt1mx2(x), x); }
static inline double dasin(double x) { return datan2(x, dsqrt1mx2(x)); }

// |x|^(2/3) = x * x^(-1/3): bit-trick seed for the inverse cube root (relative error < 6 %), division-free Newton steps
// r <- r (4 - x r^3) / 3 (error -> 2 e^2: 6e-2, 7e-3, 1e-4, 2e-8, 1e-15), the fifth for the rounding.
static inline double dpow23(double ax) {
    if (ax == 0.0) return 0.0;
    uint64_t hx = d2u(ax) >> 32;
    double r = u2d((uint64_t)(0x553EF0FFu - (uint32_t)(hx / 3u)) << 32);
    const double THIRD = 1.0 / 3.0;
    for (int i = 0; i < 5; i++) {
        double r3 = (r * r) * r;
        double h = dfma(-ax, r3, 4.0);
        r = (r * h) * THIRD;
    }
    return ax * r;
}

// ---------------------------------------------------------------- f32 entry points
static inline float sin_(float x) { return (float)dsin((double)x); }
static inline float cos_(float x) { return (float)dcos((double)x); }
static inline float atan2_(float y, float x) { return (float)datan2((double)y, (double)x); }
static inline float acos_(float x) { return (float)dacos((double)x); }
static inline float asin_(float x) { return (float)dasin((double)x); }
static inline float pow23_abs_(float x) { return (float)dpow23((double)abs_(x)); }

// IEEE binary16 conversions (exact widening; RTNE narrowing).  Reference: jmath/jmath.go:124-189
// (rygorous float_to_half_fast3 -- note that one truncates the 13th mantissa bit region via
// roundMask; the GPU texture store is RTNE).  The oracle's output store uses RTNE.
static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    return u2f(bits);
}
static inline uint16_t f32_to_f16_rtne(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to inf (>= 65520)
    if (x < 0x38800000u) {                                      // subnormal half or zero
        if (x < 0x33000000u) return (uint16_t)sign;             // < 2^-25 -> 0
        uint32_t e = x >> 23;
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126 - e;  // 14..24
        uint32_t h = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = (x - 0x38000000u) >> 13;
    uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

}  // namespace om
