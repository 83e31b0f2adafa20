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
// sin/cos: Cody-Waite reduction by pi/2 (fdlibm split), Taylor polynomials on [-pi/4, pi/4].
static inline double reduce_pio2(double x, int* q) {
    const double TWO_OVER_PI = 0.6366197723675814;
    const double PIO2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double PIO2_1T = 6.07710050650619224932e-11;  // pi/2 - PIO2_1
    double k = std::nearbyint(x * TWO_OVER_PI);
    double r = (x - k * PIO2_1) - k * PIO2_1T;
    *q = (int)((long long)k & 3);
    return r;
}
static inline double sin_poly(double r) {
    double z = r * r;
    double p = -1.0 / 1307674368000.0;
    p = 1.0 / 6227020800.0 + z * p;
    p = -1.0 / 39916800.0 + z * p;
    p = 1.0 / 362880.0 + z * p;
    p = -1.0 / 5040.0 + z * p;
    p = 1.0 / 120.0 + z * p;
    p = -1.0 / 6.0 + z * p;
    return r + r * (z * p);
}
static inline double cos_poly(double r) {
    double z = r * r;
    double p = 1.0 / 20922789888000.0;
    p = -1.0 / 87178291200.0 + z * p;
    p = 1.0 / 479001600.0 + z * p;
    p = -1.0 / 3628800.0 + z * p;
    p = 1.0 / 40320.0 + z * p;
    p = -1.0 / 720.0 + z * p;
    p = 1.0 / 24.0 + z * p;
    p = -0.5 + z * p;
    return 1.0 + z * p;
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

// atan on [0,1]: split at k/8, atan(a) = atan(k/8) + atan((a - k/8) / (1 + a k/8)).
static inline double datan01(double a) {
    static const double T[9] = {0.0,
                                0.12435499454676144,
                                0.24497866312686414,
                                0.35877067027057225,
                                0.4636476090008061,
                                0.5585993153435624,
                                0.6435011087932844,
                                0.7188299996216245,
                                0.7853981633974483};
    if (!(a >= 0.0 && a <= 1.0)) return a;  // NaN passes through
    double kf = std::nearbyint(a * 8.0);
    int k = (int)kf;
    double c = kf * 0.125;
    double t = (a - c) / (1.0 + a * c);
    double z = t * t;
    double p = 1.0 / 13.0;
    p = -1.0 / 11.0 + z * p;
    p = 1.0 / 9.0 + z * p;
    p = -1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p;
    p = -1.0 / 3.0 + z * p;
    return T[k] + (t + t * (z * p));
}
static inline double datan2(double y, double x) {
    const double PI = 3.141592653589793;
    const double PIO2 = 1.5707963267948966;
    double ax = std::fabs(x), ay = std::fabs(y);
    double r;
    if (ax == 0.0 && ay == 0.0) {
        r = 0.0;
    } else if (ay <= ax) {
        r = datan01(ay / ax);
    } else {
        r = PIO2 - datan01(ax / ay);
    }
    if (std::signbit(x)) r = PI - r;
    return std::signbit(y) ? -r : r;
}
static inline double dacos(double x) { return datan2(std::sqrt((1.0 - x) * (1.0 + x)), x); }
static inline double dasin(double x) { return datan2(x, std::sqrt((1.0 - x) * (1.0 + x))); }

// cbrt for x > 0 (normal): exponent/3 seed (fdlibm B1) + 4 Halley steps.
static inline double dcbrt_pos(double x) {
    uint64_t hx = d2u(x) >> 32;
    double t = u2d((uint64_t)(hx / 3u + 715094163u) << 32);
    for (int i = 0; i < 4; i++) {
        double t3 = t * t * t;
        t = t * ((t3 + (x + x)) / ((t3 + t3) + x));
    }
    return t;
}
// |x|^(2/3)
static inline double dpow23(double ax) {
    if (ax == 0.0) return 0.0;
    double c = dcbrt_pos(ax);
    return c * c;
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
