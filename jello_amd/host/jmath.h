// jmath.h -- f32 transform + binary16 helpers used by the host-side encoder.
// Mirrors jmath/jmath.go:89-208 of the reference (Transform, Mul, Float16bits, AlignUp).
#pragma once
#include <cstdint>
#include <cstring>

namespace jello {

struct Transform {  // jmath/jmath.go:89-94 ; column-major 2x2 + translation, x' = a x + c y + e
    float matrix[4] = {1.0f, 0.0f, 0.0f, 1.0f};
    float translation[2] = {0.0f, 0.0f};

    bool operator==(const Transform& o) const {
        return std::memcmp(matrix, o.matrix, sizeof matrix) == 0 && std::memcmp(translation, o.translation, sizeof translation) == 0;
    }
    bool operator!=(const Transform& o) const { return !(*this == o); }

    // jmath/jmath.go:100-119
    Transform mul(const Transform& other) const {
        Transform r;
        r.matrix[0] = matrix[0] * other.matrix[0] + matrix[2] * other.matrix[1];
        r.matrix[1] = matrix[1] * other.matrix[0] + matrix[3] * other.matrix[1];
        r.matrix[2] = matrix[0] * other.matrix[2] + matrix[2] * other.matrix[3];
        r.matrix[3] = matrix[1] * other.matrix[2] + matrix[3] * other.matrix[3];
        r.translation[0] = matrix[0] * other.translation[0] + matrix[2] * other.translation[1] + translation[0];
        r.translation[1] = matrix[1] * other.translation[0] + matrix[3] * other.translation[1] + translation[1];
        return r;
    }
    static Transform identity() { return Transform(); }
    static Transform from_coeffs(const double c[6]) {  // jmath.TransformFromKurbo, jmath.go:191-197
        Transform t;
        for (int i = 0; i < 4; i++) t.matrix[i] = (float)c[i];
        t.translation[0] = (float)c[4];
        t.translation[1] = (float)c[5];
        return t;
    }
};

static inline uint32_t f32_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float f32_from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// jmath/jmath.go:124-162 (Fabian Giesen's float_to_half_fast3)
static inline uint16_t float16_bits(float val) {
    const uint32_t inf32 = 255u << 23, inf16 = 31u << 23, magic = 15u << 23;
    const uint32_t sign_mask = 0x80000000u, round_mask = ~0xFFFu;
    uint32_t u = f32_bits(val);
    uint32_t sign = u & sign_mask;
    u ^= sign;
    uint16_t output;
    if (u >= inf32) {
        output = (u > inf32) ? 0x7E00 : 0x7C00;
    } else {
        uint32_t v = u & round_mask;
        v = f32_bits(f32_from_bits(v) * f32_from_bits(magic));
        v = v - round_mask;
        if (v > inf16) v = inf16;
        output = (uint16_t)(v >> 13);
    }
    return (uint16_t)(output | (uint16_t)(sign >> 16));
}

// RTNE f32 -> f16, used for gradient ramp texels we generate ourselves.
static inline uint16_t float16_bits_rtne(float f) {
    uint32_t x = f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (x < 0x38800000u) {
        if (x < 0x33000000u) return (uint16_t)sign;
        uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u, shift = 126 - e;
        uint32_t h = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = (x - 0x38000000u) >> 13, rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

template <typename T> static inline T align_up(T len, T alignment) { return (len + alignment - 1) & ~(alignment - 1); }

constexpr float kEpsilon = 1e-12f;  // jmath.Epsilon

}  // namespace jello
