// gfx.h -- drawing vocabulary of the host API: fill rules, blend modes, brushes, path elements.
// Mirrors gfx/{style,blend,brush,gradient,image,color}.go and the curve.BezPath / curve.Stroke
// inputs of scene.go.  The reference takes colours as honnef.co/go/color values and converts
// them with Convert(LinearSRGB) (gfx/color.go:27-39); that package is an un-vendored third-party
// dependency, so this API takes colours already in linear sRGB, un-premultiplied.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "jmath.h"

namespace jello {

enum class Fill : int { NonZero = 0, EvenOdd = 1 };                  // gfx/style.go:7-12
enum class Extend : int { Pad = 0, Repeat = 1, Reflect = 2 };       // gfx/brush.go:27-33

// gfx/blend.go:21-88.  Jello renumbers Compose so that SrcOver == 0 (blend.go:12-16).
enum class Mix : uint8_t {
    Normal = 0, Multiply = 1, Screen = 2, Overlay = 3, Darken = 4, Lighten = 5, ColorDodge = 6, ColorBurn = 7,
    HardLight = 8, SoftLight = 9, Difference = 10, Exclusion = 11, Hue = 12, Saturation = 13, Color = 14,
    Luminosity = 15, Clip = 128
};
enum class Compose : uint8_t {
    SrcOver = 0, Copy = 1, Dest = 2, Clear = 3, DestOver = 4, SrcIn = 5, DestIn = 6, SrcOut = 7, DestOut = 8,
    SrcAtop = 9, DestAtop = 10, Xor = 11, Plus = 12, PlusLighter = 13
};
struct BlendMode { Mix mix = Mix::Normal; Compose compose = Compose::SrcOver; };

struct Color { double r = 0, g = 0, b = 0, a = 0; };  // linear sRGB, un-premultiplied

// gfx/color.go:27-39 Premul32
static inline void premul32(const Color& c, float out[4]) {
    out[0] = (float)(c.r * c.a);
    out[1] = (float)(c.g * c.a);
    out[2] = (float)(c.b * c.a);
    out[3] = (float)c.a;
}

struct ColorStop { float offset = 0; Color color; };                  // gfx/gradient.go:11-24

struct Image {  // gfx/image.go -- RGBA8 pixels, row-major
    uint32_t width = 0, height = 0;
    const uint8_t* pixels = nullptr;
    uint64_t key = 0;  // identity for de-duplication (the Go code keys on the image.Image pointer)
    // In Go the garbage collector keeps the image.Image alive for as long as an encoding points at it.  Here the brush
    // OWNS its pixels when it came through the C API (`pixels` points into `owned`): a caller may free its array as
    // soon as Scene.fill / stroke returns, the patch is read at render time (renderer.cpp, upload_image).
    std::shared_ptr<const std::vector<uint8_t>> owned;
};

struct Brush {
    enum Kind { Solid, Linear, Radial, Sweep, ImageBrush } kind = Solid;
    Color color;                       // Solid
    double p0[2] = {0, 0}, p1[2] = {0, 0};  // Linear: start/end; Radial: centres; Sweep: centre in p0
    float r0 = 0, r1 = 0;              // Radial radii
    float t0 = 0, t1 = 0;              // Sweep start/end angle (radians)
    std::vector<ColorStop> stops;
    Extend extend = Extend::Pad;
    Image image;

    static Brush solid(const Color& c) { Brush b; b.kind = Solid; b.color = c; return b; }
};

// curve.Affine: x' = c0 x + c2 y + c4, y' = c1 x + c3 y + c5 (float64, like the reference's inputs)
struct Affine {
    double c[6] = {1, 0, 0, 1, 0, 0};
    bool is_identity() const { return c[0] == 1 && c[1] == 0 && c[2] == 0 && c[3] == 1 && c[4] == 0 && c[5] == 0; }
    Affine mul(const Affine& o) const {
        Affine r;
        r.c[0] = c[0] * o.c[0] + c[2] * o.c[1];
        r.c[1] = c[1] * o.c[0] + c[3] * o.c[1];
        r.c[2] = c[0] * o.c[2] + c[2] * o.c[3];
        r.c[3] = c[1] * o.c[2] + c[3] * o.c[3];
        r.c[4] = c[0] * o.c[4] + c[2] * o.c[5] + c[4];
        r.c[5] = c[1] * o.c[4] + c[3] * o.c[5] + c[5];
        return r;
    }
    Transform to_transform() const { return Transform::from_coeffs(c); }
};

// curve.BezPath element (honnef.co/go/curve, used by scene.go:40-214 and encoding/path.go:407-434)
enum class PathElKind : int { MoveTo = 0, LineTo = 1, QuadTo = 2, CubicTo = 3, ClosePath = 4 };
struct PathEl {
    PathElKind kind;
    double p0[2], p1[2], p2[2];
};
using BezPath = std::vector<PathEl>;

// curve.Stroke subset consumed by encoding/path.go:86-120
enum class Join : int { Bevel = 0, Miter = 1, Round = 2 };
enum class Cap : int { Butt = 0, Square = 1, Round = 2 };
struct Stroke {
    double width = 1.0;
    Join join = Join::Round;
    double miter_limit = 4.0;
    Cap start_cap = Cap::Round, end_cap = Cap::Round;
    // Dash patterns are expanded on the CPU by curve.Dash in the reference (scene.go:169-177);
    // that third-party routine is out of scope here, so dashes are rejected.
    std::vector<double> dash_pattern;
    double dash_offset = 0;
};

}  // namespace jello
