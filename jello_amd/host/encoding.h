// encoding.h -- the scene wire format: path tags, path data, draw tags, draw data, transforms,
// styles.  Mirrors encoding/encoding.go:19-438, encoding/path.go:17-489, encoding/draw.go:16-91.
// The byte streams produced here are the kernel input contract.
#pragma once
#include <cstdint>
#include <vector>

#include "gfx.h"
#include "jmath.h"

namespace jello {

struct Style {  // encoding/path.go:17-36
    uint32_t flags_and_miter_limits = 0;
    float line_width = 0;
    bool operator==(const Style& o) const { return flags_and_miter_limits == o.flags_and_miter_limits && f32_bits(line_width) == f32_bits(o.line_width); }
};

Style style_from_fill(Fill fill);            // encoding/path.go:75-84
Style style_from_stroke(const Stroke& s);    // encoding/path.go:86-120

struct RampPatch { int draw_data_offset; int stops[2]; Extend extend; };   // encoding.go:422-428
struct ImagePatch { int draw_data_offset; Image image; };                    // encoding.go:430-435
struct Patch { enum Kind { Ramp, ImageK } kind; RampPatch ramp; ImagePatch image; };

struct Resources {  // encoding.go:408-417
    std::vector<Patch> patches;
    std::vector<ColorStop> color_stops;
    void reset() { patches.clear(); color_stops.clear(); }
};

struct StreamOffsets { int path_tags = 0, path_data = 0, draw_tags = 0, draw_data = 0, transforms = 0, styles = 0; };

class Encoding {  // encoding/encoding.go:19-32
   public:
    std::vector<uint8_t> path_tags;
    std::vector<uint8_t> path_data;
    std::vector<uint32_t> draw_tags;
    std::vector<uint8_t> draw_data;
    std::vector<Transform> transforms;
    std::vector<Style> styles;
    Resources resources;
    uint32_t num_paths = 0, num_path_segments = 0, num_clips = 0, num_open_clips = 0, flags = 0;

    bool is_empty() const { return path_tags.empty(); }
    void reset();
    void append(const Encoding& other, const Transform& transform);
    StreamOffsets stream_offsets() const;
    void apply_transform(const Transform& t);
    void encode_fill_style(Fill fill) { encode_style(style_from_fill(fill)); }
    void encode_stroke_style(const Stroke& s) { encode_style(style_from_stroke(s)); }
    void encode_style(const Style& style);
    bool encode_transform(const Transform& t);
    void encode_empty_shape();
    bool encode_path(const BezPath& path, bool is_fill);
    void encode_brush(const Brush& b, float alpha);
    void encode_color(const float rgba[4]);
    void encode_linear_gradient(const float p0[2], const float p1[2], const std::vector<ColorStop>& stops, float alpha, Extend extend);
    void encode_radial_gradient(const float p0[2], const float p1[2], float r0, float r1, const std::vector<ColorStop>& stops, float alpha,
                                Extend extend);
    void encode_sweep_gradient(const float p0[2], float t0, float t1, const std::vector<ColorStop>& stops, float alpha, Extend extend);
    void encode_image(const Image& img, float alpha);
    void encode_begin_clip(BlendMode blend, float alpha);
    void encode_end_clip();
    void force_next_transform_and_style() { flags |= 3u; }
    void swap_last_path_tags();

   private:
    void add_ramp(const std::vector<ColorStop>& stops, float alpha, Extend extend);
    void push_u32(std::vector<uint8_t>& v, uint32_t x);
};

// encoding/path.go:177-489
class PathEncoder {
   public:
    PathEncoder(std::vector<uint8_t>* tags, std::vector<uint8_t>* data, uint32_t* num_segments, uint32_t* num_paths, bool is_fill)
        : tags_(tags), data_(data), num_segments_(num_segments), num_paths_(num_paths), is_fill_(is_fill) {}
    void move_to(float x, float y);
    void line_to(float x, float y);
    void quad_to(float x1, float y1, float x2, float y2);
    void cubic_to(float x1, float y1, float x2, float y2, float x3, float y3);
    void close();
    void path(const BezPath& p);
    uint32_t finish(bool insert_path_marker);
    void empty_path();

   private:
    enum State { Start, MoveTo, NonemptySubpath };
    bool last_point(float out[2]) const;
    bool is_zero_length_segment(const float p1[2], const float* p2, const float* p3) const;
    bool start_tangent_for_curve(const float p1[2], const float* p2, const float* p3, float out[2]) const;
    void insert_stroke_cap_marker_segment(bool is_closed);
    void push_f32(float v);

    std::vector<uint8_t>* tags_;
    std::vector<uint8_t>* data_;
    uint32_t* num_segments_;
    uint32_t* num_paths_;
    float first_point_[2] = {0, 0};
    float first_start_tangent_end_[2] = {0, 0};
    State state_ = Start;
    uint32_t num_encoded_segments_ = 0;
    bool is_fill_;
};

}  // namespace jello
