// encoding.cpp -- see encoding.h.  Behaviour follows encoding/encoding.go and encoding/path.go
// statement by statement (state machine of the path encoder included), because the byte streams
// are the kernels' input contract.
#include "encoding.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "jello_formats.h"

namespace jello {

static const uint32_t kForceNextTransform = 1, kForceNextStyle = 2;  // encoding.go:34-37

Style style_from_fill(Fill fill) {
    Style s;
    s.flags_and_miter_limits = (fill == Fill::EvenOdd) ? JL_STYLE_FLAGS_FILL : 0u;
    s.line_width = 0;
    return s;
}

Style style_from_stroke(const Stroke& st) {
    uint32_t style = JL_STYLE_FLAGS_STYLE;
    uint32_t join = 0;
    switch (st.join) {
        case Join::Bevel: join = JL_STYLE_FLAGS_JOIN_BEVEL; break;
        case Join::Miter: join = JL_STYLE_FLAGS_JOIN_MITER; break;
        case Join::Round: join = JL_STYLE_FLAGS_JOIN_ROUND; break;
    }
    auto cap_bits = [](Cap c) -> uint32_t {
        switch (c) {
            case Cap::Butt: return JL_STYLE_FLAGS_CAP_BUTT;
            case Cap::Square: return JL_STYLE_FLAGS_CAP_SQUARE;
            case Cap::Round: return JL_STYLE_FLAGS_CAP_ROUND;
        }
        return 0;
    };
    uint32_t start_cap = cap_bits(st.start_cap) << 2;
    uint32_t end_cap = cap_bits(st.end_cap);
    uint32_t miter_limit = float16_bits((float)st.miter_limit);
    Style s;
    s.flags_and_miter_limits = style | join | start_cap | end_cap | miter_limit;
    s.line_width = (float)st.width;
    return s;
}

void Encoding::reset() {
    path_tags.clear(); path_data.clear(); draw_tags.clear(); draw_data.clear(); transforms.clear(); styles.clear();
    resources.reset();
    num_paths = num_path_segments = num_clips = num_open_clips = flags = 0;
}

StreamOffsets Encoding::stream_offsets() const {
    StreamOffsets o;
    o.path_tags = (int)path_tags.size(); o.path_data = (int)path_data.size(); o.draw_tags = (int)draw_tags.size();
    o.draw_data = (int)draw_data.size(); o.transforms = (int)transforms.size(); o.styles = (int)styles.size();
    return o;
}

// encoding.go:59-112
void Encoding::append(const Encoding& other, const Transform& transform) {
    StreamOffsets offsets = stream_offsets();
    int stops_base = (int)resources.color_stops.size();
    for (const Patch& p : other.resources.patches) {
        Patch np = p;
        if (p.kind == Patch::Ramp) {
            np.ramp.draw_data_offset = p.ramp.draw_data_offset + offsets.draw_data;
            np.ramp.stops[0] = p.ramp.stops[0] + stops_base;
            np.ramp.stops[1] = p.ramp.stops[1] + stops_base;
        } else {
            np.image.draw_data_offset = p.image.draw_data_offset + offsets.draw_data;
        }
        resources.patches.push_back(np);
    }
    resources.color_stops.insert(resources.color_stops.end(), other.resources.color_stops.begin(), other.resources.color_stops.end());
    path_tags.insert(path_tags.end(), other.path_tags.begin(), other.path_tags.end());
    path_data.insert(path_data.end(), other.path_data.begin(), other.path_data.end());
    draw_tags.insert(draw_tags.end(), other.draw_tags.begin(), other.draw_tags.end());
    draw_data.insert(draw_data.end(), other.draw_data.begin(), other.draw_data.end());
    num_paths += other.num_paths;
    num_path_segments += other.num_path_segments;
    num_clips += other.num_clips;
    num_open_clips += other.num_open_clips;
    flags = other.flags;
    if (transform != Transform::identity()) {
        for (const Transform& t : other.transforms) transforms.push_back(transform.mul(t));
    } else {
        transforms.insert(transforms.end(), other.transforms.begin(), other.transforms.end());
    }
    styles.insert(styles.end(), other.styles.begin(), other.styles.end());
}

void Encoding::apply_transform(const Transform& t) {
    for (Transform& x : transforms) x = t.mul(x);
}

void Encoding::encode_style(const Style& style) {  // encoding.go:132-138
    if ((flags & kForceNextStyle) != 0 || styles.empty() || !(styles.back() == style)) {
        path_tags.push_back(JL_PATH_TAG_STYLE);
        styles.push_back(style);
        flags &= ~kForceNextStyle;
    }
}

bool Encoding::encode_transform(const Transform& t) {  // encoding.go:140-149
    if ((flags & kForceNextTransform) != 0 || transforms.empty() || transforms.back() != t) {
        path_tags.push_back(JL_PATH_TAG_TRANSFORM);
        transforms.push_back(t);
        flags &= ~kForceNextTransform;
        return true;
    }
    return false;
}

void Encoding::encode_empty_shape() {  // encoding.go:166-170
    PathEncoder pe(&path_tags, &path_data, &num_path_segments, &num_paths, true);
    pe.empty_path();
    pe.finish(true);
}

bool Encoding::encode_path(const BezPath& path, bool is_fill) {  // encoding.go:172-176
    PathEncoder pe(&path_tags, &path_data, &num_path_segments, &num_paths, is_fill);
    pe.path(path);
    return pe.finish(true) != 0;
}

void Encoding::push_u32(std::vector<uint8_t>& v, uint32_t x) {
    uint8_t b[4];
    std::memcpy(b, &x, 4);
    v.insert(v.end(), b, b + 4);
}

void Encoding::encode_color(const float rgba[4]) {  // encoding.go:232-238
    draw_tags.push_back(JL_DRAWTAG_FILL_COLOR);
    for (int i = 0; i < 4; i++) push_u32(draw_data, f32_bits(rgba[i]));
}

void Encoding::encode_brush(const Brush& b, float alpha) {  // encoding.go:178-230
    switch (b.kind) {
        case Brush::Solid: {
            Color c = b.color;
            c.a *= (double)alpha;
            float rgba[4];
            premul32(c, rgba);
            encode_color(rgba);
            break;
        }
        case Brush::Linear: {
            float p0[2] = {(float)b.p0[0], (float)b.p0[1]}, p1[2] = {(float)b.p1[0], (float)b.p1[1]};
            encode_linear_gradient(p0, p1, b.stops, alpha, b.extend);
            break;
        }
        case Brush::Radial: {
            float p0[2] = {(float)b.p0[0], (float)b.p0[1]}, p1[2] = {(float)b.p1[0], (float)b.p1[1]};
            encode_radial_gradient(p0, p1, b.r0, b.r1, b.stops, alpha, b.extend);
            break;
        }
        case Brush::Sweep: {
            float p0[2] = {(float)b.p0[0], (float)b.p0[1]};
            const float two_pi = (float)(2.0 * M_PI);
            encode_sweep_gradient(p0, b.t0 / two_pi, b.t1 / two_pi, b.stops, alpha, b.extend);
            break;
        }
        case Brush::ImageBrush: encode_image(b.image, 1.0f); break;
    }
}

void Encoding::add_ramp(const std::vector<ColorStop>& color_stops, float alpha, Extend extend) {  // encoding.go:240-263
    if (color_stops.size() < 2) throw std::logic_error("add_ramp called with less than 2 color stops");
    int offset = (int)draw_data.size();
    int stops_start = (int)resources.color_stops.size();
    for (ColorStop cs : color_stops) {
        if (alpha != 1.0f) cs.color.a = (double)alpha;  // gfx.ColorStop.WithAlphaFactor *sets* alpha (gradient.go:16-24)
        resources.color_stops.push_back(cs);
    }
    int stops_end = (int)resources.color_stops.size();
    Patch p;
    p.kind = Patch::Ramp;
    p.ramp = RampPatch{offset, {stops_start, stops_end}, extend};
    resources.patches.push_back(p);
}

static void single_stop_color(const std::vector<ColorStop>& stops, float alpha, float rgba[4]) {
    Color c = stops[0].color;
    c.a *= (double)alpha;
    premul32(c, rgba);
}

void Encoding::encode_linear_gradient(const float p0[2], const float p1[2], const std::vector<ColorStop>& stops, float alpha, Extend extend) {
    const float zero[4] = {0, 0, 0, 0};
    if (stops.empty()) { encode_color(zero); return; }
    if (stops.size() == 1) { float c[4]; single_stop_color(stops, alpha, c); encode_color(c); return; }
    add_ramp(stops, alpha, extend);
    draw_tags.push_back(JL_DRAWTAG_FILL_LIN_GRADIENT);
    push_u32(draw_data, 0);
    push_u32(draw_data, f32_bits(p0[0])); push_u32(draw_data, f32_bits(p0[1]));
    push_u32(draw_data, f32_bits(p1[0])); push_u32(draw_data, f32_bits(p1[1]));
}

void Encoding::encode_radial_gradient(const float p0[2], const float p1[2], float r0, float r1, const std::vector<ColorStop>& stops,
                                      float alpha, Extend extend) {
    const float zero[4] = {0, 0, 0, 0};
    const float skia_epsilon = 1.0f / (float)(1 << 12);
    if (p0[0] == p1[0] && p0[1] == p1[1] && std::fabs(r0 - r1) < skia_epsilon) { encode_color(zero); return; }
    if (stops.empty()) { encode_color(zero); return; }
    if (stops.size() == 1) { float c[4]; single_stop_color(stops, alpha, c); encode_color(c); return; }
    add_ramp(stops, alpha, extend);
    draw_tags.push_back(JL_DRAWTAG_FILL_RAD_GRADIENT);
    push_u32(draw_data, 0);
    push_u32(draw_data, f32_bits(p0[0])); push_u32(draw_data, f32_bits(p0[1]));
    push_u32(draw_data, f32_bits(p1[0])); push_u32(draw_data, f32_bits(p1[1]));
    push_u32(draw_data, f32_bits(r0)); push_u32(draw_data, f32_bits(r1));
}

void Encoding::encode_sweep_gradient(const float p0[2], float t0, float t1, const std::vector<ColorStop>& stops, float alpha, Extend extend) {
    const float zero[4] = {0, 0, 0, 0};
    const float degenerate = 1.0f / (float)(1 << 15);
    if (std::fabs(t0 - t1) < degenerate) { encode_color(zero); return; }
    if (stops.empty()) { encode_color(zero); return; }
    if (stops.size() == 1) { float c[4]; single_stop_color(stops, alpha, c); encode_color(c); return; }
    add_ramp(stops, alpha, extend);
    draw_tags.push_back(JL_DRAWTAG_FILL_SWEEP_GRADIENT);
    push_u32(draw_data, 0);
    push_u32(draw_data, f32_bits(p0[0])); push_u32(draw_data, f32_bits(p0[1]));
    push_u32(draw_data, f32_bits(t0)); push_u32(draw_data, f32_bits(t1));
}

void Encoding::encode_image(const Image& img, float /*alpha*/) {  // encoding.go:342-354
    Patch p;
    p.kind = Patch::ImageK;
    p.image = ImagePatch{(int)draw_data.size(), img};
    resources.patches.push_back(p);
    draw_tags.push_back(JL_DRAWTAG_FILL_IMAGE);
    push_u32(draw_data, 0);
    push_u32(draw_data, (img.width << 16) | (img.height & 0xFFFFu));
}

void Encoding::encode_begin_clip(BlendMode blend, float alpha) {  // encoding.go:356-366
    draw_tags.push_back(JL_DRAWTAG_BEGIN_CLIP);
    push_u32(draw_data, ((uint32_t)blend.mix << 8) | (uint32_t)blend.compose);
    push_u32(draw_data, f32_bits(alpha));
    num_clips++;
    num_open_clips++;
}

void Encoding::encode_end_clip() {  // encoding.go:368-378
    if (num_open_clips == 0) return;
    draw_tags.push_back(JL_DRAWTAG_END_CLIP);
    path_tags.push_back(JL_PATH_TAG_PATH);
    num_paths++;
    num_clips++;
    num_open_clips--;
}

void Encoding::swap_last_path_tags() {  // encoding.go:384-387
    size_t n = path_tags.size();
    std::swap(path_tags[n - 2], path_tags[n - 1]);
}

// ------------------------------------------------------------------------------------------
// PathEncoder (encoding/path.go:177-489)
// ------------------------------------------------------------------------------------------
void PathEncoder::push_f32(float v) {
    uint8_t b[4];
    std::memcpy(b, &v, 4);
    data_->insert(data_->end(), b, b + 4);
}

bool PathEncoder::last_point(float out[2]) const {
    size_t n = data_->size();
    if (n < 8) return false;
    std::memcpy(&out[0], data_->data() + n - 8, 4);
    std::memcpy(&out[1], data_->data() + n - 4, 4);
    return true;
}

void PathEncoder::move_to(float x, float y) {
    if (is_fill_) close();
    if (state_ == MoveTo) {
        data_->resize(data_->size() - 8);
    } else if (state_ == NonemptySubpath) {
        if (!is_fill_) insert_stroke_cap_marker_segment(false);
        if (!tags_->empty()) tags_->back() |= JL_PATH_TAG_SUBPATH_END;
    }
    first_point_[0] = x; first_point_[1] = y;
    push_f32(x); push_f32(y);
    state_ = MoveTo;
}

bool PathEncoder::is_zero_length_segment(const float p1[2], const float* p2_, const float* p3_) const {
    float p0[2];
    if (!last_point(p0)) throw std::logic_error("unreachable");
    const float* p2 = p2_ ? p2_ : p1;
    const float* p3 = p3_ ? p3_ : p1;
    float x_min = std::min(std::min(p0[0], p1[0]), std::min(p2[0], p3[0]));
    float x_max = std::max(std::max(p0[0], p1[0]), std::max(p2[0], p3[0]));
    float y_min = std::min(std::min(p0[1], p1[1]), std::min(p2[1], p3[1]));
    float y_max = std::max(std::max(p0[1], p1[1]), std::max(p2[1], p3[1]));
    return !(x_max - x_min > kEpsilon || y_max - y_min > kEpsilon);
}

bool PathEncoder::start_tangent_for_curve(const float p1[2], const float* p2_, const float* p3_, float out[2]) const {
    const float* p0 = first_point_;
    const float* p2 = p2_ ? p2_ : p0;
    const float* p3 = p3_ ? p3_ : p0;
    auto is_far = [&](const float* p) { return std::fabs(p[0] - p0[0]) > kEpsilon || std::fabs(p[1] - p0[1]) > kEpsilon; };
    const float* pick = nullptr;
    if (is_far(p1)) pick = p1; else if (is_far(p2)) pick = p2; else if (is_far(p3)) pick = p3;
    if (!pick) return false;
    out[0] = pick[0]; out[1] = pick[1];
    return true;
}

void PathEncoder::line_to(float x, float y) {
    if (state_ == Start) {
        if (num_encoded_segments_ == 0) { move_to(x, y); return; }
        move_to(first_point_[0], first_point_[1]);
    }
    const float p1[2] = {x, y};
    if (state_ == MoveTo) {
        float pt[2];
        if (start_tangent_for_curve(p1, nullptr, nullptr, pt)) { first_start_tangent_end_[0] = pt[0]; first_start_tangent_end_[1] = pt[1]; }
        else return;
    }
    if (is_zero_length_segment(p1, nullptr, nullptr)) return;
    push_f32(x); push_f32(y);
    tags_->push_back(JL_PATH_TAG_LINETO | JL_PATH_TAG_F32);
    state_ = NonemptySubpath;
    num_encoded_segments_++;
}

void PathEncoder::quad_to(float x1, float y1, float x2, float y2) {
    if (state_ == Start) {
        if (num_encoded_segments_ == 0) { move_to(x2, y2); return; }
        move_to(first_point_[0], first_point_[1]);
    }
    const float p1[2] = {x1, y1}, p2[2] = {x2, y2};
    if (state_ == MoveTo) {
        const float zero[2] = {0, 0};  // path.go:297 passes &[2]float32{} for p3
        float pt[2];
        if (!start_tangent_for_curve(p1, p2, zero, pt)) return;
        first_start_tangent_end_[0] = pt[0]; first_start_tangent_end_[1] = pt[1];
    }
    if (is_zero_length_segment(p1, p2, nullptr)) return;
    push_f32(x1); push_f32(y1); push_f32(x2); push_f32(y2);
    tags_->push_back(JL_PATH_TAG_QUADTO | JL_PATH_TAG_F32);
    state_ = NonemptySubpath;
    num_encoded_segments_++;
}

void PathEncoder::cubic_to(float x1, float y1, float x2, float y2, float x3, float y3) {
    if (state_ == Start) {
        if (num_encoded_segments_ == 0) { move_to(x3, y3); return; }
        move_to(first_point_[0], first_point_[1]);
    }
    const float p1[2] = {x1, y1}, p2[2] = {x2, y2}, p3[2] = {x3, y3};
    if (state_ == MoveTo) {
        float pt[2];
        if (!start_tangent_for_curve(p1, p2, p3, pt)) return;
        first_start_tangent_end_[0] = pt[0]; first_start_tangent_end_[1] = pt[1];
    }
    if (is_zero_length_segment(p1, p2, p3)) return;
    push_f32(x1); push_f32(y1); push_f32(x2); push_f32(y2); push_f32(x3); push_f32(y3);
    tags_->push_back(JL_PATH_TAG_CUBICTO | JL_PATH_TAG_F32);
    state_ = NonemptySubpath;
    num_encoded_segments_++;
}

void PathEncoder::close() {
    switch (state_) {
        case Start: return;
        case MoveTo:
            data_->resize(data_->size() - 8);
            state_ = Start;
            return;
        default: break;
    }
    if (data_->size() < 8) return;
    uint8_t first_bytes[8];
    std::memcpy(first_bytes, &first_point_[0], 4);
    std::memcpy(first_bytes + 4, &first_point_[1], 4);
    if (std::memcmp(data_->data() + data_->size() - 8, first_bytes, 8) != 0) {
        data_->insert(data_->end(), first_bytes, first_bytes + 8);
        tags_->push_back(JL_PATH_TAG_LINETO | JL_PATH_TAG_F32);
        num_encoded_segments_++;
    }
    if (!is_fill_) insert_stroke_cap_marker_segment(true);
    if (!tags_->empty()) tags_->back() |= JL_PATH_TAG_SUBPATH_END;
    state_ = Start;
}

void PathEncoder::path(const BezPath& p) {
    for (const PathEl& el : p) {
        switch (el.kind) {
            case PathElKind::MoveTo: move_to((float)el.p0[0], (float)el.p0[1]); break;
            case PathElKind::LineTo: line_to((float)el.p0[0], (float)el.p0[1]); break;
            case PathElKind::QuadTo: quad_to((float)el.p0[0], (float)el.p0[1], (float)el.p1[0], (float)el.p1[1]); break;
            case PathElKind::CubicTo:
                cubic_to((float)el.p0[0], (float)el.p0[1], (float)el.p1[0], (float)el.p1[1], (float)el.p2[0], (float)el.p2[1]);
                break;
            case PathElKind::ClosePath: close(); break;
        }
    }
}

uint32_t PathEncoder::finish(bool insert_path_marker) {
    if (is_fill_) close();
    if (state_ == MoveTo) data_->resize(data_->size() - 8);
    if (num_encoded_segments_ != 0) {
        if (!is_fill_ && state_ == NonemptySubpath) insert_stroke_cap_marker_segment(false);
        if (!tags_->empty()) tags_->back() |= JL_PATH_TAG_SUBPATH_END;
        *num_segments_ += num_encoded_segments_;
        if (insert_path_marker) {
            tags_->push_back(JL_PATH_TAG_PATH);
            *num_paths_ += 1;
        }
    }
    return num_encoded_segments_;
}

void PathEncoder::insert_stroke_cap_marker_segment(bool is_closed) {
    if (is_fill_) throw std::logic_error("invalid state");
    if (state_ != NonemptySubpath) throw std::logic_error("invalid state");
    if (is_closed) {
        line_to(first_start_tangent_end_[0], first_start_tangent_end_[1]);
    } else {
        quad_to(first_point_[0], first_point_[1], first_start_tangent_end_[0], first_start_tangent_end_[1]);
    }
}

void PathEncoder::empty_path() {
    data_->insert(data_->end(), 16, 0);
    tags_->push_back(JL_PATH_TAG_LINETO | JL_PATH_TAG_F32);
    num_encoded_segments_++;
}

}  // namespace jello
