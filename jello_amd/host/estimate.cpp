// estimate.cpp -- see estimate.h.
#include "estimate.h"

#include <algorithm>
#include <cmath>

namespace jello {

namespace {

constexpr double kRsqrtOfTol = 2.2360679775;              // tol = 0.2 (estimate.go:17)
constexpr double kSqrtOfDegreeTermCubic = 0.86602540378;  // estimate.go:372
constexpr double kSqrtOfDegreeTermQuad = 0.5;             // estimate.go:378

struct V2 { double x, y; };
V2 sub(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
double hyp(V2 v) { return std::hypot(v.x, v.y); }
V2 lerp(V2 a, V2 b, double t) { return {a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t}; }

V2 xform(const Transform& t, V2 v) {  // estimate.go:278-283 (linear part only)
    return {(double)t.matrix[0] * v.x + (double)t.matrix[2] * v.y, (double)t.matrix[1] * v.x + (double)t.matrix[3] * v.y};
}
double transform_scale(const Transform* t) {  // estimate.go:285-296
    if (!t) return 1.0;
    const float* m = t->matrix;
    double v1x = (double)m[0] + (double)m[3], v2x = (double)m[0] - (double)m[3];
    double v1y = (double)m[1] - (double)m[2], v2y = (double)m[1] + (double)m[2];
    return std::sqrt(v1x * v1x + v1y * v1y) + std::sqrt(v2x * v2x + v2y * v2y);
}
uint32_t to_u32(double v) { return v <= 0.0 ? 0u : (v >= 4294967295.0 ? 0xffffffffu : (uint32_t)v); }

double approx_arc_length_cubic(V2 p0, V2 p1, V2 p2, V2 p3) {  // estimate.go:298-303
    double chord = hyp(sub(p3, p0));
    double poly = hyp(sub(p1, p0)) + hyp(sub(p2, p1)) + hyp(sub(p3, p2));
    return 0.5 * (chord + poly);
}
double count_segments_for_cubic(V2 p0, V2 p1, V2 p2, V2 p3, const Transform& t) {  // estimate.go:305-311
    return std::ceil(approx_arc_length_cubic(xform(t, p0), xform(t, p1), xform(t, p2), xform(t, p3)) * 0.0625 * M_SQRT2);
}
double count_segments_for_quadratic(V2 p0, V2 p1, V2 p2, const Transform& t) {  // estimate.go:313-315
    return count_segments_for_cubic(p0, lerp(p1, p0, 0.333333), lerp(p1, p2, 0.333333), p2, t);
}
uint32_t count_segments_for_line(V2 p0, V2 p1, const Transform& t) {  // estimate.go:318-323
    V2 d = xform(t, sub(p0, p1));
    double segs = std::ceil(std::ceil(std::fabs(d.x)) * 0.0625) + std::ceil(std::ceil(std::fabs(d.y)) * 0.0625);
    return std::max(1u, to_u32(segs));
}
uint32_t count_segments_for_line_length(double scaled_width) {  // estimate.go:326-330
    return std::max(1u, to_u32(std::ceil(scaled_width * 0.0625 * M_SQRT2)));
}
double wang_quadratic(V2 p0, V2 p1, V2 p2, const Transform& t) {  // estimate.go:380-385
    V2 v = xform(t, V2{p0.x - 2 * p1.x + p2.x, p0.y - 2 * p1.y + p2.y});
    return std::ceil(kSqrtOfDegreeTermQuad * std::sqrt(hyp(v)) * kRsqrtOfTol);
}
double wang_cubic(V2 p0, V2 p1, V2 p2, V2 p3, const Transform& t) {  // estimate.go:387-395
    V2 v1 = xform(t, V2{p0.x - 2 * p1.x + p2.x, p0.y - 2 * p1.y + p2.y});
    V2 v2 = xform(t, V2{p1.x - 2 * p2.x + p3.x, p1.y - 2 * p2.y + p3.y});
    return std::ceil(kSqrtOfDegreeTermCubic * std::sqrt(std::max(hyp(v1), hyp(v2))) * kRsqrtOfTol);
}
void estimate_arc_lines(double scaled_stroke_width, uint32_t* arc_lines, double* line_len) {  // estimate.go:237-247
    const double min_theta = 1e-6, tol = 0.25;
    double radius = std::max(tol, scaled_stroke_width * 0.5);
    double theta = std::max(2.0 * std::acos(1.0 - tol / radius), min_theta);
    *arc_lines = std::max(2u, to_u32(std::ceil(M_PI / 2 / theta)));
    *line_len = 2.0 * std::sin(theta) * radius;
}

}  // namespace

uint32_t BumpEstimator::LineSoup::scaled_curve_line_count(double scale) const { return to_u32(std::ceil((double)curves * std::sqrt(scale))); }
uint32_t BumpEstimator::LineSoup::tally(double scale) const { return linetos + std::max(scaled_curve_line_count(scale), 5u * curve_count); }
void BumpEstimator::LineSoup::add(const LineSoup& other, double scale) {
    linetos += other.linetos;
    curves += other.scaled_curve_line_count(scale);
    curve_count += other.curve_count;
}

void BumpEstimator::append(const BumpEstimator& other, const Transform* transform) {
    double scale = transform_scale(transform);
    segments_ += to_u32(std::ceil((double)other.segments_ * scale));
    lines_.add(other.lines_, scale);
}

void BumpEstimator::count_path(const BezPath& path, const Transform& t, const Stroke* stroke) {
    uint32_t caps = 1, fill_close_lines = 1, joins = 0, lineto_lines = 0, curve_lines = 0, curve_count = 0, segments = 0;
    bool have_first = false, have_last = false;
    V2 first{0, 0}, last{0, 0};
    const double scale = transform_scale(&t);
    const double scaled_width = stroke ? stroke->width * scale : 0.0;
    const double offset_fudge = std::max(1.0, std::sqrt(scaled_width));
    for (const PathEl& el : path) {
        switch (el.kind) {
            case PathElKind::MoveTo:
                first = V2{el.p0[0], el.p0[1]};
                have_first = true;
                if (!have_last) continue;
                caps += 1;
                if (joins > 0) joins--;
                fill_close_lines += 1;
                segments += count_segments_for_line(first, last, t);
                have_last = false;
                break;
            case PathElKind::ClosePath:
                if (have_last) {
                    joins += 1;
                    lineto_lines += 1;
                    if (have_first) segments += count_segments_for_line(first, last, t);
                }
                last = first;
                have_last = have_first;
                break;
            case PathElKind::LineTo:
                last = V2{el.p0[0], el.p0[1]};
                have_last = true;
                joins += 1;
                lineto_lines += 1;
                if (have_first) segments += count_segments_for_line(first, last, t);
                break;
            case PathElKind::QuadTo: {
                V2 p0;
                if (have_last) p0 = last; else if (have_first) p0 = first; else continue;
                V2 p1{el.p0[0], el.p0[1]}, p2{el.p1[0], el.p1[1]};
                last = p2;
                have_last = true;
                double lines = offset_fudge * wang_quadratic(p0, p1, p2, t);
                curve_lines += to_u32(std::ceil(lines));
                curve_count++;
                joins++;
                double segs = offset_fudge * count_segments_for_quadratic(p0, p1, p2, t);
                segments += to_u32(std::max(std::ceil(segs), std::ceil(lines)));
                break;
            }
            case PathElKind::CubicTo: {
                V2 p0;
                if (have_last) p0 = last; else if (have_first) p0 = first; else continue;
                V2 p1{el.p0[0], el.p0[1]}, p2{el.p1[0], el.p1[1]}, p3{el.p2[0], el.p2[1]};
                last = p3;
                have_last = true;
                double lines = offset_fudge * wang_cubic(p0, p1, p2, p3, t);
                curve_lines += to_u32(std::ceil(lines));
                curve_count += 1;
                joins += 1;
                double segs = count_segments_for_cubic(p0, p1, p2, p3, t);
                segments += to_u32(std::max(std::ceil(segs), std::ceil(lines)));
                break;
            }
        }
    }
    if (!stroke) {
        lines_.linetos += lineto_lines + fill_close_lines;
        lines_.curves += curve_lines;
        lines_.curve_count += curve_count;
        segments_ += segments;
        if (have_first && have_last) segments_ += count_segments_for_line(first, last, t);  // the implicit close
        return;
    }
    // For strokes, double-count the lines to estimate offset curves.
    lines_.linetos += 2 * lineto_lines;
    lines_.curves += 2 * curve_lines;
    lines_.curve_count += 2 * curve_count;
    segments_ += 2 * segments;
    count_stroke_caps(stroke->start_cap, scaled_width, caps);
    count_stroke_caps(stroke->end_cap, scaled_width, caps);
    count_stroke_joins(stroke->join, scaled_width, stroke->miter_limit, joins);
}

BumpEstimate BumpEstimator::tally(const Transform* transform) const {
    double scale = transform_scale(transform);
    uint32_t lines = lines_.tally(scale);
    uint32_t n_segments = std::max(lines, to_u32(std::ceil((double)segments_ * scale)));
    BumpEstimate b;
    b.binning = n_segments;  // (the reference's stand-in; Scene::bump_sizes replaces it by the footprint bound)
    b.seg_counts = n_segments;
    b.segments = n_segments;
    b.lines = lines;
    return b;
}

void BumpEstimator::count_stroke_caps(Cap style, double scaled_width, uint32_t count) {
    switch (style) {
        case Cap::Butt:
            lines_.linetos += count;
            segments_ += count_segments_for_line_length(scaled_width) * count;
            break;
        case Cap::Square:
            lines_.linetos += 3 * count;
            segments_ += count_segments_for_line_length(scaled_width) * count;
            segments_ += 2 * count_segments_for_line_length(0.5 * scaled_width) * count;
            break;
        case Cap::Round: {
            uint32_t arc_lines;
            double line_len;
            estimate_arc_lines(scaled_width, &arc_lines, &line_len);
            lines_.curves += count * arc_lines;
            lines_.curve_count += 1;
            segments_ += count * arc_lines * count_segments_for_line_length(line_len);
            break;
        }
    }
}

void BumpEstimator::count_stroke_joins(Join style, double scaled_width, double miter_limit, uint32_t count) {
    switch (style) {
        case Join::Bevel:
            lines_.linetos += count;
            segments_ += count_segments_for_line_length(scaled_width) * count;
            break;
        case Join::Miter: {
            double max_miter_len = scaled_width * miter_limit;
            lines_.linetos += 2 * count;
            segments_ += 2 * count * count_segments_for_line_length(max_miter_len);
            break;
        }
        case Join::Round: {
            uint32_t arc_lines;
            double line_len;
            estimate_arc_lines(scaled_width, &arc_lines, &line_len);
            lines_.curves += count * arc_lines;
            lines_.curve_count += 1;
            segments_ += count * arc_lines * count_segments_for_line_length(line_len);
            break;
        }
    }
    // Count inner join lines
    lines_.linetos += count;
    segments_ += count_segments_for_line_length(scaled_width) * count;
}

// ---- footprint (this build's bound for tile / binning / ptcl / blend) ----------------------------------------------

static void box_of_points(const Transform& t, const double (*pts)[2], int n, float* box) {
    for (int i = 0; i < n; i++) {
        float x = t.matrix[0] * (float)pts[i][0] + t.matrix[2] * (float)pts[i][1] + t.translation[0];
        float y = t.matrix[1] * (float)pts[i][0] + t.matrix[3] * (float)pts[i][1] + t.translation[1];
        box[0] = std::min(box[0], x); box[1] = std::min(box[1], y);
        box[2] = std::max(box[2], x); box[3] = std::max(box[3], y);
    }
}

void FootprintEstimator::add(const BezPath& path, const Transform& t, const Stroke* stroke) {
    float box[4] = {1e30f, 1e30f, -1e30f, -1e30f};
    for (const PathEl& el : path) {
        const double pts[3][2] = {{el.p0[0], el.p0[1]}, {el.p1[0], el.p1[1]}, {el.p2[0], el.p2[1]}};
        int n = el.kind == PathElKind::CubicTo ? 3 : el.kind == PathElKind::QuadTo ? 2 : el.kind == PathElKind::ClosePath ? 0 : 1;
        box_of_points(t, pts, n, box);
    }
    if (box[0] > box[2]) { box[0] = box[1] = box[2] = box[3] = 0.0f; }
    if (stroke) {
        // half the width, times the longest a cap or join may reach: the miter limit, or sqrt(2) for a square cap
        double reach = 0.5 * stroke->width * std::max(stroke->join == Join::Miter ? stroke->miter_limit : 1.0, M_SQRT2);
        float grow = (float)(reach * transform_scale(&t) * 0.5 + 1.0);  // (transform_scale is the sum of both singular values)
        box[0] -= grow; box[1] -= grow; box[2] += grow; box[3] += grow;
    }
    boxes_.push_back(Box{box[0] - 1.0f, box[1] - 1.0f, box[2] + 1.0f, box[3] + 1.0f});  // one pixel of slack for flattening error
}

void FootprintEstimator::append(const FootprintEstimator& other, const Transform& t) {
    const size_t base = boxes_.size();
    for (size_t o : other.open_) open_.push_back(base + o);  // layers the appended fragment leaves open
    depth_ += (uint32_t)other.open_.size();
    for (const Box& b : other.boxes_) {
        const double pts[4][2] = {{b.x0, b.y0}, {b.x1, b.y0}, {b.x0, b.y1}, {b.x1, b.y1}};
        float box[4] = {1e30f, 1e30f, -1e30f, -1e30f};
        box_of_points(t, pts, 4, box);
        boxes_.push_back(Box{box[0], box[1], box[2], box[3]});
    }
    max_depth_ = std::max(max_depth_, depth_ - (uint32_t)other.open_.size() + other.max_depth_);
}

void FootprintEstimator::apply_transform(const Transform& t) {
    std::vector<Box> old;
    old.swap(boxes_);
    for (const Box& b : old) {
        const double pts[4][2] = {{b.x0, b.y0}, {b.x1, b.y0}, {b.x0, b.y1}, {b.x1, b.y1}};
        float box[4] = {1e30f, 1e30f, -1e30f, -1e30f};
        box_of_points(t, pts, 4, box);
        boxes_.push_back(Box{box[0], box[1], box[2], box[3]});
    }
}

void FootprintEstimator::tally(uint32_t width, uint32_t height, uint64_t* tiles, uint64_t* bin_elements, uint64_t* ptcl, uint64_t* blend) const {
    const int64_t wt = (width + 15) / 16, ht = (height + 15) / 16, wb = (wt + 15) / 16, hb = (ht + 15) / 16;
    uint64_t nt = 0, nb = 0;
    for (const Box& b : boxes_) {
        auto span = [](float lo, float hi, float unit, int64_t limit) -> int64_t {
            int64_t a = (int64_t)std::floor(std::max(lo, 0.0f) / unit), e = (int64_t)std::ceil(std::min(hi, (float)limit * unit) / unit);
            a = std::min(std::max(a, (int64_t)0), limit);
            e = std::min(std::max(e, (int64_t)0), limit);
            return std::max(e - a, (int64_t)0);
        };
        if (!(b.x0 <= b.x1) || !(b.y0 <= b.y1)) { nt += (uint64_t)(wt * ht); nb += (uint64_t)(wb * hb); continue; }  // NaN: assume everything
        nt += (uint64_t)(span(b.x0, b.x1, 16.0f, wt) * span(b.y0, b.y1, 16.0f, ht));
        nb += (uint64_t)(span(b.x0, b.x1, 256.0f, wb) * span(b.y0, b.y1, 256.0f, hb));
    }
    *tiles = nt;
    *bin_elements = nb;
    // Per (draw object, tile): at most FILL (4 words) + the longest brush command (COLOR, 5) -- or, for a clip pair,
    // BEGIN_CLIP (1) and FILL + END_CLIP (7); each 256-word chunk loses at most 2 words to its JUMP and < 9 to the
    // command that did not fit, and every tile may open one chunk it barely uses.
    *ptcl = nt * 9u + nt * 9u * 11u / 245u + (uint64_t)(wt * ht) * 256u;
    *blend = max_depth_ > 4 ? (uint64_t)(wt * ht) * 256u * (max_depth_ - 4) : 0u;
}

}  // namespace jello
