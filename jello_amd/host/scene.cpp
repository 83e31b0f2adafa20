// scene.cpp -- see scene.h.  Statement order follows scene.go:40-198 so that the tag/transform/
// style de-duplication produces the same streams.
#include "scene.h"

#include "renderer.h"

#include <algorithm>
#include <stdexcept>

namespace jello {

static BezPath zero_rect_path() {  // curve.Rect{}.PathElements(0.1)
    BezPath p;
    p.push_back(PathEl{PathElKind::MoveTo, {0, 0}, {0, 0}, {0, 0}});
    p.push_back(PathEl{PathElKind::LineTo, {0, 0}, {0, 0}, {0, 0}});
    p.push_back(PathEl{PathElKind::LineTo, {0, 0}, {0, 0}, {0, 0}});
    p.push_back(PathEl{PathElKind::LineTo, {0, 0}, {0, 0}, {0, 0}});
    p.push_back(PathEl{PathElKind::ClosePath, {0, 0}, {0, 0}, {0, 0}});
    return p;
}

void Scene::push_layer(BlendMode blend, float alpha, const Affine& clip_transform, const BezPath& clip) {  // scene.go:40-71
    Transform t = clip_transform.to_transform();
    encoding_.encode_transform(t);
    encoding_.encode_fill_style(Fill::NonZero);
    if (!encoding_.encode_path(clip, true)) {
        // Invalid layer shape: encode a valid empty path, which suppresses drawing until the pop.
        encoding_.encode_path(zero_rect_path(), true);
        encoding_.encode_empty_shape();
        BezPath degenerate;  // scene.go:69-73
        degenerate.push_back(PathEl{PathElKind::MoveTo, {0, 0}, {0, 0}, {0, 0}});
        degenerate.push_back(PathEl{PathElKind::LineTo, {0, 0}, {0, 0}, {0, 0}});
        estimator_.count_path(degenerate, t, nullptr);
        footprint_.add(zero_rect_path(), t, nullptr);
    } else {
        estimator_.count_path(clip, t, nullptr);
        footprint_.add(clip, t, nullptr);
    }
    footprint_.push_layer();
    encoding_.encode_begin_clip(blend, std::min(std::max(alpha, 0.0f), 1.0f));
}

void Scene::pop_layer() {  // scene.go:73-79
    encoding_.encode_end_clip();
    footprint_.pop_layer();
}

BumpEstimate Scene::bump_estimate(const Affine* transform) const {
    if (!transform) return estimator_.tally(nullptr);
    Transform t = transform->to_transform();
    return estimator_.tally(&t);
}

BumpSizes Scene::bump_sizes(uint32_t width, uint32_t height, uint32_t* clamped) const {
    BumpEstimate e = estimator_.tally(nullptr);
    uint64_t tiles = 0, bins = 0, ptcl = 0, blend = 0;
    footprint_.tally(width, height, &tiles, &bins, &ptcl, &blend);
    uint64_t info = 0;  // info words precede the bin data in the same buffer (resolve.go:271-276)
    for (uint32_t tag : encoding_.draw_tags) info += (tag >> 6) & 0xf;
    const uint64_t wt = (width + 15) / 16, ht = (height + 15) / 16;
    // The bounds are conservative (a path counts for every tile of its bounding box), so heavy overdraw or huge boxes can
    // put them at many GiB although the frame would fit the reference's fixed sizes: the FIRST attempt is held to 16 x the
    // reference constants (renderer/config.go:144-151; C3 needs 2.2 x at most) -- plus what cannot be less: the info words
    // and the tiles' static PTCL heads -- and the regrow loop (hip_engine.cpp) covers a frame that really needs more.
    // `clamped` (optional) receives one bit per size that was held below what the bounds ask for (bit order: bin_data, tiles,
    // lines, seg_counts, segments, blend_spill, ptcl): a caller that renders WITHOUT the regrow loop -- hipGraph capture, a
    // timed loop -- can see that the sizes may not do and run one robust render first.
    uint32_t held = 0u;
    auto cap = [&held](uint64_t v, uint64_t ref, uint64_t floor_, uint32_t bit) {
        const uint64_t want = std::max<uint64_t>(v + v / 8 + 1024, 4096);
        const uint64_t got = std::min<uint64_t>(std::min<uint64_t>(want, std::max<uint64_t>(16u * ref, floor_ + ref)), 0xfffffff0ull);
        if (got < want) held |= 1u << bit;
        return (uint32_t)got;
    };
    BumpSizes b;
    b.bin_data = cap(info + bins, 1u << 18, info, 0);
    b.tiles = cap(tiles, 1u << 21, 0, 1);
    b.lines = cap(e.lines, 1u << 21, 0, 2);
    b.seg_counts = cap(e.seg_counts, 1u << 21, 0, 3);
    b.segments = cap(e.segments, 1u << 21, 0, 4);
    b.blend_spill = cap(blend, 1u << 21, 0, 5);
    b.ptcl = cap(ptcl + wt * ht * JL_PTCL_INITIAL_ALLOC, 1u << 23, wt * ht * JL_PTCL_INITIAL_ALLOC, 6);
    if (clamped) *clamped = held;
    return b;
}

void Scene::fill(Fill style, const Affine& transform, const Brush& brush, const Affine& brush_transform, const BezPath& path) {  // scene.go:81-110
    Transform t = transform.to_transform();
    encoding_.encode_transform(t);
    encoding_.encode_fill_style(style);
    if (encoding_.encode_path(path, true)) {
        estimator_.count_path(path, t, nullptr);
        footprint_.add(path, t, nullptr);
        if (!brush_transform.is_identity()) {
            if (encoding_.encode_transform(transform.mul(brush_transform).to_transform())) encoding_.swap_last_path_tags();
        }
        encoding_.encode_brush(brush, 1.0f);
    }
}

void Scene::stroke(const Stroke& style, const Affine& transform, const Brush& brush, const Affine& brush_transform, const BezPath& shape) {  // scene.go:112-198
    if (!style.dash_pattern.empty())
        throw std::invalid_argument("dashed strokes are expanded by the third-party curve.Dash in the reference; not supported here");
    Transform t = transform.to_transform();
    encoding_.encode_transform(t);
    encoding_.encode_stroke_style(style);
    estimator_.count_path(shape, t, &style);  // scene.go:161 (counted whether or not the path encodes)
    bool encode_result = encoding_.encode_path(shape, false);
    if (encode_result) {
        footprint_.add(shape, t, &style);
        if (!brush_transform.is_identity()) {
            if (encoding_.encode_transform(transform.mul(brush_transform).to_transform())) encoding_.swap_last_path_tags();
        }
        encoding_.encode_brush(brush, 1.0f);
    }
}

}  // namespace jello
