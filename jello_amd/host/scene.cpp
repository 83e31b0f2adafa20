// scene.cpp -- see scene.h.  Statement order follows scene.go:40-198 so that the tag/transform/
// style de-duplication produces the same streams.
#include "scene.h"

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
    }
    encoding_.encode_begin_clip(blend, std::min(std::max(alpha, 0.0f), 1.0f));
}

void Scene::pop_layer() { encoding_.encode_end_clip(); }  // scene.go:73-79

void Scene::fill(Fill style, const Affine& transform, const Brush& brush, const Affine& brush_transform, const BezPath& path) {  // scene.go:81-110
    Transform t = transform.to_transform();
    encoding_.encode_transform(t);
    encoding_.encode_fill_style(style);
    if (encoding_.encode_path(path, true)) {
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
    bool encode_result = encoding_.encode_path(shape, false);
    if (encode_result) {
        if (!brush_transform.is_identity()) {
            if (encoding_.encode_transform(transform.mul(brush_transform).to_transform())) encoding_.swap_last_path_tags();
        }
        encoding_.encode_brush(brush, 1.0f);
    }
}

}  // namespace jello
