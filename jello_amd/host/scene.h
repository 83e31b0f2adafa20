// scene.h -- the user-facing drawing API.  Mirrors scene.go:20-214 (Scene.Fill / Stroke /
// PushLayer / PopLayer / Append / ApplyTransform / Encoding).  The BumpEstimator side channel
// of the reference (scene.go:22,64-69) is not used to size buffers there either
// (renderer/config.go:141-151); sizing here is done by the engine's regrow loop.
#pragma once
#include "encoding.h"

namespace jello {

class Scene {
   public:
    void reset() { encoding_.reset(); }
    Encoding& encoding() { return encoding_; }
    const Encoding& encoding() const { return encoding_; }

    void push_layer(BlendMode blend, float alpha, const Affine& clip_transform, const BezPath& clip);
    void pop_layer();
    void fill(Fill style, const Affine& transform, const Brush& brush, const Affine& brush_transform, const BezPath& path);
    void stroke(const Stroke& style, const Affine& transform, const Brush& brush, const Affine& brush_transform, const BezPath& shape);
    void append(const Scene& other, const Affine& transform) { encoding_.append(other.encoding_, transform.to_transform()); }
    void apply_transform(const Affine& transform) { encoding_.apply_transform(transform.to_transform()); }

   private:
    Encoding encoding_;
};

}  // namespace jello
