// scene.h -- the user-facing drawing API.  Mirrors scene.go:20-214 (Scene.Fill / Stroke /
// PushLayer / PopLayer / Append / ApplyTransform / Encoding), including the BumpEstimator side
// channel (scene.go:22,36-43,73,75,115,161,203).  The reference computes that estimate and never
// reads it (renderer/config.go:141-151 hard-codes the sizes); here `bump_sizes` turns it into the
// element counts the renderer allocates, so that the first attempt normally fits
// and the engine's regrow loop is only the safety net.
#pragma once
#include <cstring>
#include <unordered_map>

#include "encoding.h"
#include "estimate.h"

namespace jello {

class Scene {
   public:
    void reset() { encoding_.reset(); estimator_.reset(); footprint_.reset(); image_store_.clear(); }
    Encoding& encoding() { return encoding_; }
    const Encoding& encoding() const { return encoding_; }

    void push_layer(BlendMode blend, float alpha, const Affine& clip_transform, const BezPath& clip);
    void pop_layer();
    void fill(Fill style, const Affine& transform, const Brush& brush, const Affine& brush_transform, const BezPath& path);
    void stroke(const Stroke& style, const Affine& transform, const Brush& brush, const Affine& brush_transform, const BezPath& shape);
    void append(const Scene& other, const Affine& transform) {  // scene.go:200-204
        Transform t = transform.to_transform();
        encoding_.append(other.encoding_, t);
        estimator_.append(other.estimator_, &t);
        footprint_.append(other.footprint_, t);
    }
    void apply_transform(const Affine& transform) {
        Transform t = transform.to_transform();
        encoding_.apply_transform(t);
        BumpEstimator scaled;
        scaled.append(estimator_, &t);
        estimator_ = scaled;
        footprint_.apply_transform(t);
    }

    // scene.go:36-43 bumpEstimate
    BumpEstimate bump_estimate(const Affine* transform = nullptr) const;
    // Element counts for the bump-allocated buffers of a width x height render: lines / seg_counts / segments from the
    // BumpEstimator, tiles / bin_data / ptcl / blend_spill from the draw objects' bounding boxes; never below `floor`.
    BumpSizes bump_sizes(uint32_t width, uint32_t height, uint32_t* clamped = nullptr) const;

    // Pixel copies of the image brushes that entered through the C API, one per (key, contents): thousands of fills
    // with one image share one copy.  Entries live as long as the Scene; the patches hold references of their own.
    std::shared_ptr<const std::vector<uint8_t>> own_pixels(uint64_t key, const uint8_t* px, size_t n) {
        // A key with the top bit set is derived from the contents by the caller (Brush.image's default): equal key and length
        // mean equal pixels, so a fill with an image that is already stored costs a lookup.  Any other key is an identity the
        // caller chose -- its pixels may have changed since, so they are compared (O(pixels) per draw).
        auto it = image_store_.find(key);
        if (it != image_store_.end() && it->second->size() == n && ((key >> 63) != 0u || std::memcmp(it->second->data(), px, n) == 0)) return it->second;
        auto copy = std::make_shared<const std::vector<uint8_t>>(px, px + n);
        image_store_[key] = copy;
        return copy;
    }

   private:
    std::unordered_map<uint64_t, std::shared_ptr<const std::vector<uint8_t>>> image_store_;
    Encoding encoding_;
    BumpEstimator estimator_;
    FootprintEstimator footprint_;
};

}  // namespace jello
