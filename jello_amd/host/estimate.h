// estimate.h -- conservative sizes for the bump-allocated buffers, computed while the scene is built.
//
// BumpEstimator restates renderer/estimate.go:19-406 (itself a port of Vello's bump_estimate.rs): Wang's formula for
// the lines of a flattened curve, arc-length / 16 * sqrt(2) for tile crossings, per-cap and per-join line counts for
// strokes, `Tally` = {lines, seg_counts = segments = binning = max(lines, segments)}.  Two defects of the Go port are
// NOT reproduced (they make its result meaningless, and the reference never consumes it: scene.go:36 `bumpEstimate`
// has no caller and renderer/config.go:141-151 hard-codes the sizes):
//   * estimate.go:81 `t := s.t` shadows the transform parameter with a field that is never assigned, so every length
//     is measured under the zero matrix;
//   * `est.state` (joins, lineToLines, curveLines, curveCount, segments, firstPt, lastPt) is not reset between
//     CountPath calls, so each path re-adds the counts of all paths before it.
// Here the state is local to one count_path call and lengths use the path's transform, as in the Rust original.
//
// The reference leaves binning / ptcl / tile as TODO (estimate.go:20-22, Tally returns 0 for them).  FootprintEstimator
// is this build's own bound for those three: the device-space bounding box of every draw object's control polygon
// (widened by the stroke), counted in 16-px tiles and 256-px bins when the target size is known.
#pragma once
#include <cstdint>
#include <vector>

#include "gfx.h"
#include "jmath.h"

namespace jello {

struct BumpSizes;

struct BumpEstimate {  // renderer.BumpAllocators as element counts (estimate.go:183-196)
    uint32_t binning = 0, ptcl = 0, tile = 0, blend = 0, seg_counts = 0, segments = 0, lines = 0;
};

class BumpEstimator {
   public:
    void reset() { *this = BumpEstimator(); }
    void append(const BumpEstimator& other, const Transform* transform);          // estimate.go:57-62
    void count_path(const BezPath& path, const Transform& t, const Stroke* stroke);  // estimate.go:64-171
    BumpEstimate tally(const Transform* transform) const;                          // estimate.go:173-197

   private:
    struct LineSoup {  // estimate.go:249-275
        uint32_t linetos = 0, curves = 0, curve_count = 0;
        uint32_t scaled_curve_line_count(double scale) const;
        uint32_t tally(double scale) const;
        void add(const LineSoup& other, double scale);
    };
    void count_stroke_caps(Cap style, double scaled_width, uint32_t count);                       // estimate.go:199-213
    void count_stroke_joins(Join style, double scaled_width, double miter_limit, uint32_t count);  // estimate.go:215-235
    uint32_t segments_ = 0;
    LineSoup lines_;
};

// Bounding boxes of the draw objects in scene coordinates, for the allocators the reference's estimator leaves open.
class FootprintEstimator {
   public:
    void reset() { boxes_.clear(); open_.clear(); depth_ = max_depth_ = 0; }
    void add(const BezPath& path, const Transform& t, const Stroke* stroke);
    // push_layer: call after add() of the clip path.  pop_layer re-adds that box: the EndClip draw object is bound to
    // the clip's path (clip_leaf.wgsl) and gets its own tiles and bin entries.
    void push_layer() { depth_++; if (depth_ > max_depth_) max_depth_ = depth_; open_.push_back(boxes_.empty() ? 0 : boxes_.size() - 1); }
    void pop_layer() {
        if (depth_) depth_--;
        if (!open_.empty()) { if (open_.back() < boxes_.size()) boxes_.push_back(boxes_[open_.back()]); open_.pop_back(); }
    }
    void append(const FootprintEstimator& other, const Transform& t);
    void apply_transform(const Transform& t);
    // tiles = sum of bbox tiles (what tile_alloc allocates), bin_elements = sum of bbox bins (what binning writes),
    // ptcl = dynamic PTCL words beyond the per-tile heads, blend = spill pixels for clip depths beyond the 4 in registers
    void tally(uint32_t width, uint32_t height, uint64_t* tiles, uint64_t* bin_elements, uint64_t* ptcl, uint64_t* blend) const;

   private:
    struct Box { float x0, y0, x1, y1; };
    std::vector<Box> boxes_;
    std::vector<size_t> open_;  // boxes of the open layers' clip paths
    uint32_t depth_ = 0, max_depth_ = 0;
};

}  // namespace jello
