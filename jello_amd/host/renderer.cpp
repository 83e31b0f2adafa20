// renderer.cpp -- see renderer.h.
#include "renderer.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace jello {

// ------------------------------------------------------------------------------------------------
// recording.go
// ------------------------------------------------------------------------------------------------
static std::atomic<uint64_t> g_resource_id{0};
ResourceID next_resource_id() { return g_resource_id.fetch_add(1) + 1; }

BufferProxy new_buffer_proxy(uint64_t size, const std::string& name) { return BufferProxy{size, next_resource_id(), name}; }
ImageProxy new_image_proxy(uint32_t width, uint32_t height, JlImageFormat format) {
    ImageProxy p;
    p.width = width; p.height = height; p.format = format; p.id = next_resource_id();
    return p;
}

BufferProxy Recording::upload(const std::string& name, const void* data, size_t n) {
    Command c;
    c.kind = Command::Upload;
    c.buffer = new_buffer_proxy(n, name);
    c.data.assign((const uint8_t*)data, (const uint8_t*)data + n);
    commands.push_back(std::move(c));
    return commands.back().buffer;
}
BufferProxy Recording::upload_uniform(const std::string& name, const void* data, size_t n) {
    Command c;
    c.kind = Command::UploadUniform;
    c.buffer = new_buffer_proxy(n, name);
    c.data.assign((const uint8_t*)data, (const uint8_t*)data + n);
    commands.push_back(std::move(c));
    return commands.back().buffer;
}
ImageProxy Recording::upload_image(uint32_t w, uint32_t h, JlImageFormat format, const void* data, size_t n) {
    Command c;
    c.kind = Command::UploadImage;
    c.image = new_image_proxy(w, h, format);
    c.data.assign((const uint8_t*)data, (const uint8_t*)data + n);
    commands.push_back(std::move(c));
    return commands.back().image;
}
void Recording::write_image(const ImageProxy& img, uint32_t x, uint32_t y, uint32_t w, uint32_t h, const void* data, size_t n) {
    Command c;
    c.kind = Command::WriteImage;
    c.image = img;
    c.coords[0] = x; c.coords[1] = y; c.coords[2] = w; c.coords[3] = h;
    c.data.assign((const uint8_t*)data, (const uint8_t*)data + n);
    commands.push_back(std::move(c));
}
void Recording::dispatch(ShaderID shader, const uint32_t wg[3], std::vector<ResourceProxy> resources) {
    Command c;
    c.kind = Command::Dispatch;
    c.shader = shader;
    std::memcpy(c.wg_count, wg, sizeof c.wg_count);
    c.bindings = std::move(resources);
    commands.push_back(std::move(c));
}
void Recording::dispatch_indirect(ShaderID shader, const BufferProxy& buf, uint64_t offset, std::vector<ResourceProxy> resources) {
    Command c;
    c.kind = Command::DispatchIndirect;
    c.shader = shader;
    c.buffer = buf;
    c.offset = offset;
    c.bindings = std::move(resources);
    commands.push_back(std::move(c));
}
void Recording::download(const BufferProxy& buf) { Command c; c.kind = Command::Download; c.buffer = buf; commands.push_back(std::move(c)); }
void Recording::clear_all(const BufferProxy& buf) {
    Command c;
    c.kind = Command::Clear;
    c.buffer = buf;
    c.offset = 0;
    c.size = -1;
    commands.push_back(std::move(c));
}
void Recording::free_buffer(const BufferProxy& buf) { Command c; c.kind = Command::FreeBuffer; c.buffer = buf; commands.push_back(std::move(c)); }
void Recording::free_image(const ImageProxy& img) { Command c; c.kind = Command::FreeImage; c.image = img; commands.push_back(std::move(c)); }
void Recording::free_resource(const ResourceProxy& r) {
    switch (r.kind) {
        case ResourceProxy::Buffer: free_buffer(r.buffer); break;
        case ResourceProxy::Image: free_image(r.image); break;
        default: throw std::logic_error("free_resource: unhandled resource kind");
    }
}

// ------------------------------------------------------------------------------------------------
// config.go
// ------------------------------------------------------------------------------------------------
static const uint32_t kPathReduceWg = 256, kPathBboxWg = 256, kFlattenWg = 256, kClipReduceWg = 256, kTileW = 16, kTileH = 16;

static void set3(uint32_t* d, uint32_t x, uint32_t y = 1, uint32_t z = 1) { d[0] = x; d[1] = y; d[2] = z; }

WorkgroupCounts new_workgroup_counts(const JlLayout& layout, uint32_t width_in_tiles, uint32_t height_in_tiles, uint32_t num_path_tags) {  // config.go:181-231
    uint32_t num_paths = layout.n_path, num_draw_objects = layout.n_drawobj, num_clips = layout.n_clip;
    uint32_t path_tag_padded = align_up(num_path_tags, 4 * kPathReduceWg);
    uint32_t path_tag_wgs = path_tag_padded / (4 * kPathReduceWg);
    bool use_large = path_tag_wgs > kPathReduceWg;
    uint32_t reduced_size = use_large ? align_up(path_tag_wgs, kPathReduceWg) : path_tag_wgs;
    uint32_t draw_object_wgs = (num_draw_objects + kPathBboxWg - 1) / kPathBboxWg;
    uint32_t draw_monoid_wgs = std::min(draw_object_wgs, kPathBboxWg);
    uint32_t flatten_wgs = (num_path_tags + kFlattenWg - 1) / kFlattenWg;
    uint32_t num_clips_minus_one = num_clips > 0 ? num_clips - 1 : 0;
    uint32_t clip_reduce_wgs = num_clips_minus_one / kClipReduceWg;
    uint32_t clip_wgs = (num_clips + kClipReduceWg - 1) / kClipReduceWg;
    uint32_t path_wgs = (num_paths + kPathBboxWg - 1) / kPathBboxWg;
    uint32_t width_in_bins = (width_in_tiles + 15) / 16, height_in_bins = (height_in_tiles + 15) / 16;
    WorkgroupCounts w;
    w.use_large_path_scan = use_large;
    set3(w.path_reduce, path_tag_wgs);
    set3(w.path_reduce2, kPathReduceWg);
    set3(w.path_scan1, reduced_size / kPathReduceWg);
    set3(w.path_scan, path_tag_wgs);
    set3(w.bbox_clear, draw_object_wgs);
    set3(w.flatten, flatten_wgs);
    set3(w.draw_reduce, draw_monoid_wgs);
    set3(w.draw_leaf, draw_monoid_wgs);
    set3(w.clip_reduce, clip_reduce_wgs);
    set3(w.clip_leaf, clip_wgs);
    set3(w.binning, draw_object_wgs);
    set3(w.tile_alloc, path_wgs);
    set3(w.path_count_setup, 1);
    set3(w.backdrop, path_wgs);
    set3(w.coarse, width_in_bins, height_in_bins);
    set3(w.path_tiling_setup, 1);
    set3(w.fine, width_in_tiles, height_in_tiles);
    return w;
}

static uint64_t bs(uint64_t n, uint64_t elem) { return std::max<uint64_t>(n, 1) * elem; }  // NewBufferSize + sizeInBytes

BufferSizes new_buffer_sizes(const JlLayout& layout, const WorkgroupCounts& wg, const BumpSizes& bump) {  // config.go:125-179
    uint32_t num_paths = layout.n_path, num_draw_objects = layout.n_drawobj, num_clips = layout.n_clip;
    uint32_t path_tag_wgs = wg.path_reduce[0];
    uint32_t reduced_size = wg.use_large_path_scan ? align_up(path_tag_wgs, kPathReduceWg) : path_tag_wgs;
    uint32_t draw_monoid_wgs = wg.draw_reduce[0];
    uint32_t binning_wgs = wg.binning[0];
    uint32_t num_paths_aligned = align_up(num_paths, 256u);
    BufferSizes b;
    b.path_reduced = bs(reduced_size, sizeof(JlTagMonoid));
    b.path_reduced2 = bs(kPathReduceWg, sizeof(JlTagMonoid));
    b.path_reduced_scan = bs(reduced_size, sizeof(JlTagMonoid));
    b.path_monoids = bs((uint64_t)path_tag_wgs * kPathReduceWg, sizeof(JlTagMonoid));
    b.path_bboxes = bs(num_paths, sizeof(JlPathBbox));
    b.draw_reduced = bs(draw_monoid_wgs, sizeof(JlDrawMonoid));
    b.draw_monoids = bs(num_draw_objects, sizeof(JlDrawMonoid));
    b.info = bs(layout.bin_data_start, 4);
    b.clip_inps = bs(num_clips, sizeof(JlClipInp));
    b.clip_els = bs(num_clips, sizeof(JlClipEl));
    b.clip_bics = bs(num_clips / kClipReduceWg, sizeof(JlClipBic));
    b.clip_bboxes = bs(num_clips, 16);
    b.draw_bboxes = bs(num_paths, 16);
    b.bump_alloc = bs(1, sizeof(JlBump));
    b.indirect_count = bs(1, sizeof(JlIndirectCount));
    b.bin_headers = bs((uint64_t)binning_wgs * 256, sizeof(JlBinHeader));
    b.paths = bs(num_paths_aligned, sizeof(JlPath));
    b.lines = bs(bump.lines, sizeof(JlLineSoup));
    b.bin_data = bs(bump.bin_data, 4);
    b.tiles = bs(bump.tiles, sizeof(JlTile));
    b.seg_counts = bs(bump.seg_counts, sizeof(JlSegmentCount));
    b.segments = bs(bump.segments, sizeof(JlSegment));
    b.blend_spill = bs(bump.blend_spill, 16);
    b.ptcl = bs(bump.ptcl, 4);
    return b;
}

static uint32_t next_multiple_of(uint32_t x, uint32_t y) { uint32_t r = x % y; return r == 0 ? x : x + y - r; }

RenderConfig new_render_config(const JlLayout& layout, uint32_t width, uint32_t height, const Color& base_color, const BumpSizes& bump) {  // config.go:94-123
    uint32_t new_width = next_multiple_of(width, kTileW), new_height = next_multiple_of(height, kTileH);
    uint32_t width_in_tiles = new_width / kTileW, height_in_tiles = new_height / kTileH;
    uint32_t num_path_tags = (layout.pathdata_base - layout.pathtag_base) * 4;  // Layout.pathTagsSize, config.go:82-86
    RenderConfig rc;
    rc.workgroup_counts = new_workgroup_counts(layout, width_in_tiles, height_in_tiles, num_path_tags);
    rc.buffer_sizes = new_buffer_sizes(layout, rc.workgroup_counts, bump);
    JlConfig& g = rc.gpu;
    std::memset(&g, 0, sizeof g);
    g.width_in_tiles = width_in_tiles;
    g.height_in_tiles = height_in_tiles;
    g.target_width = width;
    g.target_height = height;
    premul32(base_color, g.base_color);
    g.layout = layout;
    g.lines_size = (uint32_t)(rc.buffer_sizes.lines / sizeof(JlLineSoup));
    g.binning_size = (uint32_t)(rc.buffer_sizes.bin_data / 4) - layout.bin_data_start;
    g.tiles_size = (uint32_t)(rc.buffer_sizes.tiles / sizeof(JlTile));
    g.seg_counts_size = (uint32_t)(rc.buffer_sizes.seg_counts / sizeof(JlSegmentCount));
    g.segments_size = (uint32_t)(rc.buffer_sizes.segments / sizeof(JlSegment));
    g.blend_size = (uint32_t)(rc.buffer_sizes.blend_spill / 16);
    g.ptcl_size = (uint32_t)(rc.buffer_sizes.ptcl / 4);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// ramp_cache.go.  color.Step (honnef.co/go/color, un-vendored) interpolates in sRGB and converts
// to linear sRGB; its sampling positions are not visible from the reference tree.  Own definition:
// n samples at t = i/(n-1), interpolated per channel in gamma-encoded sRGB (alpha linearly),
// decoded to linear, premultiplied, stored as RTNE binary16.  PARITY UNPINNED for ramp texels.
// ------------------------------------------------------------------------------------------------
static const uint32_t kNumSamples = 512, kRetainedCount = 64;

static double srgb_encode(double l) { return l <= 0.0031308 ? 12.92 * l : 1.055 * std::pow(l, 1.0 / 2.4) - 0.055; }
static double srgb_decode(double s) { return s <= 0.04045 ? s / 12.92 : std::pow((s + 0.055) / 1.055, 2.4); }

static void premul16(const Color& c, uint16_t out[4]) {  // gfx/color.go:11-25 (uses jmath.Float16bits)
    out[0] = float16_bits((float)(c.r * c.a));
    out[1] = float16_bits((float)(c.g * c.a));
    out[2] = float16_bits((float)(c.b * c.a));
    out[3] = float16_bits((float)c.a);
}

static std::vector<uint16_t> make_ramp(const ColorStop* stops_in, size_t n_in) {  // ramp_cache.go:219-261
    if (n_in < 2) throw std::logic_error("make_ramp needs at least two stops");
    std::vector<ColorStop> stops(stops_in, stops_in + n_in);
    if (stops[0].offset != 0) {
        ColorStop first = stops[0];
        first.offset = 0;
        stops.insert(stops.begin(), first);
    }
    std::vector<uint16_t> out;
    out.reserve(kNumSamples * 4);
    int remaining = (int)kNumSamples;
    for (size_t i = 1; i < stops.size(); i++) {
        const ColorStop& prev = stops[i - 1];
        const ColorStop& stop = stops[i];
        int n;
        if (i == stops.size() - 1) {
            n = remaining;
        } else {
            float frac = stop.offset - prev.offset;
            n = (int)std::round((float)kNumSamples * frac);  // jmath.Round32 = ties away from zero
            n = std::min(remaining, n);
        }
        remaining -= n;
        if (n == 0) continue;
        if (n == 1) {
            uint16_t t[4];
            premul16(stop.color, t);
            out.insert(out.end(), t, t + 4);
            continue;
        }
        double a0[3] = {srgb_encode(prev.color.r), srgb_encode(prev.color.g), srgb_encode(prev.color.b)};
        double a1[3] = {srgb_encode(stop.color.r), srgb_encode(stop.color.g), srgb_encode(stop.color.b)};
        for (int s = 0; s < n; s++) {
            double t = (double)s / (double)(n - 1);
            Color c;
            c.r = srgb_decode(a0[0] + (a1[0] - a0[0]) * t);
            c.g = srgb_decode(a0[1] + (a1[1] - a0[1]) * t);
            c.b = srgb_decode(a0[2] + (a1[2] - a0[2]) * t);
            c.a = prev.color.a + (stop.color.a - prev.color.a) * t;
            uint16_t tx[4];
            premul16(c, tx);
            out.insert(out.end(), tx, tx + 4);
        }
    }
    if (out.size() != kNumSamples * 4) throw std::logic_error("make_ramp: wrong sample count");
    return out;
}

void Resolver::ramp_maintain() {  // ramp_cache.go:42-52
    epoch_++;
    if (mapping_.size() > kRetainedCount) {
        for (auto it = mapping_.begin(); it != mapping_.end();) {
            if (it->second.id >= kRetainedCount) it = mapping_.erase(it); else ++it;
        }
        ramp_data_.resize((size_t)kRetainedCount * kNumSamples * 4);
    }
}

uint32_t Resolver::ramp_add(const ColorStop* stops, size_t n) {  // ramp_cache.go:54-109
    std::string key;
    uint64_t len = n;
    key.append((const char*)&len, 8);
    for (size_t i = 0; i < n; i++) {
        key.append((const char*)&stops[i].offset, 4);
        key.append((const char*)&stops[i].color.r, 8);
        key.append((const char*)&stops[i].color.g, 8);
        key.append((const char*)&stops[i].color.b, 8);
        key.append((const char*)&stops[i].color.a, 8);
    }
    auto it = mapping_.find(key);
    if (it != mapping_.end()) {
        it->second.epoch = epoch_;
        return it->second.id;
    }
    if (mapping_.size() < kRetainedCount) {
        uint32_t id = (uint32_t)(ramp_data_.size() / (kNumSamples * 4));
        std::vector<uint16_t> r = make_ramp(stops, n);
        ramp_data_.insert(ramp_data_.end(), r.begin(), r.end());
        mapping_[key] = RampEntry{id, epoch_};
        return id;
    }
    for (auto jt = mapping_.begin(); jt != mapping_.end(); ++jt) {
        if (jt->second.epoch + 2 < epoch_) {
            uint32_t reuse = jt->second.id;
            mapping_.erase(jt);
            std::vector<uint16_t> r = make_ramp(stops, n);
            std::copy(r.begin(), r.end(), ramp_data_.begin() + (size_t)reuse * kNumSamples * 4);
            mapping_[key] = RampEntry{reuse, epoch_};
            return reuse;
        }
    }
    uint32_t id = (uint32_t)(ramp_data_.size() / (kNumSamples * 4));
    std::vector<uint16_t> r = make_ramp(stops, n);
    ramp_data_.insert(ramp_data_.end(), r.begin(), r.end());
    return id;
}

// ------------------------------------------------------------------------------------------------
// resolve.go:64-283.  One packer serves both the patched and the solid-only case; the streams and
// Layout are identical to resolveSolidPathsOnly when there are no patches.
// ------------------------------------------------------------------------------------------------
Resolver::Resolved Resolver::resolve(const Encoding& enc) {
    Resolved out;
    std::memset(&out.layout, 0, sizeof out.layout);
    struct RP { int kind; int off; uint32_t word; };
    std::vector<RP> patches;
    std::map<uint64_t, uint32_t> image_ix;
    if (!enc.resources.patches.empty()) {
        ramp_maintain();
        for (const Patch& p : enc.resources.patches) {
            if (p.kind == Patch::Ramp) {
                uint32_t ramp_id = ramp_add(enc.resources.color_stops.data() + p.ramp.stops[0], (size_t)(p.ramp.stops[1] - p.ramp.stops[0]));
                patches.push_back(RP{0, p.ramp.draw_data_offset, (ramp_id << 2) | (uint32_t)p.ramp.extend});
            } else {
                uint32_t idx;
                auto it = image_ix.find(p.image.image.key);
                if (it != image_ix.end()) {
                    idx = it->second;
                } else {
                    idx = (uint32_t)out.images.size();
                    out.images.push_back(p.image.image);
                    image_ix[p.image.image.key] = idx;
                }
                patches.push_back(RP{1, p.image.draw_data_offset, idx});
            }
        }
    }
    JlLayout& layout = out.layout;
    layout.n_path = enc.num_paths;
    layout.n_clip = enc.num_clips;
    size_t num_path_tags = enc.path_tags.size() + enc.num_open_clips;
    size_t path_tag_padded = align_up<size_t>(num_path_tags, 4 * kPathReduceWg);
    size_t buffer_size = path_tag_padded + enc.path_data.size() + (enc.draw_tags.size() + enc.num_open_clips) * 4 + enc.draw_data.size() +
                         enc.transforms.size() * sizeof(Transform) + enc.styles.size() * sizeof(Style);
    std::vector<uint8_t>& data = out.packed;
    data.reserve(buffer_size);
    auto words = [&]() { return (uint32_t)(data.size() / 4); };
    // Path tag stream
    layout.pathtag_base = words();
    data.insert(data.end(), enc.path_tags.begin(), enc.path_tags.end());
    for (uint32_t i = 0; i < enc.num_open_clips; i++) data.push_back(JL_PATH_TAG_PATH);
    data.resize(path_tag_padded, 0);
    // Path data stream
    layout.pathdata_base = words();
    data.insert(data.end(), enc.path_data.begin(), enc.path_data.end());
    // Draw tag stream; bin data follows draw info
    layout.drawtag_base = words();
    for (uint32_t tag : enc.draw_tags) layout.bin_data_start += (tag >> 6) & 0xf;
    {
        const uint8_t* p = (const uint8_t*)enc.draw_tags.data();
        data.insert(data.end(), p, p + enc.draw_tags.size() * 4);
        for (uint32_t i = 0; i < enc.num_open_clips; i++) {
            uint32_t t = JL_DRAWTAG_END_CLIP;
            const uint8_t* q = (const uint8_t*)&t;
            data.insert(data.end(), q, q + 4);
        }
    }
    // Draw data stream, with ramp ids / image indices patched in
    layout.drawdata_base = words();
    {
        size_t pos = 0;
        for (const RP& rp : patches) {
            if (pos < (size_t)rp.off) data.insert(data.end(), enc.draw_data.begin() + pos, enc.draw_data.begin() + rp.off);
            const uint8_t* q = (const uint8_t*)&rp.word;
            data.insert(data.end(), q, q + 4);
            pos = (size_t)rp.off + 4;
        }
        if (pos < enc.draw_data.size()) data.insert(data.end(), enc.draw_data.begin() + pos, enc.draw_data.end());
    }
    // Transform stream
    layout.transform_base = words();
    {
        const uint8_t* p = (const uint8_t*)enc.transforms.data();
        data.insert(data.end(), p, p + enc.transforms.size() * sizeof(Transform));
    }
    // Style stream
    layout.style_base = words();
    {
        const uint8_t* p = (const uint8_t*)enc.styles.data();
        data.insert(data.end(), p, p + enc.styles.size() * sizeof(Style));
    }
    layout.n_drawobj = layout.n_path;
    if (buffer_size != data.size()) throw std::logic_error("resolve: buffer size mismatch");
    if (!enc.resources.patches.empty()) {
        out.ramps.data = ramp_data_;
        out.ramps.width = kNumSamples;
        out.ramps.height = (uint32_t)(ramp_data_.size() / (kNumSamples * 4));
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// mask.go:43-105
// ------------------------------------------------------------------------------------------------
static const uint8_t kMaskPattern8[8] = {0, 5, 3, 7, 1, 4, 6, 2};
static const uint8_t kMaskPattern16[16] = {1, 8, 4, 11, 15, 7, 3, 12, 0, 9, 5, 13, 2, 10, 6, 14};

static uint32_t one_mask(double slope, double translation, bool is_pos, const uint8_t* pattern, int n, double step) {
    if (is_pos) translation = 1. - translation;
    uint32_t result = 0;
    for (int i = 0; i < n; i++) {
        double y = ((double)i + 0.5) * step;
        double x = ((double)pattern[i] + 0.5) * step;
        if (!is_pos) y = 1. - y;
        if ((x - (1.0 - translation)) * (1. - slope) - (y - translation) * slope >= 0.) result |= 1u << i;
    }
    return result;
}
std::vector<uint8_t> make_mask_lut8() {
    std::vector<uint8_t> out;
    const int W = 32, H = 32, half = H / 2;
    for (int i = 0; i < W * H; i++) {
        int u = i % W, v = i / W;
        bool is_pos = v >= half;
        double y = ((double)(v % half) + 0.5) * (1.0 / (double)half);
        double x = ((double)u + 0.5) * (1.0 / (double)W);
        out.push_back((uint8_t)one_mask(y, x, is_pos, kMaskPattern8, 8, 0.125));
    }
    return out;
}
std::vector<uint8_t> make_mask_lut16() {
    std::vector<uint8_t> out;
    const int W = 64, H = 64, half = H / 2;
    for (int i = 0; i < W * H; i++) {
        int u = i % W, v = i / W;
        bool is_pos = v >= half;
        double y = ((double)(v % half) + 0.5) * (1.0 / (double)half);
        double x = ((double)u + 0.5) * (1.0 / (double)W);
        uint32_t m = one_mask(y, x, is_pos, kMaskPattern16, 16, 0.0625);
        out.push_back((uint8_t)(m & 0xff));
        out.push_back((uint8_t)(m >> 8));
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// render.go:81-588
// ------------------------------------------------------------------------------------------------
Renderer::Result Renderer::render_full(const Encoding& enc, Resolver& resolver, const FullShaders& shaders, const RenderParams& params,
                                       bool robust) {
    Result res;
    Recording& recording = res.recording;
    last_buffers.clear();
    {   // the deepest nesting of BEGIN_CLIP ... END_CLIP in the draw tag stream
        uint32_t depth = 0, deepest = 0;
        for (uint32_t tag : enc.draw_tags) {
            if (tag == JL_DRAWTAG_BEGIN_CLIP) { depth++; if (depth > deepest) deepest = depth; }
            else if (tag == JL_DRAWTAG_END_CLIP && depth > 0) depth--;
        }
        recording.max_clip_depth = deepest;
    }
    Resolver::Resolved rs = resolver.resolve(enc);
    const JlLayout& layout = rs.layout;
    ImageProxy gradient_image;
    if (rs.ramps.height == 0) {
        gradient_image = new_image_proxy(1, 1, JL_RGBA16_FLOAT);
    } else {
        gradient_image = recording.upload_image(rs.ramps.width, rs.ramps.height, JL_RGBA16_FLOAT, rs.ramps.data.data(), rs.ramps.data.size() * 2);
    }
    std::vector<ImageProxy> image_proxies;
    if (rs.images.empty()) {
        if (empty_.width == 0) {
            const uint8_t zero[4] = {0, 0, 0, 0};
            empty_ = recording.upload_image(1, 1, JL_RGBA8, zero, 4);
        }
        image_proxies.push_back(empty_);
    }
    for (const Image& img : rs.images) {
        auto it = images_.find(img.key);
        if (it != images_.end()) {
            image_proxies.push_back(it->second);
        } else {
            ImageProxy proxy = recording.upload_image(img.width, img.height, JL_RGBA8_SRGB, img.pixels, (size_t)img.width * img.height * 4);
            image_proxies.push_back(proxy);
            images_[img.key] = proxy;
        }
    }
    {   // binning.wgsl:52,131 / coarse.wgsl: a bin is an index into a 256-entry table, so bins past the 256th are lost without
        // a trace in the reference (a wrong frame); the recording is refused here instead
        const uint64_t wb = ((uint64_t)params.width + 255u) / 256u, hb = ((uint64_t)params.height + 255u) / 256u;
        if (wb * hb > 256u)
            throw std::invalid_argument("render: a " + std::to_string(params.width) + " x " + std::to_string(params.height) + " target has " +
                                        std::to_string(wb * hb) + " bins of 256 x 256 px; binning and coarse address 256 (at most 4096 x 4096)");
    }
    res.config = new_render_config(layout, params.width, params.height, params.base_color, params.bump_sizes);
    const BufferSizes& sizes = res.config.buffer_sizes;
    const WorkgroupCounts& wg = res.config.workgroup_counts;
    std::vector<uint8_t> packed = std::move(rs.packed);
    if (packed.empty()) packed.assign(4, 0);
    auto R = [](const BufferProxy& b) { return ResourceProxy::of(b); };
    auto named = [&](uint64_t size, const char* name) {
        BufferProxy b = new_buffer_proxy(size, name);
        last_buffers[name] = b;
        return b;
    };
    BufferProxy scene_buf = recording.upload("scene", packed.data(), packed.size());
    last_buffers["scene"] = scene_buf;
    BufferProxy config_buf = recording.upload_uniform("config", &res.config.gpu, sizeof(JlConfig));
    last_buffers["config"] = config_buf;
    BufferProxy info_bin_data_buf = named(sizes.bin_data, "infoBinDataBuf");
    BufferProxy tile_buf = named(sizes.tiles, "tileBuf");
    BufferProxy segments_buf = named(sizes.segments, "segmentsBuf");
    BufferProxy ptcl_buf = named(sizes.ptcl, "ptclBuf");
    BufferProxy reduced_buf = named(sizes.path_reduced, "reducedBuf");
    recording.dispatch(shaders.pathtag_reduce, wg.path_reduce, {R(config_buf), R(scene_buf), R(reduced_buf)});
    BufferProxy pathtag_parent = reduced_buf;
    bool use_large_path_scan = wg.use_large_path_scan && !shaders.pathtag_is_cpu;
    BufferProxy reduced2_buf, reduced_scan_buf;
    if (use_large_path_scan) {
        reduced2_buf = named(sizes.path_reduced2, "reduced2Buf");
        recording.dispatch(shaders.pathtag_reduce2, wg.path_reduce2, {R(reduced_buf), R(reduced2_buf)});
        reduced_scan_buf = named(sizes.path_reduced_scan, "reducedScanBuf");
        recording.dispatch(shaders.pathtag_scan1, wg.path_scan1, {R(reduced_buf), R(reduced2_buf), R(reduced_scan_buf)});
        pathtag_parent = reduced_scan_buf;
    }
    BufferProxy tagmonoid_buf = named(sizes.path_monoids, "tagmonoidBuf");
    ShaderID pathtag_scan = use_large_path_scan ? shaders.pathtag_scan_large : shaders.pathtag_scan_small;
    recording.dispatch(pathtag_scan, wg.path_scan, {R(config_buf), R(scene_buf), R(pathtag_parent), R(tagmonoid_buf)});
    recording.free_buffer(reduced_buf);
    if (use_large_path_scan) {
        recording.free_buffer(reduced2_buf);
        recording.free_buffer(reduced_scan_buf);
    }
    BufferProxy path_bbox_buf = named(sizes.path_bboxes, "pathBboxBuf");
    recording.dispatch(shaders.bbox_clear, wg.bbox_clear, {R(config_buf), R(path_bbox_buf)});
    BufferProxy bump_buf = named(sizes.bump_alloc, "bumpBuf");
    recording.clear_all(bump_buf);
    BufferProxy lines_buf = named(sizes.lines, "linesBuf");
    recording.dispatch(shaders.flatten, wg.flatten,
                       {R(config_buf), R(scene_buf), R(tagmonoid_buf), R(path_bbox_buf), R(bump_buf), R(lines_buf)});
    BufferProxy draw_reduced_buf = named(sizes.draw_reduced, "drawReducedBuf");
    recording.dispatch(shaders.draw_reduce, wg.draw_reduce, {R(config_buf), R(scene_buf), R(draw_reduced_buf)});
    BufferProxy draw_monoid_buf = named(sizes.draw_monoids, "drawMonoidBuf");
    BufferProxy clip_inp_buf = named(sizes.clip_inps, "clipInpBuf");
    recording.dispatch(shaders.draw_leaf, wg.draw_leaf,
                       {R(config_buf), R(scene_buf), R(draw_reduced_buf), R(path_bbox_buf), R(draw_monoid_buf), R(info_bin_data_buf),
                        R(clip_inp_buf)});
    recording.free_buffer(draw_reduced_buf);
    BufferProxy clip_el_buf = named(sizes.clip_els, "clipElBuf");
    BufferProxy clip_bic_buf = named(sizes.clip_bics, "clipBicBuf");
    if (wg.clip_reduce[0] > 0)
        recording.dispatch(shaders.clip_reduce, wg.clip_reduce, {R(clip_inp_buf), R(path_bbox_buf), R(clip_bic_buf), R(clip_el_buf)});
    BufferProxy clip_bbox_buf = named(sizes.clip_bboxes, "clipBboxBuf");
    if (wg.clip_leaf[0] > 0)
        recording.dispatch(shaders.clip_leaf, wg.clip_leaf,
                           {R(config_buf), R(clip_inp_buf), R(path_bbox_buf), R(clip_bic_buf), R(clip_el_buf), R(draw_monoid_buf),
                            R(clip_bbox_buf)});
    recording.free_buffer(clip_inp_buf);
    recording.free_buffer(clip_bic_buf);
    recording.free_buffer(clip_el_buf);
    BufferProxy draw_bbox_buf = named(sizes.draw_bboxes, "drawBboxBuf");
    BufferProxy bin_header_buf = named(sizes.bin_headers, "binHeaderBuf");
    recording.dispatch(shaders.binning, wg.binning,
                       {R(config_buf), R(draw_monoid_buf), R(path_bbox_buf), R(clip_bbox_buf), R(draw_bbox_buf), R(bump_buf),
                        R(info_bin_data_buf), R(bin_header_buf)});
    recording.free_buffer(draw_monoid_buf);
    recording.free_buffer(path_bbox_buf);
    recording.free_buffer(clip_bbox_buf);
    BufferProxy path_buf = named(sizes.paths, "pathBuf");
    recording.dispatch(shaders.tile_alloc, wg.tile_alloc,
                       {R(config_buf), R(scene_buf), R(draw_bbox_buf), R(bump_buf), R(path_buf), R(tile_buf)});
    recording.free_buffer(draw_bbox_buf);
    recording.free_buffer(tagmonoid_buf);
    BufferProxy indirect_count_buf = named(sizes.indirect_count, "indirectCount");
    recording.dispatch(shaders.path_count_setup, wg.path_count_setup, {R(bump_buf), R(indirect_count_buf)});
    BufferProxy seg_counts_buf = named(sizes.seg_counts, "segCountsBuf");
    recording.dispatch_indirect(shaders.path_count, indirect_count_buf, 0,
                                {R(config_buf), R(bump_buf), R(lines_buf), R(path_buf), R(tile_buf), R(seg_counts_buf)});
    recording.dispatch(shaders.backdrop_dyn, wg.backdrop, {R(config_buf), R(bump_buf), R(path_buf), R(tile_buf)});
    recording.dispatch(shaders.coarse, wg.coarse,
                       {R(config_buf), R(scene_buf), R(draw_monoid_buf), R(bin_header_buf), R(info_bin_data_buf), R(path_buf), R(tile_buf),
                        R(bump_buf), R(ptcl_buf)});
    recording.dispatch(shaders.path_tiling_setup, wg.path_tiling_setup, {R(bump_buf), R(indirect_count_buf), R(ptcl_buf)});
    recording.dispatch_indirect(shaders.path_tiling, indirect_count_buf, 0,
                                {R(bump_buf), R(seg_counts_buf), R(lines_buf), R(path_buf), R(tile_buf), R(segments_buf)});
    recording.free_buffer(indirect_count_buf);
    recording.free_buffer(seg_counts_buf);
    recording.free_buffer(lines_buf);
    recording.free_buffer(scene_buf);
    recording.free_buffer(draw_monoid_buf);
    recording.free_buffer(bin_header_buf);
    recording.free_buffer(path_buf);
    ImageProxy out_image = new_image_proxy(params.width, params.height, JL_RGBA16_FLOAT);
    BufferProxy blend_spill_buf = named(sizes.blend_spill, "blend_spill");
    if (robust) recording.download(bump_buf);
    recording.free_buffer(bump_buf);

    // RecordFine (render.go:465-547)
    ResourceProxy images_res;
    images_res.kind = ResourceProxy::ImageArray;
    images_res.image_array = image_proxies;
    std::vector<ResourceProxy> fine_bindings = {R(config_buf), R(segments_buf), R(ptcl_buf), R(info_bin_data_buf), R(blend_spill_buf),
                                                ResourceProxy::of(out_image), ResourceProxy::of(gradient_image), images_res};
    switch (params.antialiasing_method) {
        case AaConfig::Area: recording.dispatch(shaders.fine_area, wg.fine, fine_bindings); break;
        default: {
            // render.go:495-507 caches one LUT per renderer and would keep the 8-sample table after a switch to
            // 16 samples; here the cached buffer is tied to the mode it was built for.
            if (mask_buf_.kind == ResourceProxy::None || mask_aa_ != params.antialiasing_method) {
                mask_aa_ = params.antialiasing_method;
                std::vector<uint8_t> lut = params.antialiasing_method == AaConfig::Msaa16 ? make_mask_lut16() : make_mask_lut8();
                mask_buf_ = ResourceProxy::of(recording.upload("mask lut", lut.data(), lut.size()));
            }
            fine_bindings.push_back(mask_buf_);
            recording.dispatch(params.antialiasing_method == AaConfig::Msaa16 ? shaders.fine_msaa16 : shaders.fine_msaa8, wg.fine, fine_bindings);
            break;
        }
    }
    recording.free_buffer(config_buf);
    recording.free_buffer(tile_buf);
    recording.free_buffer(segments_buf);
    recording.free_buffer(ptcl_buf);
    recording.free_image(gradient_image);
    recording.free_buffer(info_bin_data_buf);
    recording.free_buffer(blend_spill_buf);
    res.out_image = ResourceProxy::of(out_image);
    return res;
}

}  // namespace jello
