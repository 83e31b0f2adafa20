// capi.cpp -- flat C entry points over the C++ host layer (Scene / Resolver / Renderer / Engine)
// so that Python (ctypes) can drive it: tests/, bench.py and __graft_entry__.py.  This is glue for
// the build image, which has no Go toolchain; Go callers use the reference's own Scene/renderer
// packages plus the cgo shim of INTEGRATION.md and never see this file.
#include <cstring>
#include <memory>
#include <string>

#include "hip_engine.h"
#include "scene.h"

using namespace jello;

extern "C" {

struct jl_path_el { int32_t kind; int32_t pad; double pts[6]; };
struct jl_color_stop { float offset; float pad; double rgba[4]; };
struct jl_brush {
    int32_t kind;    // Brush::Kind
    int32_t extend;  // Extend
    double color[4];
    double p0[2], p1[2];
    float r0, r1, t0, t1;
    const jl_color_stop* stops;
    int32_t n_stops;
    uint32_t image_width, image_height;
    const uint8_t* image_pixels;
    uint64_t image_key;
};
struct jl_stroke { double width; int32_t join; int32_t start_cap; int32_t end_cap; int32_t pad; double miter_limit; };
struct jl_bump_sizes { uint32_t bin_data, tiles, lines, seg_counts, segments, blend_spill, ptcl; };
struct jl_render_params { double base_color[4]; uint32_t width, height; int32_t aa; uint32_t pad; jl_bump_sizes bump; };

struct jl_binding_c {
    uint32_t kind, count;
    uint64_t id, size;
    uint32_t width, height;
    int32_t format, pad;
    const uint64_t* ids;        // image array: ids
    const uint32_t* dims;       // image array: width,height,format triples
};
struct jl_command_c {
    int32_t kind, shader;
    uint32_t wg[3];
    uint32_t pad;
    uint64_t buf_id, buf_size;
    const char* buf_name;
    uint64_t img_id;
    uint32_t img_w, img_h;
    int32_t img_format, n_bindings;
    const uint8_t* data;
    uint64_t data_len;
    uint64_t offset;
    int64_t size;
    const jl_binding_c* bindings;
    uint32_t coords[4];  // WriteImage: x, y, width, height
};

static thread_local std::string g_err;
const char* jl_last_error() { return g_err.c_str(); }

#define GUARD(expr, failval)                \
    try {                                   \
        expr;                               \
    } catch (const std::exception& e) {     \
        g_err = e.what();                   \
        return failval;                     \
    }

static BezPath to_path(const jl_path_el* els, int n) {
    BezPath p;
    p.reserve((size_t)n);
    for (int i = 0; i < n; i++) {
        PathEl e;
        e.kind = (PathElKind)els[i].kind;
        e.p0[0] = els[i].pts[0]; e.p0[1] = els[i].pts[1]; e.p1[0] = els[i].pts[2]; e.p1[1] = els[i].pts[3];
        e.p2[0] = els[i].pts[4]; e.p2[1] = els[i].pts[5];
        p.push_back(e);
    }
    return p;
}
static Affine to_affine(const double* c) {
    Affine a;
    if (c) std::memcpy(a.c, c, sizeof a.c);
    return a;
}
static Brush to_brush(Scene* scene, const jl_brush* b) {
    Brush r;
    r.kind = (Brush::Kind)b->kind;
    r.extend = (Extend)b->extend;
    r.color = Color{b->color[0], b->color[1], b->color[2], b->color[3]};
    r.p0[0] = b->p0[0]; r.p0[1] = b->p0[1]; r.p1[0] = b->p1[0]; r.p1[1] = b->p1[1];
    r.r0 = b->r0; r.r1 = b->r1; r.t0 = b->t0; r.t1 = b->t1;
    for (int i = 0; i < b->n_stops; i++) {
        ColorStop cs;
        cs.offset = b->stops[i].offset;
        cs.color = Color{b->stops[i].rgba[0], b->stops[i].rgba[1], b->stops[i].rgba[2], b->stops[i].rgba[3]};
        r.stops.push_back(cs);
    }
    r.image.width = b->image_width; r.image.height = b->image_height; r.image.key = b->image_key;
    if (b->image_pixels && b->image_width && b->image_height) {
        // the caller's array is only borrowed for the duration of this call (it used to be kept as a raw pointer and read
        // at render time: a use-after-free for every caller that dropped its array)
        r.image.owned = scene->own_pixels(b->image_key, b->image_pixels, (size_t)b->image_width * b->image_height * 4);
        r.image.pixels = r.image.owned->data();
    }
    return r;
}

// ---- Scene ----
void* jl_scene_new() { return new Scene(); }
void jl_scene_free(void* s) { delete (Scene*)s; }
void jl_scene_reset(void* s) { ((Scene*)s)->reset(); }
int jl_scene_fill(void* s, int fill_rule, const double* transform, const jl_brush* brush, const double* brush_transform, const jl_path_el* els, int n) {
    GUARD(((Scene*)s)->fill((Fill)fill_rule, to_affine(transform), to_brush((Scene*)s, brush), to_affine(brush_transform), to_path(els, n)), -1);
    return 0;
}
int jl_scene_stroke(void* s, const jl_stroke* st, const double* transform, const jl_brush* brush, const double* brush_transform,
                    const jl_path_el* els, int n) {
    Stroke k;
    k.width = st->width; k.join = (Join)st->join; k.start_cap = (Cap)st->start_cap; k.end_cap = (Cap)st->end_cap; k.miter_limit = st->miter_limit;
    GUARD(((Scene*)s)->stroke(k, to_affine(transform), to_brush((Scene*)s, brush), to_affine(brush_transform), to_path(els, n)), -1);
    return 0;
}
int jl_scene_push_layer(void* s, int mix, int compose, float alpha, const double* transform, const jl_path_el* els, int n) {
    BlendMode bm;
    bm.mix = (Mix)mix; bm.compose = (Compose)compose;
    GUARD(((Scene*)s)->push_layer(bm, alpha, to_affine(transform), to_path(els, n)), -1);
    return 0;
}
void jl_scene_pop_layer(void* s) { ((Scene*)s)->pop_layer(); }
void jl_scene_append(void* s, const void* other, const double* transform) { ((Scene*)s)->append(*(const Scene*)other, to_affine(transform)); }
void jl_scene_apply_transform(void* s, const double* transform) { ((Scene*)s)->apply_transform(to_affine(transform)); }
// raw stream access: which = 0 path_tags, 1 path_data, 2 draw_tags, 3 draw_data, 4 transforms, 5 styles
uint64_t jl_scene_stream(void* s, int which, const void** ptr) {
    Encoding& e = ((Scene*)s)->encoding();
    switch (which) {
        case 0: *ptr = e.path_tags.data(); return e.path_tags.size();
        case 1: *ptr = e.path_data.data(); return e.path_data.size();
        case 2: *ptr = e.draw_tags.data(); return e.draw_tags.size() * 4;
        case 3: *ptr = e.draw_data.data(); return e.draw_data.size();
        case 4: *ptr = e.transforms.data(); return e.transforms.size() * sizeof(Transform);
        case 5: *ptr = e.styles.data(); return e.styles.size() * sizeof(Style);
        default: *ptr = nullptr; return 0;
    }
}
void jl_scene_counts(void* s, uint32_t out[4]) {
    Encoding& e = ((Scene*)s)->encoding();
    out[0] = e.num_paths; out[1] = e.num_path_segments; out[2] = e.num_clips; out[3] = e.num_open_clips;
}

// Scene.bumpEstimate (scene.go:36-43) turned into buffer sizes for a width x height render (Scene::bump_sizes)
void jl_scene_bump_sizes(void* s, uint32_t width, uint32_t height, jl_bump_sizes* out) {
    BumpSizes b = ((Scene*)s)->bump_sizes(width, height);
    out->bin_data = b.bin_data; out->tiles = b.tiles; out->lines = b.lines; out->seg_counts = b.seg_counts; out->segments = b.segments;
    out->blend_spill = b.blend_spill; out->ptcl = b.ptcl;
}
// which of those sizes were held below the estimator's bounds (the first attempt is capped at 16 x the reference's constants):
// bit i = field i of jl_bump_sizes.  0 = the sizes are the bounds; otherwise render once with the regrow loop before capturing a graph
uint32_t jl_scene_bump_sizes_clamped(void* s, uint32_t width, uint32_t height) {
    uint32_t held = 0u;
    (void)((Scene*)s)->bump_sizes(width, height, &held);
    return held;
}
// the raw tally of the BumpEstimator: out = {binning, ptcl, tile, blend, seg_counts, segments, lines}
void jl_scene_bump_estimate(void* s, const double* transform, uint32_t* out) {
    Affine a = to_affine(transform);
    BumpEstimate e = ((Scene*)s)->bump_estimate(transform ? &a : nullptr);
    out[0] = e.binning; out[1] = e.ptcl; out[2] = e.tile; out[3] = e.blend; out[4] = e.seg_counts; out[5] = e.segments; out[6] = e.lines;
}

// Bulk helper for the synthetic benchmark scenes: for each i, Fill(NonZero, identity, solid fill_rgba[i])
// of the closed cubic pts[i] and, if widths[i] > 0, Stroke(width, join, caps, solid stroke_rgba[i]) of the
// open cubic.  Exactly equivalent to calling jl_scene_fill / jl_scene_stroke in a loop.
int jl_scene_fill_stroke_cubics(void* s, int n, const double* pts, const double* fill_rgba, const double* stroke_rgba, const double* widths,
                                int join, int start_cap, int end_cap) {
    Scene* sc = (Scene*)s;
    Affine id;
    try {
        for (int i = 0; i < n; i++) {
            const double* p = pts + (size_t)i * 8;
            BezPath path;
            path.push_back(PathEl{PathElKind::MoveTo, {p[0], p[1]}, {0, 0}, {0, 0}});
            path.push_back(PathEl{PathElKind::CubicTo, {p[2], p[3]}, {p[4], p[5]}, {p[6], p[7]}});
            const double* fc = fill_rgba + (size_t)i * 4;
            sc->fill(Fill::NonZero, id, Brush::solid(Color{fc[0], fc[1], fc[2], fc[3]}), id, path);
            if (widths && widths[i] > 0) {
                Stroke st;
                st.width = widths[i]; st.join = (Join)join; st.start_cap = (Cap)start_cap; st.end_cap = (Cap)end_cap;
                const double* scol = stroke_rgba + (size_t)i * 4;
                sc->stroke(st, id, Brush::solid(Color{scol[0], scol[1], scol[2], scol[3]}), id, path);
            }
        }
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
    return 0;
}

// ---- Recording (record-only: needs no GPU) ----
struct RecHandle {
    Renderer::Result result;
    std::map<std::string, BufferProxy> buffers;
    std::vector<jl_command_c> flat;
    std::vector<std::vector<jl_binding_c>> flat_bindings;
    std::vector<std::vector<uint64_t>> ids;
    std::vector<std::vector<uint32_t>> dims;
};
static RenderParams to_params(const jl_render_params* p) {
    RenderParams rp;
    rp.base_color = Color{p->base_color[0], p->base_color[1], p->base_color[2], p->base_color[3]};
    rp.width = p->width; rp.height = p->height; rp.antialiasing_method = (AaConfig)p->aa;
    if (p->bump.lines) {
        rp.bump_sizes.bin_data = p->bump.bin_data; rp.bump_sizes.tiles = p->bump.tiles; rp.bump_sizes.lines = p->bump.lines;
        rp.bump_sizes.seg_counts = p->bump.seg_counts; rp.bump_sizes.segments = p->bump.segments;
        rp.bump_sizes.blend_spill = p->bump.blend_spill; rp.bump_sizes.ptcl = p->bump.ptcl;
    }
    return rp;
}
static void flatten_recording(RecHandle* h) {
    const Recording& rec = h->result.recording;
    h->flat.clear(); h->flat_bindings.clear(); h->ids.clear(); h->dims.clear();
    h->flat_bindings.reserve(rec.commands.size());
    size_t n_arrays = 0;
    for (const Command& c : rec.commands) for (const ResourceProxy& r : c.bindings) if (r.kind == ResourceProxy::ImageArray) n_arrays++;
    h->ids.reserve(n_arrays); h->dims.reserve(n_arrays);
    for (const Command& c : rec.commands) {
        jl_command_c f;
        std::memset(&f, 0, sizeof f);
        f.kind = (int)c.kind; f.shader = c.shader;
        std::memcpy(f.wg, c.wg_count, sizeof f.wg);
        f.buf_id = c.buffer.id; f.buf_size = c.buffer.size; f.buf_name = c.buffer.name.c_str();
        f.img_id = c.image.id; f.img_w = c.image.width; f.img_h = c.image.height; f.img_format = (int)c.image.format;
        f.data = c.data.data(); f.data_len = c.data.size();
        f.offset = c.offset; f.size = c.size;
        std::memcpy(f.coords, c.coords, sizeof f.coords);
        h->flat_bindings.emplace_back();
        std::vector<jl_binding_c>& fb = h->flat_bindings.back();
        for (const ResourceProxy& r : c.bindings) {
            jl_binding_c b;
            std::memset(&b, 0, sizeof b);
            b.kind = (uint32_t)r.kind;
            if (r.kind == ResourceProxy::Buffer) { b.id = r.buffer.id; b.size = r.buffer.size; }
            else if (r.kind == ResourceProxy::Image) { b.id = r.image.id; b.width = r.image.width; b.height = r.image.height; b.format = (int)r.image.format; }
            else if (r.kind == ResourceProxy::ImageArray) {
                h->ids.emplace_back(); h->dims.emplace_back();
                for (const ImageProxy& ip : r.image_array) {
                    h->ids.back().push_back(ip.id);
                    h->dims.back().push_back(ip.width); h->dims.back().push_back(ip.height); h->dims.back().push_back((uint32_t)ip.format);
                }
                b.count = (uint32_t)h->ids.back().size();
                b.ids = h->ids.back().data();
                b.dims = h->dims.back().data();
            }
            fb.push_back(b);
        }
        f.n_bindings = (int)fb.size();
        f.bindings = fb.data();
        h->flat.push_back(f);
    }
}

struct HostState { Renderer renderer; Resolver resolver; FullShaders shaders; };
void* jl_host_new() { return new HostState(); }
void jl_host_free(void* h) { delete (HostState*)h; }

void* jl_record(void* host, void* scene, const jl_render_params* params, int robust) {
    HostState* hs = (HostState*)host;
    std::unique_ptr<RecHandle> h(new RecHandle());
    GUARD(h->result = hs->renderer.render_full(((Scene*)scene)->encoding(), hs->resolver, hs->shaders, to_params(params), robust != 0), nullptr);
    h->buffers = hs->renderer.last_buffers;
    flatten_recording(h.get());
    return h.release();
}
void jl_recording_free(void* r) { delete (RecHandle*)r; }
int jl_recording_len(void* r) { return (int)((RecHandle*)r)->flat.size(); }
const jl_command_c* jl_recording_commands(void* r) { return ((RecHandle*)r)->flat.data(); }
const JlConfig* jl_recording_config(void* r) { return &((RecHandle*)r)->result.config.gpu; }
void jl_recording_target(void* r, uint64_t* id, uint32_t* w, uint32_t* h) {
    const ImageProxy& ip = ((RecHandle*)r)->result.out_image.image;
    *id = ip.id; *w = ip.width; *h = ip.height;
}
// lookup of a named buffer proxy of the recording ("linesBuf", "ptclBuf", ...): returns id, writes size
uint64_t jl_recording_buffer(void* r, const char* name, uint64_t* size) {
    RecHandle* h = (RecHandle*)r;
    auto it = h->buffers.find(name);
    if (it == h->buffers.end()) { if (size) *size = 0; return 0; }
    if (size) *size = it->second.size;
    return it->second.id;
}
void jl_recording_wg_counts(void* r, uint32_t* out, int n) {  // flattened WorkgroupCounts, 3 words each, struct order
    const WorkgroupCounts& w = ((RecHandle*)r)->result.config.workgroup_counts;
    const uint32_t* fields[] = {w.path_reduce, w.path_reduce2, w.path_scan1, w.path_scan, w.bbox_clear, w.flatten, w.draw_reduce, w.draw_leaf,
                                w.clip_reduce, w.clip_leaf, w.binning, w.tile_alloc, w.path_count_setup, w.backdrop, w.coarse,
                                w.path_tiling_setup, w.fine};
    int k = 0;
    for (auto f : fields) for (int i = 0; i < 3 && k < n; i++) out[k++] = f[i];
    if (k < n) out[k] = w.use_large_path_scan ? 1u : 0u;
}

// ---- PTCL statistics (profiling aid: the algorithmic-bytes model of the fine stage) ----
// out[0] words visited (incl. blend_ix, JUMP, END)  out[1] sum of n_segs over CMD_FILL  out[2] info words read
// out[3] gradient/image texel fetches per pixel-command (x256 px)  out[4] blend-spill pixels written+read  out[5] tiles
// out[6] FILL commands  out[7] COLOR commands  Returns 0, or -1 on a malformed stream.
int jl_ptcl_stats(const uint32_t* ptcl, uint64_t n_words, uint32_t width_in_tiles, uint32_t height_in_tiles, uint64_t* out) {
    for (int i = 0; i < 8; i++) out[i] = 0;
    uint64_t n_tiles = (uint64_t)width_in_tiles * height_in_tiles;
    for (uint64_t t = 0; t < n_tiles; t++) {
        uint64_t ix = t * 64;
        if (ix >= n_words) return -1;
        out[0] += 1;
        ix += 1;
        uint32_t depth = 0;
        for (uint64_t guard = 0;; guard++) {
            if (ix >= n_words || guard > (1u << 24)) return -1;
            uint32_t tag = ptcl[ix];
            if (tag == JL_CMD_END) { out[0] += 1; break; }
            switch (tag) {
                case JL_CMD_FILL: out[0] += 4; out[1] += ptcl[ix + 1] >> 1; out[6] += 1; ix += 4; break;
                case JL_CMD_SOLID: out[0] += 1; ix += 1; break;
                case JL_CMD_COLOR: out[0] += 5; out[7] += 1; ix += 5; break;
                case JL_CMD_LIN_GRAD: out[0] += 3; out[2] += 3; out[3] += 256; ix += 3; break;
                case JL_CMD_RAD_GRAD: out[0] += 3; out[2] += 9; out[3] += 256; ix += 3; break;
                case JL_CMD_SWEEP_GRAD: out[0] += 3; out[2] += 8; out[3] += 256; ix += 3; break;
                case JL_CMD_IMAGE: out[0] += 2; out[2] += 8; out[3] += 4 * 256; ix += 2; break;
                case JL_CMD_BEGIN_CLIP: out[0] += 1; if (depth >= JL_BLEND_STACK_SPLIT) out[4] += 256; depth++; ix += 1; break;
                case JL_CMD_END_CLIP: out[0] += 3; if (depth > 0) depth--; if (depth >= JL_BLEND_STACK_SPLIT) out[4] += 256; ix += 3; break;
                case JL_CMD_JUMP: out[0] += 2; ix = ptcl[ix + 1]; break;
                default: return -1;
            }
        }
        out[5] += 1;
    }
    return 0;
}

// ---- Engine (needs the GPU) ----
void* jl_engine_new(int device) {
    Engine* e = nullptr;
    GUARD(e = new Engine(device), nullptr);
    return e;
}
void jl_engine_free(void* e) { delete (Engine*)e; }
void* jl_engine_ctx(void* e) { return ((Engine*)e)->ctx(); }
int jl_engine_run(void* e, void* rec, unsigned flags, uint64_t ext_image_id, void* ext_image_ptr) {
    Engine* eng = (Engine*)e;
    RecHandle* h = (RecHandle*)rec;
    std::vector<ExternalImage> ext;
    if (ext_image_ptr) {
        ImageProxy ip = h->result.out_image.image;
        if (ext_image_id) ip.id = ext_image_id;
        ext.push_back(ExternalImage{ip, ext_image_ptr});
    }
    GUARD(eng->run_recording(h->result.recording, ext, {}, flags), -1);
    return 0;
}
int jl_engine_release(void* e, void* rec) {
    Engine* eng = (Engine*)e;
    RecHandle* h = (RecHandle*)rec;
    Engine::Frame f;
    f.recording = h->result.recording;
    GUARD(eng->release(f), -1);
    return 0;
}
// One-call RenderToTexture with the regrow loop; returns a recording handle of the final attempt
// (retain=1 keeps every buffer alive for inspection until jl_engine_release).
void* jl_engine_render(void* e, void* scene, const jl_render_params* params, void* out_device, int robust, int retain, uint32_t* bump_out,
                       int* attempts) {
    Engine* eng = (Engine*)e;
    std::unique_ptr<RecHandle> h(new RecHandle());
    Engine::Frame f;
    GUARD(f = eng->render_to_texture(((Scene*)scene)->encoding(), to_params(params), out_device, robust != 0, retain != 0), nullptr);
    h->result.recording = std::move(f.recording);
    h->result.config = f.config;
    h->result.out_image = ResourceProxy::of(f.target);
    h->buffers = f.buffers;
    if (bump_out) std::memcpy(bump_out, &f.bump, sizeof(JlBump));
    if (attempts) *attempts = f.attempts;
    flatten_recording(h.get());
    return h.release();
}

}  // extern "C"
