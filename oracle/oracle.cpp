// oracle/oracle.cpp -- TEST INFRASTRUCTURE.  Sequential CPU restatement of the reference's
// WGSL compute pipeline (engine/wgpu_engine/shaders/original/*.wgsl, truth per SURVEY 2.2),
// used only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the CHECKER.
// Nothing in the product path (jello_amd/) may import, link or call this file.
//
// Parity status: the reference ships no tests, fixtures or golden vectors for this path and its
// WGSL executor (honnef.co/go/wgpu v0.0.0-20240719115612-5d243632325b + wgpu-linux-amd64
// v0.1904.1, go.mod:9,15) cannot be built here (no Go, no Vulkan).  The oracle is therefore
// pinned only by (a) the hand-derived known answers of SURVEY Appendix D (tests/golden/c1_kat.json),
// (b) structural invariants, and (c) agreement of its transcendental kernels with libm.
// "PARITY UNPINNED" against an executed reference -- see DESIGN.md.
//
// Canonical allocation order: every `atomicAdd` on a bump allocator in the WGSL is
// executed here in invocation order (workgroup-major, local id ascending) -- the same order the
// reference's own Go CPU shaders use (shaders/cpu/cpu.go, flatten.go) -- with coarse walking
// bin -> tile -> draw object (cpu.go:1096-1270).
//
// Calling convention mirrors the reference's CPU-shader signature
// `func(arena, numWgsX uint32, bindings []cpu.CPUBinding)` (engine/wgpu_engine/wgpu.go:60-63);
// binding order = WGSL @binding order = renderer/render.go dispatch order.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "omath.h"

// Host threads (oracle_set_threads; default 1) for the stages whose invocations are independent (path_tiling: one segment
// record per crossing; fine: disjoint pixels per tile), and -- only when oracle_set_parallel_alloc(1) -- for the three
// stages that ALLOCATE in canonical order (flatten: lines; path_count: segment counts; coarse: PTCL chunks, segments,
// blend space): those then run count -> exclusive scan -> write over chunks of their canonical order, every chunk writing
// at the offset the serial walk would have reached there, so the buffers are identical to the serial ones
// (tests/test_oracle_parallel.py asserts it).  The parallel forms exist for bench.py's cpu_baseline ("all
// cores"); every parity test runs the serial forms.
static int g_oracle_threads = 1;
static int g_oracle_parallel_alloc = 0;
static inline void atomic_min_i32(int32_t* p, int32_t v) {
    int32_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
static inline void atomic_max_i32(int32_t* p, int32_t v) {
    int32_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

using namespace om;

// ---------------------------------------------------------------------------------------------
// Layouts (restated independently of include/jello_formats.h)
// ---------------------------------------------------------------------------------------------
struct Config {  // renderer/config.go:25-80, shared/config.wgsl:5-42
    uint32_t width_in_tiles, height_in_tiles, target_width, target_height;
    float base_color[4];
    uint32_t n_drawobj, n_path, n_clip, bin_data_start;
    uint32_t pathtag_base, pathdata_base, drawtag_base, drawdata_base, transform_base, style_base;
    uint32_t lines_size, binning_size, tiles_size, seg_counts_size, segments_size, blend_size, ptcl_size;
};
static_assert(sizeof(Config) == 100, "config");
struct TagMonoid { uint32_t trans_ix, pathseg_ix, pathseg_offset, style_ix, path_ix; };
struct PathBbox { int32_t x0, y0, x1, y1; uint32_t draw_flags, trans_ix; };
struct LineSoup { uint32_t path_ix, pad; float p0[2], p1[2]; };
struct SegmentCount { uint32_t line_ix, counts; };
struct Segment { float p0[2], p1[2], y_edge; uint32_t pad; };
struct Path { uint32_t bbox[4], tiles, pad[3]; };
struct Tile { int32_t backdrop; uint32_t segment_count_or_ix; };
struct DrawMonoid { uint32_t path_ix, clip_ix, scene_offset, info_offset; };
struct ClipInp { uint32_t ix; int32_t path_ix; };
struct Bic { uint32_t a, b; };
struct ClipEl { uint32_t parent_ix, pad[3]; float bbox[4]; };
struct BinHeader { uint32_t element_count, chunk_offset; };
struct Bump { uint32_t failed, binning, ptcl, tile, seg_counts, segments, blend, lines; };
struct Indirect { uint32_t x, y, z, pad; };
static_assert(sizeof(TagMonoid) == 20 && sizeof(PathBbox) == 24 && sizeof(LineSoup) == 24, "l");
static_assert(sizeof(Segment) == 24 && sizeof(Path) == 32 && sizeof(Tile) == 8 && sizeof(ClipEl) == 32, "l");

enum { STAGE_BINNING = 1, STAGE_TILE_ALLOC = 2, STAGE_FLATTEN = 4, STAGE_PATH_COUNT = 8, STAGE_COARSE = 16 };
enum { WG = 256 };

struct OBuf { void* p; uint64_t n; };  // pointer + size in bytes

template <typename T> struct View {
    T* p; size_t n;
    View(const OBuf& b) : p((T*)b.p), n((size_t)(b.n / sizeof(T))) {}
    // WGSL robust buffer access: out-of-bounds reads yield zero, writes are dropped.
    T rd(size_t i) const { if (i < n) return p[i]; T z; std::memset(&z, 0, sizeof(T)); return z; }
    T* at(size_t i) const { return i < n ? &p[i] : nullptr; }
    void wr(size_t i, const T& v) const { if (i < n) p[i] = v; }
};

struct V2 { float x, y; };
static inline V2 v2(float x, float y) { return V2{x, y}; }
static inline V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
static inline V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
static inline V2 operator*(V2 a, float s) { return v2(a.x * s, a.y * s); }
static inline V2 operator*(float s, V2 a) { return v2(s * a.x, s * a.y); }
static inline V2 operator-(V2 a) { return v2(-a.x, -a.y); }
static inline float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
static inline float length(V2 a) { return sqrt_(a.x * a.x + a.y * a.y); }
static inline V2 normalize(V2 a) { float l = length(a); return v2(a.x / l, a.y / l); }
static inline V2 vmix(V2 a, V2 b, float t) { return v2(mix_(a.x, b.x, t), mix_(a.y, b.y, t)); }
static inline bool veq(V2 a, V2 b) { return a.x == b.x && a.y == b.y; }

struct Transform { float m[4]; float t[2]; };  // shared/transform.wgsl:6-9
static inline V2 transform_apply(const Transform& t, V2 p) {
    return v2(t.m[0] * p.x + t.m[2] * p.y + t.t[0], t.m[1] * p.x + t.m[3] * p.y + t.t[1]);
}
static inline Transform transform_inverse(const Transform& t) {  // shared/transform.wgsl:15-20
    float inv_det = 1.0f / (t.m[0] * t.m[3] - t.m[1] * t.m[2]);
    Transform r;
    r.m[0] = inv_det * t.m[3]; r.m[1] = inv_det * -t.m[1]; r.m[2] = inv_det * -t.m[2]; r.m[3] = inv_det * t.m[0];
    float ntx = -t.t[0], nty = -t.t[1];
    // mat2x2(inv_mat.xy, inv_mat.zw) * v = col0 * v.x + col1 * v.y
    r.t[0] = r.m[0] * ntx + r.m[2] * nty;
    r.t[1] = r.m[1] * ntx + r.m[3] * nty;
    return r;
}
static inline Transform transform_mul(const Transform& a, const Transform& b) {  // :22-27
    Transform r;
    r.m[0] = a.m[0] * b.m[0] + a.m[2] * b.m[1];
    r.m[1] = a.m[1] * b.m[0] + a.m[3] * b.m[1];
    r.m[2] = a.m[0] * b.m[2] + a.m[2] * b.m[3];
    r.m[3] = a.m[1] * b.m[2] + a.m[3] * b.m[3];
    r.t[0] = a.m[0] * b.t[0] + a.m[2] * b.t[1] + a.t[0];
    r.t[1] = a.m[1] * b.t[0] + a.m[3] * b.t[1] + a.t[1];
    return r;
}
static Transform read_transform(const View<uint32_t>& scene, uint32_t transform_base, uint32_t ix) {
    Transform t;
    uint32_t base = transform_base + ix * 6u;
    for (int i = 0; i < 4; i++) t.m[i] = u2f(scene.rd((size_t)base + i));
    t.t[0] = u2f(scene.rd((size_t)base + 4));
    t.t[1] = u2f(scene.rd((size_t)base + 5));
    return t;
}

// ---------------------------------------------------------------------------------------------
// Path tag monoid (shared/pathtag.wgsl:48-71)
// ---------------------------------------------------------------------------------------------
static inline uint32_t popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
static TagMonoid reduce_tag(uint32_t tag_word) {
    TagMonoid c;
    uint32_t point_count = tag_word & 0x3030303u;
    c.pathseg_ix = popc((point_count * 7u) & 0x4040404u);
    c.trans_ix = popc(tag_word & (0x20u * 0x1010101u));
    uint32_t n_points = point_count + ((tag_word >> 2) & 0x1010101u);
    uint32_t a = n_points + (n_points & (((tag_word >> 3) & 0x1010101u) * 15u));
    a += a >> 8;
    a += a >> 16;
    c.pathseg_offset = a & 0xffu;
    c.path_ix = popc(tag_word & (0x10u * 0x1010101u));
    c.style_ix = popc(tag_word & (0x40u * 0x1010101u)) * 2u;
    return c;
}
static TagMonoid combine(TagMonoid a, TagMonoid b) {
    TagMonoid c;
    c.trans_ix = a.trans_ix + b.trans_ix;
    c.pathseg_ix = a.pathseg_ix + b.pathseg_ix;
    c.pathseg_offset = a.pathseg_offset + b.pathseg_offset;
    c.style_ix = a.style_ix + b.style_ix;
    c.path_ix = a.path_ix + b.path_ix;
    return c;
}

// pathtag_reduce.wgsl:21-42 -- [config, scene, reduced]
static void pathtag_reduce(uint32_t n_wg, OBuf* b) {
    const Config& cfg = *(Config*)b[0].p;
    View<uint32_t> scene(b[1]);
    View<TagMonoid> reduced(b[2]);
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        TagMonoid agg = {};
        for (uint32_t i = 0; i < WG; i++) agg = combine(agg, reduce_tag(scene.rd((size_t)cfg.pathtag_base + wg * WG + i)));
        reduced.wr(wg, agg);
    }
}
// pathtag_reduce2.wgsl:23-41 -- [reduced_in, reduced]
static void pathtag_reduce2(uint32_t n_wg, OBuf* b) {
    View<TagMonoid> in(b[0]);
    View<TagMonoid> out(b[1]);
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        TagMonoid agg = {};
        for (uint32_t i = 0; i < WG; i++) agg = combine(agg, in.rd((size_t)wg * WG + i));
        out.wr(wg, agg);
    }
}
// pathtag_scan1.wgsl:26-67 -- [reduced, reduced2, tag_monoids(=reduced_scan)]
static void pathtag_scan1(uint32_t n_wg, OBuf* b) {
    View<TagMonoid> reduced(b[0]);
    View<TagMonoid> reduced2(b[1]);
    View<TagMonoid> out(b[2]);
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        TagMonoid tm = {};
        for (uint32_t l = 0; l < wg && l < WG; l++) tm = combine(tm, reduced2.rd(l));
        for (uint32_t i = 0; i < WG; i++) {
            out.wr((size_t)wg * WG + i, tm);
            tm = combine(tm, reduced.rd((size_t)wg * WG + i));
        }
    }
}
// pathtag_scan.wgsl:28-76 -- [config, scene, reduced|reduced_scan, tag_monoids]
static void pathtag_scan(uint32_t n_wg, OBuf* b, bool small) {
    const Config& cfg = *(Config*)b[0].p;
    View<uint32_t> scene(b[1]);
    View<TagMonoid> reduced(b[2]);
    View<TagMonoid> out(b[3]);
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        TagMonoid tm = {};
        if (small) {
            for (uint32_t l = 0; l < wg && l < WG; l++) tm = combine(tm, reduced.rd(l));
        } else {
            tm = reduced.rd(wg);
        }
        for (uint32_t i = 0; i < WG; i++) {
            size_t ix = (size_t)wg * WG + i;
            out.wr(ix, tm);
            tm = combine(tm, reduce_tag(scene.rd((size_t)cfg.pathtag_base + ix)));
        }
    }
}
// bbox_clear.wgsl:13-24 -- [config, path_bboxes]
static void bbox_clear(uint32_t n_wg, OBuf* b) {
    const Config& cfg = *(Config*)b[0].p;
    View<PathBbox> bb(b[1]);
    for (uint32_t ix = 0; ix < n_wg * WG; ix++) {
        if (ix < cfg.n_path && ix < bb.n) {
            bb.p[ix].x0 = 0x7fffffff; bb.p[ix].y0 = 0x7fffffff;
            bb.p[ix].x1 = (int32_t)0x80000000; bb.p[ix].y1 = (int32_t)0x80000000;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// flatten.wgsl
// ---------------------------------------------------------------------------------------------
namespace fl {

struct Ctx {
    const Config* cfg;
    View<uint32_t> scene;
    View<TagMonoid> tag_monoids;
    View<PathBbox> path_bboxes;
    Bump* bump;
    View<LineSoup> lines;
    float bbox[4];  // var<private> bbox (flatten.wgsl:807)
    bool skip_bbox = false, atomic_bbox = false;  // the parallel form: count pass / write pass (see g_oracle_parallel_alloc)
    Ctx(OBuf* b) : cfg((Config*)b[0].p), scene(b[1]), tag_monoids(b[2]), path_bboxes(b[3]), bump((Bump*)b[4].p), lines(b[5]) {}
};

struct CubicParams { float th0, th1, chord_len, err; };
struct EulerParams { float th0, th1, k0, k1, ch; };
struct EulerSeg { V2 p0, p1; EulerParams params; };
struct CubicPoints { V2 p0, p1, p2, p3; };
struct PointDeriv { V2 point, deriv; };
struct PathTagData { uint32_t tag_byte; TagMonoid monoid; };

static const float DERIV_THRESH = 1e-6f;
static const float DERIV_THRESH_SQUARED = DERIV_THRESH * DERIV_THRESH;
static const float DERIV_EPS = 1e-6f;
static const float SUBDIV_LIMIT = 1.0f / 65536.0f;
static const float K1_THRESH = 1e-3f;
static const float DIST_THRESH = 1e-3f;
static const float TANGENT_THRESH = 1e-6f;

// flatten.wgsl:94-133
static CubicParams cubic_from_points_derivs(V2 p0, V2 p1, V2 q0, V2 q1, float dt) {
    V2 chord = p1 - p0;
    float chord_squared = dot(chord, chord);
    float chord_len = sqrt_(chord_squared);
    if (chord_squared < DERIV_THRESH_SQUARED) {
        float chord_err = sqrt_((float)(9.0 / 32.0) * (dot(q0, q0) + dot(q1, q1))) * dt;
        return CubicParams{0.0f, 0.0f, DERIV_THRESH, chord_err};
    }
    float scale = dt / chord_squared;
    V2 h0 = v2(q0.x * chord.x + q0.y * chord.y, q0.y * chord.x - q0.x * chord.y);
    float th0 = atan2_(h0.y, h0.x);
    float d0 = length(h0) * scale;
    V2 h1 = v2(q1.x * chord.x + q1.y * chord.y, q1.x * chord.y - q1.y * chord.x);
    float th1 = atan2_(h1.y, h1.x);
    float d1 = length(h1) * scale;
    float cth0 = cos_(th0);
    float cth1 = cos_(th1);
    float err = 2.0f;
    if (cth0 * cth1 >= 0.0f) {
        const float TWO_THIRDS = (float)(2.0 / 3.0);
        float e0 = TWO_THIRDS / fmax_(1.0f + cth0, 1e-9f);
        float e1 = TWO_THIRDS / fmax_(1.0f + cth1, 1e-9f);
        float s0 = sin_(th0);
        float s1 = sin_(th1);
        float s01 = cth0 * s1 + cth1 * s0;
        float amin = 0.15f * (2.0f * e0 * s0 + 2.0f * e1 * s1 - e0 * e1 * s01);
        float a = 0.15f * (2.0f * d0 * s0 + 2.0f * d1 * s1 - d0 * d1 * s01);
        float aerr = abs_(a - amin);
        float symm = abs_(th0 + th1);
        float asymm = abs_(th0 - th1);
        float dist = length(v2(d0 - e0, d1 - e1));
        float symm2 = symm * symm;
        float ctr = (4.625e-6f * symm * symm2 + 7.5e-3f * asymm) * symm2;
        float halo = (5e-3f * symm + 7e-2f * asymm) * dist;
        err = ctr + 1.55f * aerr + halo;
    }
    err *= chord_len;
    return CubicParams{th0, th1, chord_len, err};
}

// flatten.wgsl:135-158
static EulerParams es_params_from_angles(float th0, float th1) {
    float k0 = th0 + th1;
    float dth = th1 - th0;
    float d2 = dth * dth;
    float k2 = k0 * k0;
    float a = 6.0f;
    a -= d2 * (float)(1.0 / 70.0);
    a -= (d2 * d2) * (float)(1.0 / 10780.0);
    a += (d2 * d2 * d2) * 2.769178184818219e-07f;
    float b = -0.1f + d2 * (float)(1.0 / 4200.0) + d2 * d2 * 1.6959677820260655e-05f;
    float c = (float)(-1.0 / 1400.0) + d2 * 6.84915970574303e-05f - k2 * 7.936475029053326e-06f;
    a += (b + c * k2) * k2;
    float k1 = dth * a;
    float ch = 1.0f;
    ch -= d2 * (float)(1.0 / 40.0);
    ch += (d2 * d2) * 0.00034226190482569864f;
    ch -= (d2 * d2 * d2) * 1.9349474568904524e-06f;
    float b_ = (float)(-1.0 / 24.0) + d2 * 0.0024702380951963226f - d2 * d2 * 3.7297408997537985e-05f;
    float c_ = (float)(1.0 / 1920.0) - d2 * 4.87350869747975e-05f - k2 * 3.1001936068463107e-06f;
    ch += (b_ + c_ * k2) * k2;
    return EulerParams{th0, th1, k0, k1, ch};
}
static float es_params_eval_th(const EulerParams& p, float t) {  // :160-162
    return (p.k0 + 0.5f * p.k1 * (t - 1.0f)) * t - p.th0;
}
// flatten.wgsl:165-195
static V2 integ_euler_10(float k0, float k1) {
    float t1_1 = k0;
    float t1_2 = 0.5f * k1;
    float t2_2 = t1_1 * t1_1;
    float t2_3 = 2.0f * (t1_1 * t1_2);
    float t2_4 = t1_2 * t1_2;
    float t3_4 = t2_2 * t1_2 + t2_3 * t1_1;
    float t3_6 = t2_4 * t1_2;
    float t4_4 = t2_2 * t2_2;
    float t4_5 = 2.0f * (t2_2 * t2_3);
    float t4_6 = 2.0f * (t2_2 * t2_4) + t2_3 * t2_3;
    float t4_7 = 2.0f * (t2_3 * t2_4);
    float t4_8 = t2_4 * t2_4;
    float t5_6 = t4_4 * t1_2 + t4_5 * t1_1;
    float t5_8 = t4_6 * t1_2 + t4_7 * t1_1;
    float t6_6 = t4_4 * t2_2;
    float t6_7 = t4_4 * t2_3 + t4_5 * t2_2;
    float t6_8 = t4_4 * t2_4 + t4_5 * t2_3 + t4_6 * t2_2;
    float t7_8 = t6_6 * t1_2 + t6_7 * t1_1;
    float t8_8 = t6_6 * t2_2;
    float u = 1.0f;
    u -= (float)(1.0 / 24.0) * t2_2 + (float)(1.0 / 160.0) * t2_4;
    u += (float)(1.0 / 1920.0) * t4_4 + (float)(1.0 / 10752.0) * t4_6 + (float)(1.0 / 55296.0) * t4_8;
    u -= (float)(1.0 / 322560.0) * t6_6 + (float)(1.0 / 1658880.0) * t6_8;
    u += (float)(1.0 / 92897280.0) * t8_8;
    float v = (float)(1.0 / 12.0) * t1_2;
    v -= (float)(1.0 / 480.0) * t3_4 + (float)(1.0 / 2688.0) * t3_6;
    v += (float)(1.0 / 53760.0) * t5_6 + (float)(1.0 / 276480.0) * t5_8;
    v -= (float)(1.0 / 11612160.0) * t7_8;
    return v2(u, v);
}
static V2 es_params_eval(const EulerParams& p, float t) {  // :197-209
    float thm = es_params_eval_th(p, t * 0.5f);
    float k0 = p.k0, k1 = p.k1;
    V2 uv = integ_euler_10((k0 + k1 * (0.5f * t - 0.5f)) * t, k1 * t * t);
    float scale = t / p.ch;
    float s = scale * sin_(thm);
    float c = scale * cos_(thm);
    float x = uv.x * c - uv.y * s;
    float y = -uv.y * c - uv.x * s;
    return v2(x, y);
}
static V2 es_params_eval_with_offset(const EulerParams& p, float t, float offset) {  // :211-215
    float th = es_params_eval_th(p, t);
    V2 v = offset * v2(sin_(th), cos_(th));
    return es_params_eval(p, t) + v;
}
static V2 es_seg_eval_with_offset(const EulerSeg& es, float t, float normalized_offset) {  // :222-226
    V2 chord = es.p1 - es.p0;
    V2 xy = es_params_eval_with_offset(es.params, t, normalized_offset);
    return es.p0 + v2(chord.x * xy.x - chord.y * xy.y, chord.x * xy.y + chord.y * xy.x);
}
static float pow_1_5_signed(float x) { return x * sqrt_(abs_(x)); }

static const float BREAK1 = 0.8f, BREAK2 = 1.25f, BREAK3 = 2.1f;
static const float SIN_SCALE = 1.0976991822760038f;
static const float QUAD_A1 = 0.6406f, QUAD_B1 = -0.81f, QUAD_C1 = 0.9148117935952064f;
static const float QUAD_A2 = 0.5f, QUAD_B2 = -0.156f, QUAD_C2 = 0.16145779359520596f;
static const float QUAD_W1 = 0.5f * QUAD_B1 / QUAD_A1;
static const float QUAD_V1 = 1.0f / QUAD_A1;
static const float QUAD_U1 = QUAD_W1 * QUAD_W1 - QUAD_C1 / QUAD_A1;
static const float QUAD_W2 = 0.5f * QUAD_B2 / QUAD_A2;
static const float QUAD_V2 = 1.0f / QUAD_A2;
static const float QUAD_U2 = QUAD_W2 * QUAD_W2 - QUAD_C2 / QUAD_A2;
static const float FRAC_PI_4 = 0.7853981633974483f;
static const float CBRT_9_8 = 1.040041911525952f;

static float espc_int_approx(float x) {  // :250-262
    float y = abs_(x);
    float a;
    if (y < BREAK1) {
        a = sin_(SIN_SCALE * y) * (1.0f / SIN_SCALE);
    } else if (y < BREAK2) {
        a = (float)(2.8284271247461903 / 3.0) * pow_1_5_signed(y - 1.0f) + FRAC_PI_4;
    } else {
        bool lt = y < BREAK3;
        float qa = lt ? QUAD_A1 : QUAD_A2, qb = lt ? QUAD_B1 : QUAD_B2, qc = lt ? QUAD_C1 : QUAD_C2;
        a = (qa * y + qb) * y + qc;
    }
    return a * sign_(x);
}
static float espc_int_inv_approx(float x) {  // :264-278
    float y = abs_(x);
    float a;
    if (y < 0.7010707591262915f) {
        a = asin_(y * SIN_SCALE) * (1.0f / SIN_SCALE);
    } else if (y < 0.903249293595206f) {
        float b = y - FRAC_PI_4;
        float u = pow23_abs_(b) * sign_(b);
        a = u * CBRT_9_8 + 1.0f;
    } else {
        bool lt = y < 2.038857793595206f;
        float qu = lt ? QUAD_U1 : QUAD_U2, qv = lt ? QUAD_V1 : QUAD_V2, qw = lt ? QUAD_W1 : QUAD_W2;
        a = sqrt_(qu + qv * y) - qw;
    }
    return a * sign_(x);
}
static PointDeriv eval_cubic_and_deriv(V2 p0, V2 p1, V2 p2, V2 p3, float t) {  // :285-293
    float m = 1.0f - t;
    float mm = m * m;
    float mt = m * t;
    float tt = t * t;
    V2 p = p0 * (mm * m) + (p1 * (3.0f * mm) + p2 * (3.0f * mt) + p3 * tt) * t;
    V2 q = (p1 - p0) * mm + (p2 - p1) * (2.0f * mt) + (p3 - p2) * tt;
    return PointDeriv{p, q};
}
static V2 cubic_start_tangent(V2 p0, V2 p1, V2 p2, V2 p3) {  // :295-301
    const float EPS = 1e-12f;
    V2 d01 = p1 - p0, d02 = p2 - p0, d03 = p3 - p0;
    V2 inner = (dot(d02, d02) > EPS) ? d02 : d03;
    return (dot(d01, d01) > EPS) ? d01 : inner;
}
static V2 cubic_end_tangent(V2 p0, V2 p1, V2 p2, V2 p3) {  // :303-309
    const float EPS = 1e-12f;
    V2 d23 = p3 - p2, d13 = p3 - p1, d03 = p3 - p0;
    V2 inner = (dot(d13, d13) > EPS) ? d13 : d03;
    return (dot(d23, d23) > EPS) ? d23 : inner;
}

// flatten.wgsl:749-756
static void write_line(Ctx& c, uint32_t line_ix, uint32_t path_ix, V2 p0, V2 p1) {
    c.bbox[0] = fmin_(c.bbox[0], fmin_(p0.x, p1.x));
    c.bbox[1] = fmin_(c.bbox[1], fmin_(p0.y, p1.y));
    c.bbox[2] = fmax_(c.bbox[2], fmax_(p0.x, p1.x));
    c.bbox[3] = fmax_(c.bbox[3], fmax_(p0.y, p1.y));
    if (line_ix < c.cfg->lines_size && line_ix < c.lines.n) {
        LineSoup l; l.path_ix = path_ix; l.pad = 0; l.p0[0] = p0.x; l.p0[1] = p0.y; l.p1[0] = p1.x; l.p1[1] = p1.y;
        c.lines.p[line_ix] = l;
    }
}
static void write_line_with_transform(Ctx& c, uint32_t line_ix, uint32_t path_ix, V2 p0, V2 p1, const Transform& t) {
    write_line(c, line_ix, path_ix, transform_apply(t, p0), transform_apply(t, p1));
}
static uint32_t alloc_lines(Ctx& c, uint32_t n) { uint32_t ix = c.bump->lines; c.bump->lines += n; return ix; }
static void output_line_with_transform(Ctx& c, uint32_t path_ix, V2 p0, V2 p1, const Transform& t) {
    uint32_t line_ix = alloc_lines(c, 1);
    write_line_with_transform(c, line_ix, path_ix, p0, p1, t);
}
static void output_two_lines_with_transform(Ctx& c, uint32_t path_ix, V2 p00, V2 p01, V2 p10, V2 p11, const Transform& t) {
    uint32_t line_ix = alloc_lines(c, 2);
    write_line_with_transform(c, line_ix, path_ix, p00, p01, t);
    write_line_with_transform(c, line_ix + 1, path_ix, p10, p11, t);
}

enum { ESPC_ROBUST_NORMAL = 0, ESPC_ROBUST_LOW_K1 = 1, ESPC_ROBUST_LOW_DIST = 2 };

// flatten.wgsl:328-477
// instrumentation only (tools/flatten_stats.py): [jobs, attempts, lines, pieces, histogram of attempts per job (28 bins)]
static uint64_t g_flatten_stats[32];
static uint64_t g_flatten_depth[20];  // histogram over jobs of the deepest accepted piece (dt = 2^-depth), bin 19 = deeper
extern "C" void oracle_flatten_depth(uint64_t* out, int reset) {
    for (int i = 0; i < 20; i++) { out[i] = g_flatten_depth[i]; if (reset) g_flatten_depth[i] = 0; }
}
static float* g_flatten_pairs = nullptr;  // optional (root error * scale, attempts) log for tools/flatten_stats.py
static size_t g_flatten_pairs_n = 0, g_flatten_pairs_cap = 0;
extern "C" void oracle_flatten_pairs(float* buf, size_t cap) { g_flatten_pairs = buf; g_flatten_pairs_cap = cap; g_flatten_pairs_n = 0; }
extern "C" size_t oracle_flatten_pairs_count() { return g_flatten_pairs_n; }
extern "C" void oracle_flatten_stats(uint64_t* out, int reset) {
    for (int i = 0; i < 32; i++) { out[i] = g_flatten_stats[i]; if (reset) g_flatten_stats[i] = 0; }
}
static void flatten_euler(Ctx& c, const CubicPoints& cubic, uint32_t path_ix, const Transform& local_to_device,
                          float offset, V2 start_p, V2 end_p) {
    V2 p0, p1, p2, p3;
    float scale;
    Transform transform;
    V2 t_start = start_p, t_end = end_p;
    if (offset == 0.0f) {
        const Transform& t = local_to_device;
        p0 = transform_apply(t, cubic.p0);
        p1 = transform_apply(t, cubic.p1);
        p2 = transform_apply(t, cubic.p2);
        p3 = transform_apply(t, cubic.p3);
        scale = 1.0f;
        transform = Transform{{1.0f, 0.0f, 0.0f, 1.0f}, {0.0f, 0.0f}};
        t_start = p0;
        t_end = p3;
    } else {
        p0 = cubic.p0; p1 = cubic.p1; p2 = cubic.p2; p3 = cubic.p3;
        transform = local_to_device;
        const float* mat = transform.m;
        scale = 0.5f * length(v2(mat[0] + mat[3], mat[1] - mat[2])) + length(v2(mat[0] - mat[3], mat[1] + mat[2]));
    }
    if (veq(p0, p1) && veq(p0, p2) && veq(p0, p3)) return;

    const float tol = 0.25f;
    uint32_t t0_u = 0u;
    float dt = 1.0f;
    V2 last_p = p0;
    V2 last_q = p1 - p0;
    if (dot(last_q, last_q) < DERIV_THRESH_SQUARED) last_q = eval_cubic_and_deriv(p0, p1, p2, p3, DERIV_EPS).deriv;
    float last_t = 0.0f;
    V2 lp0 = t_start;
    uint32_t st_attempts = 0u, st_pieces = 0u, st_depth = 0u;
    float st_root_err = 0.0f;
    for (;;) {
        float t0 = (float)t0_u * dt;
        if (t0 == 1.0f) break;
        st_attempts++;
        float t1 = t0 + dt;
        V2 this_p0 = last_p;
        V2 this_q0 = last_q;
        PointDeriv this_pq1 = eval_cubic_and_deriv(p0, p1, p2, p3, t1);
        if (dot(this_pq1.deriv, this_pq1.deriv) < DERIV_THRESH_SQUARED) {
            PointDeriv new_pq1 = eval_cubic_and_deriv(p0, p1, p2, p3, t1 - DERIV_EPS);
            this_pq1.deriv = new_pq1.deriv;
            if (t1 < 1.0f) {
                this_pq1.point = new_pq1.point;
                t1 = t1 - DERIV_EPS;
            }
        }
        float actual_dt = t1 - last_t;
        CubicParams cp = cubic_from_points_derivs(this_p0, this_pq1.point, this_q0, this_pq1.deriv, actual_dt);
        if (st_attempts == 1u) st_root_err = cp.err * scale;
        if (cp.err * scale <= tol || dt <= SUBDIV_LIMIT) {
            EulerParams ep = es_params_from_angles(cp.th0, cp.th1);
            EulerSeg es{this_p0, this_pq1.point, ep};
            float k0 = es.params.k0 - 0.5f * es.params.k1;
            float k1 = es.params.k1;
            float normalized_offset = offset / cp.chord_len;
            float dist_scaled = normalized_offset * es.params.ch;
            float scale_multiplier = sqrt_(0.125f * scale * cp.chord_len / (es.params.ch * tol));
            float a = 0.0f, b = 0.0f, integral = 0.0f, int0 = 0.0f, n_frac;
            int robust = ESPC_ROBUST_NORMAL;
            if (abs_(k1) < K1_THRESH) {
                float k = es.params.k0;
                n_frac = sqrt_(abs_(k * (k * dist_scaled + 1.0f)));
                robust = ESPC_ROBUST_LOW_K1;
            } else if (abs_(dist_scaled) < DIST_THRESH) {
                a = k1;
                b = k0;
                int0 = pow_1_5_signed(b);
                float int1 = pow_1_5_signed(a + b);
                integral = int1 - int0;
                n_frac = (float)(2.0 / 3.0) * integral / a;
                robust = ESPC_ROBUST_LOW_DIST;
            } else {
                a = -2.0f * dist_scaled * k1;
                b = -1.0f - 2.0f * dist_scaled * k0;
                int0 = espc_int_approx(b);
                float int1 = espc_int_approx(a + b);
                integral = int1 - int0;
                float k_peak = k0 - k1 * b / a;
                float integrand_peak = sqrt_(abs_(k_peak * (k_peak * dist_scaled + 1.0f)));
                n_frac = integral * integrand_peak / a;
            }
            float n = clamp_(ceil_(n_frac * scale_multiplier), 1.0f, 100.0f);
            uint32_t n_u = to_u32(n);
            for (uint32_t i = 0; i < n_u; i++) {
                V2 lp1;
                if (i + 1u == n_u && t1 == 1.0f) {
                    lp1 = t_end;
                } else {
                    float t = (float)(i + 1u) / n;
                    float s = t;
                    if (robust != ESPC_ROBUST_LOW_K1) {
                        float u = integral * t + int0;
                        float inv;
                        if (robust == ESPC_ROBUST_LOW_DIST) {
                            inv = pow23_abs_(u) * sign_(u);
                        } else {
                            inv = espc_int_inv_approx(u);
                        }
                        s = (inv - b) / a;
                    }
                    lp1 = es_seg_eval_with_offset(es, s, normalized_offset);
                }
                V2 l0 = (offset >= 0.0f) ? lp0 : lp1;
                V2 l1 = (offset >= 0.0f) ? lp1 : lp0;
                output_line_with_transform(c, path_ix, l0, l1, transform);
                lp0 = lp1;
            }
            st_pieces++;
            { uint32_t d = 0u; float q = dt; while (q < 1.0f && d < 19u) { q *= 2.0f; d++; } if (d > st_depth) st_depth = d; }
            if (!g_oracle_parallel_alloc) g_flatten_stats[2] += n_u;
            last_p = this_pq1.point;
            last_q = this_pq1.deriv;
            last_t = t1;
            t0_u += 1u;
            uint32_t shift = (t0_u == 0u) ? 32u : (uint32_t)__builtin_ctz(t0_u);
            t0_u = (shift >= 32u) ? 0u : (t0_u >> shift);
            dt *= (float)(1u << (shift & 31u));
        } else {
            t0_u = t0_u * 2u;
            dt *= 0.5f;
        }
    }
    if (g_oracle_parallel_alloc) return;  // (the statistics below are for the serial form: shared counters)
    g_flatten_stats[0] += 1u;
    g_flatten_depth[st_depth] += 1u;
    g_flatten_stats[1] += st_attempts;
    g_flatten_stats[3] += st_pieces;
    g_flatten_stats[4 + (st_attempts < 27u ? st_attempts : 27u)] += 1u;
    if (g_flatten_pairs && g_flatten_pairs_n < g_flatten_pairs_cap) {
        g_flatten_pairs[2 * g_flatten_pairs_n] = st_root_err;
        g_flatten_pairs[2 * g_flatten_pairs_n + 1] = (float)st_attempts;
        g_flatten_pairs_n++;
    }
}

// flatten.wgsl:490-517
static void flatten_arc(Ctx& c, uint32_t path_ix, V2 begin, V2 end, V2 center, float angle, const Transform& transform) {
    V2 p0 = transform_apply(transform, begin);
    V2 r = begin - center;
    const float MIN_THETA = 0.0001f;
    const float tol = 0.25f;
    float radius = fmax_(tol, length(p0 - transform_apply(transform, center)));
    float theta = fmax_(MIN_THETA, 2.0f * acos_(1.0f - tol / radius));
    uint32_t n_lines = umax_(1u, to_u32(ceil_(angle / theta)));
    float cs = cos_(theta);
    float sn = sin_(theta);
    uint32_t line_ix = alloc_lines(c, n_lines);
    for (uint32_t i = 0; i < n_lines - 1u; i++) {
        r = v2(cs * r.x + sn * r.y, -sn * r.x + cs * r.y);  // mat2x2(c, -s, s, c) * r
        V2 p1 = transform_apply(transform, center + r);
        write_line(c, line_ix + i, path_ix, p0, p1);
        p0 = p1;
    }
    V2 p1 = transform_apply(transform, end);
    write_line(c, line_ix + n_lines - 1u, path_ix, p0, p1);
}
// flatten.wgsl:519-543
static void draw_cap(Ctx& c, uint32_t path_ix, uint32_t cap_style, V2 point, V2 cap0, V2 cap1, V2 offset_tangent,
                     const Transform& transform) {
    if (cap_style == 0x02000000u) {
        flatten_arc(c, path_ix, cap0, cap1, point, 3.1415927f, transform);
        return;
    }
    V2 start = cap0, end = cap1;
    bool is_square = (cap_style == 0x01000000u);
    uint32_t line_ix = alloc_lines(c, is_square ? 3u : 1u);
    if (is_square) {
        V2 v = offset_tangent;
        V2 p0 = start + v;
        V2 p1 = end + v;
        write_line_with_transform(c, line_ix + 1u, path_ix, start, p0, transform);
        write_line_with_transform(c, line_ix + 2u, path_ix, p1, end, transform);
        start = p0;
        end = p1;
    }
    write_line_with_transform(c, line_ix, path_ix, start, end, transform);
}
// flatten.wgsl:545-614
static void draw_join(Ctx& c, uint32_t path_ix, uint32_t style_flags, V2 p0, V2 tan_prev, V2 tan_next, V2 n_prev,
                      V2 n_next, const Transform& transform) {
    V2 front0 = p0 + n_prev;
    V2 front1 = p0 + n_next;
    V2 back0 = p0 - n_next;
    V2 back1 = p0 - n_prev;
    float cr = tan_prev.x * tan_next.y - tan_prev.y * tan_next.x;
    float d = dot(tan_prev, tan_next);
    switch (style_flags & 0x30000000u) {
        case 0u: output_two_lines_with_transform(c, path_ix, front0, front1, back0, back1, transform); break;
        case 0x10000000u: {
            float hypot = length(v2(cr, d));
            float miter_limit = f16_to_f32((uint16_t)(style_flags & 0xFFFFu));
            uint32_t line_ix;
            if (2.0f * hypot < (hypot + d) * miter_limit * miter_limit && cr != 0.0f) {
                bool is_backside = cr > 0.0f;
                V2 fp_last = is_backside ? back1 : front0;
                V2 fp_this = is_backside ? back0 : front1;
                V2 p = is_backside ? back0 : front0;
                V2 v = fp_this - fp_last;
                float h = (tan_prev.x * v.y - tan_prev.y * v.x) / cr;
                V2 miter_pt = fp_this - tan_next * h;
                line_ix = alloc_lines(c, 3u);
                write_line_with_transform(c, line_ix, path_ix, p, miter_pt, transform);
                line_ix += 1u;
                if (is_backside) back0 = miter_pt; else front0 = miter_pt;
            } else {
                line_ix = alloc_lines(c, 2u);
            }
            write_line_with_transform(c, line_ix, path_ix, front0, front1, transform);
            write_line_with_transform(c, line_ix + 1u, path_ix, back0, back1, transform);
            break;
        }
        case 0x20000000u: {
            V2 arc0, arc1, other0, other1;
            if (cr > 0.0f) { arc0 = back0; arc1 = back1; other0 = front0; other1 = front1; }
            else { arc0 = front0; arc1 = front1; other0 = back0; other1 = back1; }
            flatten_arc(c, path_ix, arc0, arc1, p0, abs_(atan2_(cr, d)), transform);
            output_line_with_transform(c, path_ix, other0, other1, transform);
            break;
        }
        default: break;
    }
}

static V2 read_f32_point(Ctx& c, uint32_t ix) {
    return v2(u2f(c.scene.rd((size_t)c.cfg->pathdata_base + ix)), u2f(c.scene.rd((size_t)c.cfg->pathdata_base + ix + 1u)));
}
static V2 read_i16_point(Ctx& c, uint32_t ix) {
    uint32_t raw = c.scene.rd((size_t)c.cfg->pathdata_base + ix);
    float x = (float)((int32_t)(raw << 16) >> 16);
    float y = (float)((int32_t)raw >> 16);
    return v2(x, y);
}
// flatten.wgsl:668-682
static PathTagData compute_tag_monoid(Ctx& c, uint32_t ix) {
    uint32_t tag_word = c.scene.rd((size_t)c.cfg->pathtag_base + (ix >> 2));
    uint32_t shift = (ix & 3u) * 8u;
    TagMonoid tm = reduce_tag(tag_word & ((1u << shift) - 1u));
    tm = combine(c.tag_monoids.rd(ix >> 2), tm);
    uint32_t tag_byte = (tag_word >> shift) & 0xffu;
    tm.trans_ix -= 1u;
    tm.style_ix -= 2u;
    return PathTagData{tag_byte, tm};
}
// flatten.wgsl:691-747
static CubicPoints read_path_segment(Ctx& c, const PathTagData& tag, bool is_stroke) {
    V2 p0 = v2(0, 0), p1 = v2(0, 0), p2 = v2(0, 0), p3 = v2(0, 0);
    uint32_t seg_type = tag.tag_byte & 3u;
    uint32_t pathseg_offset = tag.monoid.pathseg_offset;
    bool is_stroke_cap_marker = is_stroke && (tag.tag_byte & 4u) != 0u;
    bool is_open = seg_type == 2u;
    if ((tag.tag_byte & 8u) != 0u) {
        p0 = read_f32_point(c, pathseg_offset);
        p1 = read_f32_point(c, pathseg_offset + 2u);
        if (seg_type >= 2u) {
            p2 = read_f32_point(c, pathseg_offset + 4u);
            if (seg_type == 3u) p3 = read_f32_point(c, pathseg_offset + 6u);
        }
    } else {
        p0 = read_i16_point(c, pathseg_offset);
        p1 = read_i16_point(c, pathseg_offset + 1u);
        if (seg_type >= 2u) {
            p2 = read_i16_point(c, pathseg_offset + 2u);
            if (seg_type == 3u) p3 = read_i16_point(c, pathseg_offset + 3u);
        }
    }
    if (is_stroke_cap_marker && is_open) {
        p0 = p1;
        p1 = p2;
        seg_type = 1u;
    }
    const float THIRD = (float)(1.0 / 3.0);
    if (seg_type == 1u) {
        p3 = p1;
        p2 = vmix(p3, p0, THIRD);
        p1 = vmix(p0, p3, THIRD);
    } else if (seg_type == 2u) {
        p3 = p2;
        p2 = vmix(p1, p2, THIRD);
        p1 = vmix(p1, p0, THIRD);
    }
    return CubicPoints{p0, p1, p2, p3};
}

// flatten.wgsl:809-901, one invocation per tag byte
static void invocation(Ctx& c, uint32_t ix) {
    const Config& cfg = *c.cfg;
    c.bbox[0] = 1e31f; c.bbox[1] = 1e31f; c.bbox[2] = -1e31f; c.bbox[3] = -1e31f;
    PathTagData tag = compute_tag_monoid(c, ix);
    uint32_t path_ix = tag.monoid.path_ix;
    uint32_t style_ix = tag.monoid.style_ix;
    uint32_t trans_ix = tag.monoid.trans_ix;
    PathBbox* out = c.path_bboxes.at(path_ix);
    uint32_t style_flags = c.scene.rd((size_t)cfg.style_base + style_ix);
    uint32_t draw_flags = ((style_flags & 0x40000000u) == 0u) ? 0u : 1u;
    if ((tag.tag_byte & 0x10u) != 0u && out) {
        out->draw_flags = draw_flags;
        out->trans_ix = trans_ix;
    }
    uint32_t seg_type = tag.tag_byte & 3u;
    if (seg_type != 0u) {
        bool is_stroke = (style_flags & 0x80000000u) != 0u;
        Transform transform = read_transform(c.scene, cfg.transform_base, trans_ix);
        CubicPoints pts = read_path_segment(c, tag, is_stroke);
        if (is_stroke) {
            float linewidth = u2f(c.scene.rd((size_t)cfg.style_base + style_ix + 1u));
            float offset = 0.5f * linewidth;
            bool is_open = (tag.tag_byte & 3u) != 1u;
            bool is_stroke_cap_marker = (tag.tag_byte & 4u) != 0u;
            if (is_stroke_cap_marker) {
                if (is_open) {
                    V2 tangent = cubic_start_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
                    V2 offset_tangent = offset * normalize(tangent);
                    V2 n = v2(offset_tangent.y * -1.0f, offset_tangent.x * 1.0f);
                    draw_cap(c, path_ix, (style_flags & 0x0C000000u) >> 2, pts.p0, pts.p0 - n, pts.p0 + n, -offset_tangent, transform);
                }
            } else {
                // read_neighboring_segment (flatten.wgsl:790-800)
                PathTagData ntag = compute_tag_monoid(c, ix + 1u);
                CubicPoints npts = read_path_segment(c, ntag, true);
                bool n_is_closed = (ntag.tag_byte & 3u) == 1u;
                bool n_is_marker = (ntag.tag_byte & 4u) != 0u;
                bool do_join = !n_is_marker || n_is_closed;
                V2 neighbor_tangent = cubic_start_tangent(npts.p0, npts.p1, npts.p2, npts.p3);

                const float TT = TANGENT_THRESH * TANGENT_THRESH;
                V2 tan_start = cubic_start_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
                if (dot(tan_start, tan_start) < TT) tan_start = v2(TANGENT_THRESH, 0.0f);
                V2 tan_prev = cubic_end_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
                if (dot(tan_prev, tan_prev) < TT) tan_prev = v2(TANGENT_THRESH, 0.0f);
                V2 tan_next = neighbor_tangent;
                if (dot(tan_next, tan_next) < TT) tan_next = v2(TANGENT_THRESH, 0.0f);
                V2 n_start = offset * normalize(v2(-tan_start.y, tan_start.x));
                V2 offset_tangent = offset * normalize(tan_prev);
                V2 n_prev = v2(offset_tangent.y * -1.0f, offset_tangent.x * 1.0f);
                V2 tnn = normalize(tan_next);
                V2 n_next = v2((offset * tnn.y) * -1.0f, (offset * tnn.x) * 1.0f);
                flatten_euler(c, pts, path_ix, transform, offset, pts.p0 + n_start, pts.p3 + n_prev);
                flatten_euler(c, pts, path_ix, transform, -offset, pts.p0 - n_start, pts.p3 - n_prev);
                if (do_join) {
                    draw_join(c, path_ix, style_flags, pts.p3, tan_prev, tan_next, n_prev, n_next, transform);
                } else {
                    draw_cap(c, path_ix, (style_flags & 0x03000000u), pts.p3, pts.p3 + n_prev, pts.p3 - n_prev, offset_tangent, transform);
                }
            }
        } else {
            flatten_euler(c, pts, path_ix, transform, 0.0f, pts.p0, pts.p3);
        }
        if ((c.bbox[2] > c.bbox[0] || c.bbox[3] > c.bbox[1]) && out && !c.skip_bbox) {
            if (c.atomic_bbox) {  // (integer min / max: the order of the merges does not matter)
                atomic_min_i32(&out->x0, to_i32(floor_(c.bbox[0])));
                atomic_min_i32(&out->y0, to_i32(floor_(c.bbox[1])));
                atomic_max_i32(&out->x1, to_i32(ceil_(c.bbox[2])));
                atomic_max_i32(&out->y1, to_i32(ceil_(c.bbox[3])));
            } else {
                out->x0 = imin_(out->x0, to_i32(floor_(c.bbox[0])));
                out->y0 = imin_(out->y0, to_i32(floor_(c.bbox[1])));
                out->x1 = imax_(out->x1, to_i32(ceil_(c.bbox[2])));
                out->y1 = imax_(out->y1, to_i32(ceil_(c.bbox[3])));
            }
        }
    }
}
}  // namespace fl

// [config, scene, tag_monoids, path_bboxes, bump, lines]
static void flatten(uint32_t n_wg, OBuf* b) {
    const uint32_t n = n_wg * WG;
    if (!(g_oracle_parallel_alloc && g_oracle_threads > 1)) {
        fl::Ctx c(b);
        for (uint32_t ix = 0; ix < n; ix++) fl::invocation(c, ix);
        return;
    }
    // count -> scan -> write over chunks of consecutive tag bytes
    const uint32_t n_chunks = (uint32_t)g_oracle_threads * 16u;
    const uint32_t len = (n + n_chunks - 1u) / n_chunks;
    std::vector<uint32_t> first(n_chunks + 1u, 0u);
    Bump* shared = (Bump*)b[4].p;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_oracle_threads)
    for (uint32_t ch = 0; ch < n_chunks; ch++) {
        fl::Ctx c(b);
        Bump local = {};
        c.bump = &local;
        c.lines.n = 0;  // nothing is written
        c.skip_bbox = true;
        for (uint32_t ix = ch * len; ix < umin_(n, (ch + 1u) * len); ix++) fl::invocation(c, ix);
        first[ch + 1u] = local.lines;
    }
    const uint32_t start = shared->lines;
    first[0] = start;
    for (uint32_t ch = 0; ch < n_chunks; ch++) first[ch + 1u] += first[ch];
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_oracle_threads)
    for (uint32_t ch = 0; ch < n_chunks; ch++) {
        fl::Ctx c(b);
        Bump local = {};
        local.lines = first[ch];
        c.bump = &local;
        c.atomic_bbox = true;
        for (uint32_t ix = ch * len; ix < umin_(n, (ch + 1u) * len); ix++) fl::invocation(c, ix);
    }
    shared->lines = first[n_chunks];
}

// ---------------------------------------------------------------------------------------------
// draw_reduce.wgsl / draw_leaf.wgsl
// ---------------------------------------------------------------------------------------------
static DrawMonoid map_draw_tag(uint32_t t) {  // shared/drawtag.wgsl:46-53
    DrawMonoid c;
    c.path_ix = (t != 0u) ? 1u : 0u;
    c.clip_ix = t & 1u;
    c.scene_offset = (t >> 2) & 7u;
    c.info_offset = (t >> 6) & 0xfu;
    return c;
}
static DrawMonoid dcombine(DrawMonoid a, DrawMonoid b) {
    return DrawMonoid{a.path_ix + b.path_ix, a.clip_ix + b.clip_ix, a.scene_offset + b.scene_offset, a.info_offset + b.info_offset};
}
static uint32_t read_draw_tag(const Config& cfg, const View<uint32_t>& scene, uint32_t ix) {  // shared/util.wgsl:15-24
    return ix < cfg.n_drawobj ? scene.rd((size_t)cfg.drawtag_base + ix) : 0u;
}
// draw_reduce.wgsl:22-55 -- [config, scene, reduced]
static void draw_reduce(uint32_t n_wg, OBuf* b) {
    const Config& cfg = *(Config*)b[0].p;
    View<uint32_t> scene(b[1]);
    View<DrawMonoid> reduced(b[2]);
    uint32_t num_blocks_total = (cfg.n_drawobj + (WG - 1u)) / WG;
    uint32_t n_blocks_base = num_blocks_total / WG;
    uint32_t remainder = num_blocks_total % WG;
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        uint32_t first_block = n_blocks_base * wg + umin_(wg, remainder);
        uint32_t n_blocks = n_blocks_base + (wg < remainder ? 1u : 0u);
        DrawMonoid agg = {};
        for (uint32_t i = 0; i < n_blocks * WG; i++) agg = dcombine(agg, map_draw_tag(read_draw_tag(cfg, scene, first_block * WG + i)));
        reduced.wr(wg, agg);
    }
}
static Transform from_poly2(V2 p0, V2 p1) {  // draw_leaf.wgsl:279-284
    return Transform{{p1.y - p0.y, p0.x - p1.x, p1.x - p0.x, p1.y - p0.y}, {p0.x, p0.y}};
}
static Transform two_point_to_unit_line(V2 p0, V2 p1) {  // draw_leaf.wgsl:272-277
    Transform tmp1 = from_poly2(p0, p1);
    Transform inv = transform_inverse(tmp1);
    Transform tmp2 = from_poly2(v2(0.0f, 0.0f), v2(1.0f, 0.0f));
    return transform_mul(tmp2, inv);
}
// draw_leaf.wgsl:52-270 -- [config, scene, reduced, path_bbox, draw_monoid, info, clip_inp]
static void draw_leaf(uint32_t n_wg, OBuf* b) {
    const Config& cfg = *(Config*)b[0].p;
    View<uint32_t> scene(b[1]);
    View<DrawMonoid> reduced(b[2]);
    View<PathBbox> path_bbox(b[3]);
    View<DrawMonoid> draw_monoid(b[4]);
    View<uint32_t> info(b[5]);
    View<ClipInp> clip_inp(b[6]);
    uint32_t num_blocks_total = (cfg.n_drawobj + WG - 1u) / WG;
    uint32_t n_blocks_base = num_blocks_total / WG;
    uint32_t remainder = num_blocks_total % WG;
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        DrawMonoid prefix = {};
        for (uint32_t l = 0; l < wg && l < WG; l++) prefix = dcombine(prefix, reduced.rd(l));
        uint32_t first_block = n_blocks_base * wg + umin_(wg, remainder);
        uint32_t n_blocks = n_blocks_base + (wg < remainder ? 1u : 0u);
        DrawMonoid m = prefix;
        for (uint32_t i = 0; i < n_blocks * WG; i++) {
            uint32_t ix = first_block * WG + i;
            uint32_t tag_word = read_draw_tag(cfg, scene, ix);
            if (ix < cfg.n_drawobj) draw_monoid.wr(ix, m);
            uint32_t dd = cfg.drawdata_base + m.scene_offset;
            uint32_t di = m.info_offset;
            if (tag_word == 0x50u || tag_word == 0x114u || tag_word == 0x29cu || tag_word == 0x254u || tag_word == 0x248u ||
                tag_word == 0x9u) {
                PathBbox bbox = path_bbox.rd(m.path_ix);
                Transform transform = {};
                uint32_t draw_flags = bbox.draw_flags;
                if (tag_word == 0x114u || tag_word == 0x29cu || tag_word == 0x254u || tag_word == 0x248u)
                    transform = read_transform(scene, cfg.transform_base, bbox.trans_ix);
                auto S = [&](uint32_t k) { return scene.rd((size_t)dd + k); };
                switch (tag_word) {
                    case 0x50u: info.wr(di, draw_flags); break;
                    case 0x114u: {
                        info.wr(di, draw_flags);
                        V2 p0 = v2(u2f(S(1)), u2f(S(2)));
                        V2 p1 = v2(u2f(S(3)), u2f(S(4)));
                        p0 = transform_apply(transform, p0);
                        p1 = transform_apply(transform, p1);
                        V2 dxy = p1 - p0;
                        float scale = 1.0f / dot(dxy, dxy);
                        V2 line_xy = dxy * scale;
                        float line_c = -dot(p0, line_xy);
                        info.wr(di + 1u, f2u(line_xy.x));
                        info.wr(di + 2u, f2u(line_xy.y));
                        info.wr(di + 3u, f2u(line_c));
                        break;
                    }
                    case 0x29cu: {
                        const float GRADIENT_EPSILON = 1.0f / (float)(1u << 12);
                        info.wr(di, draw_flags);
                        V2 p0 = v2(u2f(S(1)), u2f(S(2)));
                        V2 p1 = v2(u2f(S(3)), u2f(S(4)));
                        float r0 = u2f(S(5));
                        float r1 = u2f(S(6));
                        Transform user_to_gradient = transform_inverse(transform);
                        Transform xform = {};
                        float focal_x = 0.0f, radius = 0.0f;
                        uint32_t kind = 0u, flags = 0u;
                        if (abs_(r0 - r1) <= GRADIENT_EPSILON) {
                            kind = 2u;  // STRIP
                            float scaled = r0 / length(p0 - p1);
                            xform = transform_mul(two_point_to_unit_line(p0, p1), user_to_gradient);
                            radius = scaled * scaled;
                        } else {
                            kind = 4u;  // CONE
                            if (veq(p0, p1)) {
                                kind = 1u;  // CIRCULAR
                                p0 = v2(p0.x + GRADIENT_EPSILON, p0.y + GRADIENT_EPSILON);
                            }
                            if (r1 == 0.0f) {
                                flags |= 1u;  // SWAPPED
                                V2 tmp_p = p0; p0 = p1; p1 = tmp_p;
                                float tmp_r = r0; r0 = r1; r1 = tmp_r;
                            }
                            focal_x = r0 / (r0 - r1);
                            V2 cf = (1.0f - focal_x) * p0 + focal_x * p1;
                            radius = r1 / length(cf - p1);
                            Transform user_to_unit_line = transform_mul(two_point_to_unit_line(cf, p1), user_to_gradient);
                            Transform user_to_scaled;
                            if (abs_(radius - 1.0f) <= GRADIENT_EPSILON) {
                                kind = 3u;  // FOCAL_ON_CIRCLE
                                float scale = 0.5f * abs_(1.0f - focal_x);
                                user_to_scaled = transform_mul(Transform{{scale, 0.0f, 0.0f, scale}, {0.0f, 0.0f}}, user_to_unit_line);
                            } else {
                                float a = radius * radius - 1.0f;
                                float scale_ratio = abs_(1.0f - focal_x) / a;
                                float scale_x = radius * scale_ratio;
                                float scale_y = sqrt_(abs_(a)) * scale_ratio;
                                user_to_scaled = transform_mul(Transform{{scale_x, 0.0f, 0.0f, scale_y}, {0.0f, 0.0f}}, user_to_unit_line);
                            }
                            xform = user_to_scaled;
                        }
                        for (int k = 0; k < 4; k++) info.wr(di + 1u + k, f2u(xform.m[k]));
                        info.wr(di + 5u, f2u(xform.t[0]));
                        info.wr(di + 6u, f2u(xform.t[1]));
                        info.wr(di + 7u, f2u(focal_x));
                        info.wr(di + 8u, f2u(radius));
                        info.wr(di + 9u, (flags << 3) | kind);
                        break;
                    }
                    case 0x254u: {
                        info.wr(di, draw_flags);
                        V2 p0 = v2(u2f(S(1)), u2f(S(2)));
                        Transform xform = transform_mul(transform, Transform{{1.0f, 0.0f, 0.0f, 1.0f}, {p0.x, p0.y}});
                        Transform inv = transform_inverse(xform);
                        for (int k = 0; k < 4; k++) info.wr(di + 1u + k, f2u(inv.m[k]));
                        info.wr(di + 5u, f2u(inv.t[0]));
                        info.wr(di + 6u, f2u(inv.t[1]));
                        info.wr(di + 7u, S(3));
                        info.wr(di + 8u, S(4));
                        break;
                    }
                    case 0x248u: {
                        info.wr(di, draw_flags);
                        Transform inv = transform_inverse(transform);
                        for (int k = 0; k < 4; k++) info.wr(di + 1u + k, f2u(inv.m[k]));
                        info.wr(di + 5u, f2u(inv.t[0]));
                        info.wr(di + 6u, f2u(inv.t[1]));
                        info.wr(di + 7u, S(0));
                        info.wr(di + 8u, S(1));
                        break;
                    }
                    default: break;
                }
            }
            if (tag_word == 0x9u || tag_word == 0x21u) {
                uint32_t path_ix = ~ix;
                if (tag_word == 0x9u) path_ix = m.path_ix;
                clip_inp.wr(m.clip_ix, ClipInp{ix, (int32_t)path_ix});
            }
            m = dcombine(m, map_draw_tag(tag_word));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// clip_reduce.wgsl / clip_leaf.wgsl
// ---------------------------------------------------------------------------------------------
static Bic bic_combine(Bic x, Bic y) {  // shared/clip.wgsl:9-12
    uint32_t m = umin_(x.b, y.a);
    return Bic{x.a + y.a - m, x.b + y.b - m};
}
static void bbox_intersect(const float* a, const float* b, float* o) {  // shared/bbox.wgsl:21-23
    float r0 = fmax_(a[0], b[0]), r1 = fmax_(a[1], b[1]), r2 = fmin_(a[2], b[2]), r3 = fmin_(a[3], b[3]);
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}
// clip_reduce.wgsl:24-67 -- [clip_inp, path_bboxes, reduced(bics), clip_out(els)]
static void clip_reduce(uint32_t n_wg, OBuf* b) {
    View<ClipInp> clip_inp(b[0]);
    View<PathBbox> path_bboxes(b[1]);
    View<Bic> reduced(b[2]);
    View<ClipEl> clip_out(b[3]);
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        // suffix[l] = bic_combine over elements l..255 (reverse scan)
        Bic suffix[WG + 1];
        suffix[WG] = Bic{0, 0};
        for (int l = WG - 1; l >= 0; l--) {
            int32_t inp = clip_inp.rd((size_t)wg * WG + l).path_ix;
            bool is_push = inp >= 0;
            Bic bic{1u - (is_push ? 1u : 0u), is_push ? 1u : 0u};
            suffix[l] = bic_combine(bic, suffix[l + 1]);
        }
        reduced.wr(wg, suffix[0]);
        uint32_t size = suffix[0].b;
        uint32_t sh_parent[WG], sh_path_ix[WG];
        std::memset(sh_parent, 0, sizeof sh_parent);
        std::memset(sh_path_ix, 0, sizeof sh_path_ix);
        for (uint32_t l = 0; l < WG; l++) {
            int32_t inp = clip_inp.rd((size_t)wg * WG + l).path_ix;
            bool is_push = inp >= 0;
            Bic bic = suffix[l + 1];
            if (is_push && bic.a == 0u) {
                uint32_t local_ix = size - bic.b - 1u;
                sh_parent[local_ix] = l;
                sh_path_ix[local_ix] = (uint32_t)inp;
            }
        }
        for (uint32_t l = 0; l < size && l < WG; l++) {
            PathBbox pb = path_bboxes.rd(sh_path_ix[l]);
            ClipEl el = {};
            el.parent_ix = sh_parent[l] + wg * WG;
            el.bbox[0] = (float)pb.x0; el.bbox[1] = (float)pb.y0; el.bbox[2] = (float)pb.x1; el.bbox[3] = (float)pb.y1;
            clip_out.wr((size_t)wg * WG + l, el);
        }
    }
}
// clip_leaf.wgsl:80-207 -- [config, clip_inp, path_bboxes, reduced, clip_els, draw_monoids, clip_bboxes]
// Sequential restatement: a stack of open BeginClips replaces the bicyclic-semigroup search; the
// bbox of an element is the intersection along its parent chain with WGSL bbox_intersect
// (max xy / min zw -- the Go twin's all-max variant, cpu.go:442-447, is a known divergence).
static void clip_leaf(uint32_t n_wg, OBuf* b) {
    const Config& cfg = *(Config*)b[0].p;
    View<ClipInp> clip_inp(b[1]);
    View<PathBbox> path_bboxes(b[2]);
    View<DrawMonoid> draw_monoids(b[5]);
    struct Bb { float v[4]; };
    View<Bb> clip_bboxes(b[6]);
    struct El { uint32_t clip_ix; float bbox[4]; };
    std::vector<El> stack;
    const float INF_BBOX[4] = {-1e9f, -1e9f, 1e9f, 1e9f};
    uint32_t n = umin_(cfg.n_clip, n_wg * WG);
    for (uint32_t g = 0; g < n; g++) {
        ClipInp ci = clip_inp.rd(g);
        Bb out;
        if (ci.path_ix >= 0) {
            PathBbox pb = path_bboxes.rd((size_t)ci.path_ix);
            float own[4] = {(float)pb.x0, (float)pb.y0, (float)pb.x1, (float)pb.y1};
            El el;
            el.clip_ix = g;
            const float* parent = stack.empty() ? INF_BBOX : stack.back().bbox;
            bbox_intersect(parent, own, el.bbox);
            std::memcpy(out.v, el.bbox, 16);
            stack.push_back(el);
        } else {
            if (stack.empty()) {  // unbalanced input: WGSL would read clip_inp[-1]; leave untouched
                std::memcpy(out.v, INF_BBOX, 16);
                clip_bboxes.wr(g, out);
                continue;
            }
            El tos = stack.back();
            stack.pop_back();
            ClipInp parent_clip = clip_inp.rd(tos.clip_ix);
            uint32_t ix = ~(uint32_t)ci.path_ix;
            if (DrawMonoid* dm = draw_monoids.at(ix)) {
                dm->path_ix = (uint32_t)parent_clip.path_ix;
                dm->scene_offset = draw_monoids.rd(parent_clip.ix).scene_offset;
            }
            std::memcpy(out.v, stack.empty() ? INF_BBOX : stack.back().bbox, 16);
        }
        clip_bboxes.wr(g, out);
    }
}

// ---------------------------------------------------------------------------------------------
// binning.wgsl:58-184 -- [config, draw_monoids, path_bbox, clip_bbox, intersected_bbox, bump, bin_data, bin_header]
// ---------------------------------------------------------------------------------------------
static void binning(uint32_t n_wg, OBuf* b) {
    const Config& cfg = *(Config*)b[0].p;
    View<DrawMonoid> draw_monoids(b[1]);
    View<PathBbox> path_bbox_buf(b[2]);
    struct Bb { float v[4]; };
    View<Bb> clip_bbox_buf(b[3]);
    View<Bb> intersected_bbox(b[4]);
    Bump* bump = (Bump*)b[5].p;
    View<uint32_t> bin_data(b[6]);
    View<BinHeader> bin_header(b[7]);
    const float SX = 0.00390625f, SY = 0.00390625f;
    if (bump->lines > cfg.lines_size) {
        bump->failed |= STAGE_FLATTEN;
        return;
    }
    int32_t width_in_bins = (int32_t)((cfg.width_in_tiles + 15u) / 16u);
    int32_t height_in_bins = (int32_t)((cfg.height_in_tiles + 15u) / 16u);
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        int32_t X0[WG], Y0[WG], X1[WG], Y1[WG];
        uint32_t count[WG];
        std::memset(count, 0, sizeof count);
        for (uint32_t l = 0; l < WG; l++) {
            uint32_t element_ix = wg * WG + l;
            int32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
            if (element_ix < cfg.n_drawobj) {
                DrawMonoid dm = draw_monoids.rd(element_ix);
                float clip_bbox[4] = {-1e9f, -1e9f, 1e9f, 1e9f};
                if (dm.clip_ix > 0u) {
                    Bb cb = clip_bbox_buf.rd(umin_(dm.clip_ix - 1u, cfg.n_clip - 1u));
                    std::memcpy(clip_bbox, cb.v, 16);
                }
                PathBbox pb = path_bbox_buf.rd(dm.path_ix);
                float pbf[4] = {(float)pb.x0, (float)pb.y0, (float)pb.x1, (float)pb.y1};
                Bb bbox;
                bbox_intersect(clip_bbox, pbf, bbox.v);
                intersected_bbox.wr(element_ix, bbox);
                if (bbox.v[0] < bbox.v[2] && bbox.v[1] < bbox.v[3]) {
                    x0 = to_i32(floor_(bbox.v[0] * SX));
                    y0 = to_i32(floor_(bbox.v[1] * SY));
                    x1 = to_i32(ceil_(bbox.v[2] * SX));
                    y1 = to_i32(ceil_(bbox.v[3] * SY));
                }
            }
            x0 = iclamp_(x0, 0, width_in_bins);
            y0 = iclamp_(y0, 0, height_in_bins);
            x1 = iclamp_(x1, 0, width_in_bins);
            y1 = iclamp_(y1, 0, height_in_bins);
            if (x0 == x1) y1 = y0;
            X0[l] = x0; Y0[l] = y0; X1[l] = x1; Y1[l] = y1;
            for (int32_t y = y0; y < y1; y++)
                for (int32_t x = x0; x < x1; x++) {
                    int32_t bin = y * width_in_bins + x;
                    if (bin >= 0 && bin < WG) count[bin]++;
                }
        }
        uint32_t chunk_offset[WG];
        for (uint32_t l = 0; l < WG; l++) {
            uint32_t element_count = count[l];
            uint32_t off = bump->binning;
            bump->binning += element_count;
            if (off + element_count > cfg.binning_size) {
                off = 0u;
                bump->failed |= STAGE_BINNING;
            }
            chunk_offset[l] = off;
            bin_header.wr((size_t)wg * WG + l, BinHeader{element_count, off});
        }
        uint32_t cursor[WG];
        std::memset(cursor, 0, sizeof cursor);
        for (uint32_t l = 0; l < WG; l++) {
            uint32_t element_ix = wg * WG + l;
            for (int32_t y = Y0[l]; y < Y1[l]; y++)
                for (int32_t x = X0[l]; x < X1[l]; x++) {
                    int32_t bin = y * width_in_bins + x;
                    if (bin < 0 || bin >= WG) continue;
                    uint32_t idx = cursor[bin]++;
                    bin_data.wr((size_t)cfg.bin_data_start + chunk_offset[bin] + idx, element_ix);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// tile_alloc.wgsl:35-123 -- [config, scene, draw_bboxes, bump, paths, tiles]
// ---------------------------------------------------------------------------------------------
static void tile_alloc(uint32_t n_wg, OBuf* b) {
    const Config& cfg = *(Config*)b[0].p;
    View<uint32_t> scene(b[1]);
    struct Bb { float v[4]; };
    View<Bb> draw_bboxes(b[2]);
    Bump* bump = (Bump*)b[3].p;
    View<Path> paths(b[4]);
    View<Tile> tiles(b[5]);
    if ((bump->failed & (STAGE_BINNING | STAGE_FLATTEN)) != 0u) return;
    const float SX = 1.0f / 16.0f, SY = 1.0f / 16.0f;
    for (uint32_t wg = 0; wg < n_wg; wg++) {
        uint32_t bb[WG][4];
        uint32_t incl[WG];
        uint32_t total = 0;
        for (uint32_t l = 0; l < WG; l++) {
            uint32_t drawobj_ix = wg * WG + l;
            uint32_t drawtag = 0u;
            if (drawobj_ix < cfg.n_drawobj) drawtag = scene.rd((size_t)cfg.drawtag_base + drawobj_ix);
            int32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
            if (drawtag != 0u && drawtag != 0x21u) {
                Bb bbox = draw_bboxes.rd(drawobj_ix);
                if (bbox.v[0] < bbox.v[2] && bbox.v[1] < bbox.v[3]) {
                    x0 = to_i32(floor_(bbox.v[0] * SX));
                    y0 = to_i32(floor_(bbox.v[1] * SY));
                    x1 = to_i32(ceil_(bbox.v[2] * SX));
                    y1 = to_i32(ceil_(bbox.v[3] * SY));
                }
            }
            uint32_t ux0 = (uint32_t)iclamp_(x0, 0, (int32_t)cfg.width_in_tiles);
            uint32_t uy0 = (uint32_t)iclamp_(y0, 0, (int32_t)cfg.height_in_tiles);
            uint32_t ux1 = (uint32_t)iclamp_(x1, 0, (int32_t)cfg.width_in_tiles);
            uint32_t uy1 = (uint32_t)iclamp_(y1, 0, (int32_t)cfg.height_in_tiles);
            uint32_t tile_count = (ux1 - ux0) * (uy1 - uy0);
            bb[l][0] = ux0; bb[l][1] = uy0; bb[l][2] = ux1; bb[l][3] = uy1;
            total += tile_count;
            incl[l] = total;
        }
        uint32_t offset = bump->tile;
        bump->tile += total;
        if (offset + total > cfg.tiles_size) {
            offset = 0u;
            bump->failed |= STAGE_TILE_ALLOC;
        }
        if (Path* p = paths.at((size_t)wg * WG + (WG - 1u))) p->tiles = offset;
        for (uint32_t l = 0; l < WG; l++) {
            uint32_t drawobj_ix = wg * WG + l;
            if (drawobj_ix < cfg.n_drawobj) {
                uint32_t tile_subix = l > 0 ? incl[l - 1] : 0u;
                Path path = {};
                std::memcpy(path.bbox, bb[l], 16);
                path.tiles = offset + tile_subix;
                paths.wr(drawobj_ix, path);
            }
        }
        for (uint32_t i = 0; i < total; i++) tiles.wr((size_t)offset + i, Tile{0, 0u});
    }
}

// path_count_setup.wgsl:17-27 -- [bump, indirect]
static void path_count_setup(OBuf* b) {
    Bump* bump = (Bump*)b[0].p;
    Indirect* ind = (Indirect*)b[1].p;
    if (bump->failed != 0u) ind->x = 0u; else ind->x = (bump->lines + (WG - 1u)) / WG;
    ind->y = 1u; ind->z = 1u;
}

static inline uint32_t span(float a, float b) { return to_u32(fmax_(ceil_(fmax_(a, b)) - floor_(fmin_(a, b)), 1.0f)); }
static const float ONE_MINUS_ULP = 0.99999994f;
static const float ROBUST_EPSILON = 2e-7f;
static const float TILE_SCALE = 0.0625f;

// ---------------------------------------------------------------------------------------------
// path_count.wgsl:51-202 -- [config, bump, lines, paths, tile, seg_counts]
// ---------------------------------------------------------------------------------------------
static void path_count(uint32_t n_wg, OBuf* b) {
    const Config& cfg = *(Config*)b[0].p;
    Bump* bump = (Bump*)b[1].p;
    View<LineSoup> lines(b[2]);
    View<Path> paths(b[3]);
    View<Tile> tile(b[4]);
    View<SegmentCount> seg_counts(b[5]);
    uint32_t n_lines = bump->lines;
    // one line: `cursor` is the segment-count allocator (bump.seg_counts in the serial form); count_only = the first pass of
    // the parallel form (how many crossings the line allocates, nothing written)
    auto one_line = [&](uint32_t gid, uint32_t& cursor, bool count_only) {
        if (!(gid < n_lines)) return;
        LineSoup line = lines.rd(gid);
        V2 lp0 = v2(line.p0[0], line.p0[1]), lp1 = v2(line.p1[0], line.p1[1]);
        bool is_down = lp1.y >= lp0.y;
        V2 xy0 = is_down ? lp0 : lp1;
        V2 xy1 = is_down ? lp1 : lp0;
        V2 s0 = xy0 * TILE_SCALE;
        V2 s1 = xy1 * TILE_SCALE;
        uint32_t count_x = span(s0.x, s1.x) - 1u;
        uint32_t count = count_x + span(s0.y, s1.y);
        uint32_t line_ix = gid;
        float dx = abs_(s1.x - s0.x);
        float dy = s1.y - s0.y;
        if (dx + dy == 0.0f) return;
        if (dy == 0.0f && floor_(s0.y) == s0.y) return;
        float idxdy = 1.0f / (dx + dy);
        float a = dx * idxdy;
        bool is_positive_slope = s1.x >= s0.x;
        float x_sign = is_positive_slope ? 1.0f : -1.0f;
        float xt0 = floor_(s0.x * x_sign);
        float c = s0.x * x_sign - xt0;
        float y0 = floor_(s0.y);
        float ytop = (s0.y == s1.y) ? ceil_(s0.y) : (y0 + 1.0f);
        float bb = fmin_((dy * c + dx * (ytop - s0.y)) * idxdy, ONE_MINUS_ULP);
        float robust_err = floor_(a * ((float)count - 1.0f) + bb) - (float)count_x;
        if (robust_err != 0.0f) a -= ROBUST_EPSILON * sign_(robust_err);
        float x0 = xt0 * x_sign + (is_positive_slope ? 0.0f : -1.0f);

        Path path = paths.rd(line.path_ix);
        int32_t bbox[4] = {(int32_t)path.bbox[0], (int32_t)path.bbox[1], (int32_t)path.bbox[2], (int32_t)path.bbox[3]};
        float xmin = fmin_(s0.x, s1.x);
        int32_t stride = bbox[2] - bbox[0];
        if (s0.y >= (float)bbox[3] || s1.y <= (float)bbox[1] || xmin >= (float)bbox[2] || stride == 0) return;
        uint32_t imin = 0u;
        if (s0.y < (float)bbox[1]) {
            float iminf = round_(((float)bbox[1] - y0 + bb - a) / (1.0f - a)) - 1.0f;
            if (y0 + iminf - floor_(a * iminf + bb) < (float)bbox[1]) iminf += 1.0f;
            imin = to_u32(iminf);
        }
        uint32_t imax = count;
        if (s1.y > (float)bbox[3]) {
            float imaxf = round_(((float)bbox[3] - y0 + bb - a) / (1.0f - a)) - 1.0f;
            if (y0 + imaxf - floor_(a * imaxf + bb) < (float)bbox[3]) imaxf += 1.0f;
            imax = to_u32(imaxf);
        }
        int32_t delta = is_down ? -1 : 1;
        int32_t ymin = 0, ymax = 0;
        if (fmax_(s0.x, s1.x) <= (float)bbox[0]) {
            ymin = to_i32(ceil_(s0.y));
            ymax = to_i32(ceil_(s1.y));
            imax = imin;
        } else {
            float fudge = is_positive_slope ? 0.0f : 1.0f;
            if (xmin < (float)bbox[0]) {
                float f = round_((x_sign * ((float)bbox[0] - x0) - bb + fudge) / a);
                if ((x0 + x_sign * floor_(a * f + bb) < (float)bbox[0]) == is_positive_slope) f += 1.0f;
                int32_t ynext = to_i32(y0 + f - floor_(a * f + bb) + 1.0f);
                if (is_positive_slope) {
                    if (to_u32(f) > imin) {
                        ymin = to_i32(y0 + ((y0 == s0.y) ? 0.0f : 1.0f));
                        ymax = ynext;
                        imin = to_u32(f);
                    }
                } else {
                    if (to_u32(f) < imax) {
                        ymin = ynext;
                        ymax = to_i32(ceil_(s1.y));
                        imax = to_u32(f);
                    }
                }
            }
            if (fmax_(s0.x, s1.x) > (float)bbox[2]) {
                float f = round_((x_sign * ((float)bbox[2] - x0) - bb + fudge) / a);
                if ((x0 + x_sign * floor_(a * f + bb) < (float)bbox[2]) == is_positive_slope) f += 1.0f;
                if (is_positive_slope) imax = umin_(imax, to_u32(f)); else imin = umax_(imin, to_u32(f));
            }
        }
        imax = umax_(imin, imax);
        if (count_only) { cursor += imax - imin; return; }
        ymin = imax_(ymin, bbox[1]);
        ymax = imin_(ymax, bbox[3]);
        for (int32_t y = ymin; y < ymax; y++) {
            int32_t base = (int32_t)path.tiles + (y - bbox[1]) * stride;
            if (Tile* t = tile.at((size_t)(uint32_t)base)) t->backdrop += delta;
        }
        float last_z = floor_(a * ((float)imin - 1.0f) + bb);
        uint32_t seg_base = cursor;
        cursor += imax - imin;
        for (uint32_t i = imin; i < imax; i++) {
            uint32_t subix = i;
            float zf = a * (float)subix + bb;
            float z = floor_(zf);
            int32_t y = to_i32(y0 + (float)subix - z);
            int32_t x = to_i32(x0 + x_sign * z);
            int32_t base = (int32_t)path.tiles + (y - bbox[1]) * stride - bbox[0];
            bool top_edge = (subix == 0u) ? (y0 == s0.y) : (last_z == z);
            if (top_edge && x + 1 < bbox[2]) {
                int32_t x_bump = imax_(x + 1, bbox[0]);
                if (Tile* t = tile.at((size_t)(uint32_t)(base + x_bump))) t->backdrop += delta;
            }
            uint32_t seg_within_slice = 0;
            if (Tile* t = tile.at((size_t)(uint32_t)(base + x))) { seg_within_slice = t->segment_count_or_ix; t->segment_count_or_ix += 1u; }
            uint32_t counts = (seg_within_slice << 16) | subix;
            uint32_t seg_ix = seg_base + i - imin;
            if (seg_ix < cfg.seg_counts_size) seg_counts.wr(seg_ix, SegmentCount{line_ix, counts});
            last_z = z;
        }
    };
    const uint32_t n = n_wg * WG;
    if (!(g_oracle_parallel_alloc && g_oracle_threads > 1)) {
        for (uint32_t gid = 0; gid < n; gid++) one_line(gid, bump->seg_counts, false);
        return;
    }
    // Lines are in path order and the tiles of different paths are disjoint, so chunks of whole paths share nothing but the
    // allocator: count -> scan -> write over such chunks.
    const uint32_t nl = umin_(n, n_lines);
    const uint32_t n_chunks = (uint32_t)g_oracle_threads * 16u;
    const uint32_t len = (nl + n_chunks - 1u) / n_chunks;
    std::vector<uint32_t> lo(n_chunks + 1u, nl), first(n_chunks + 1u, 0u);
    for (uint32_t ch = 0; ch <= n_chunks; ch++) {
        uint32_t p = umin_(nl, ch * len);
        while (p > 0u && p < nl && lines.rd(p).path_ix == lines.rd(p - 1u).path_ix) p++;  // to the next path boundary
        lo[ch] = p;
    }
    lo[0] = 0u;
    for (uint32_t ch = 1; ch <= n_chunks; ch++) lo[ch] = umax_(lo[ch], lo[ch - 1u]);
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_oracle_threads)
    for (uint32_t ch = 0; ch < n_chunks; ch++) {
        uint32_t cnt = 0u;
        for (uint32_t gid = lo[ch]; gid < lo[ch + 1u]; gid++) one_line(gid, cnt, true);
        first[ch + 1u] = cnt;
    }
    first[0] = bump->seg_counts;
    for (uint32_t ch = 0; ch < n_chunks; ch++) first[ch + 1u] += first[ch];
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_oracle_threads)
    for (uint32_t ch = 0; ch < n_chunks; ch++) {
        uint32_t cursor = first[ch];
        for (uint32_t gid = lo[ch]; gid < lo[ch + 1u]; gid++) one_line(gid, cursor, false);
    }
    bump->seg_counts = first[n_chunks];
}



// backdrop_dyn.wgsl:28-86 -- [config, bump, paths, tiles]
static void backdrop_dyn(uint32_t n_wg, OBuf* b) {
    const Config& cfg = *(Config*)b[0].p;
    Bump* bump = (Bump*)b[1].p;
    View<Path> paths(b[2]);
    View<Tile> tiles(b[3]);
    if (bump->failed != 0u) return;
    for (uint32_t ix = 0; ix < n_wg * WG; ix++) {
        if (!(ix < cfg.n_drawobj)) continue;
        Path path = paths.rd(ix);
        uint32_t width = path.bbox[2] - path.bbox[0];
        uint32_t rows = path.bbox[3] - path.bbox[1];
        if (width == 0u) continue;
        for (uint32_t r = 0; r < rows; r++) {
            size_t tile_ix = (size_t)path.tiles + (size_t)r * width;
            int32_t sum = tiles.rd(tile_ix).backdrop;
            for (uint32_t x = 1; x < width; x++) {
                tile_ix++;
                sum += tiles.rd(tile_ix).backdrop;
                if (Tile* t = tiles.at(tile_ix)) t->backdrop = sum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// coarse.wgsl:153-462 -- [config, scene, draw_monoids, bin_headers, info_bin_data, paths, tiles, bump, ptcl]
// ---------------------------------------------------------------------------------------------
namespace co {
struct Ctx {
    const Config* cfg; Bump* bump; View<uint32_t> ptcl; View<Tile> tiles;
    uint32_t cmd_offset, cmd_limit;
    bool dry = false;  // the counting pass of the parallel form: nothing is written (g_oracle_parallel_alloc)
};
static void alloc_cmd(Ctx& c, uint32_t size) {  // coarse.wgsl:70-88
    if (c.cmd_offset + size >= c.cmd_limit) {
        uint32_t ptcl_dyn_start = c.cfg->width_in_tiles * c.cfg->height_in_tiles * 64u;
        uint32_t new_cmd = ptcl_dyn_start + c.bump->ptcl;
        c.bump->ptcl += 256u;
        if (new_cmd + 256u > c.cfg->ptcl_size) {
            new_cmd = 0u;
            c.bump->failed |= STAGE_COARSE;
        }
        c.ptcl.wr(c.cmd_offset, 12u);
        c.ptcl.wr((size_t)c.cmd_offset + 1u, new_cmd);
        c.cmd_offset = new_cmd;
        c.cmd_limit = c.cmd_offset + (256u - 2u);
    }
}
static void write_path(Ctx& c, Tile tile, uint32_t tile_ix, uint32_t draw_flags) {  // :90-112
    uint32_t n_segs = tile.segment_count_or_ix;
    if (n_segs != 0u) {
        uint32_t seg_ix = c.bump->segments;
        c.bump->segments += n_segs;
        if (!c.dry) if (Tile* t = c.tiles.at(tile_ix)) t->segment_count_or_ix = ~seg_ix;
        alloc_cmd(c, 4u);
        c.ptcl.wr(c.cmd_offset, 1u);
        bool even_odd = (draw_flags & 1u) != 0u;
        c.ptcl.wr((size_t)c.cmd_offset + 1u, (n_segs << 1) | (even_odd ? 1u : 0u));
        c.ptcl.wr((size_t)c.cmd_offset + 2u, seg_ix);
        c.ptcl.wr((size_t)c.cmd_offset + 3u, (uint32_t)tile.backdrop);
        c.cmd_offset += 4u;
    } else {
        alloc_cmd(c, 1u);
        c.ptcl.wr(c.cmd_offset, 3u);
        c.cmd_offset += 1u;
    }
}
}  // namespace co

static void coarse(uint32_t n_wg_x, uint32_t n_wg_y, OBuf* b) {
    const Config& cfg = *(Config*)b[0].p;
    View<uint32_t> scene(b[1]);
    View<DrawMonoid> draw_monoids(b[2]);
    View<BinHeader> bin_headers(b[3]);
    View<uint32_t> info_bin_data(b[4]);
    View<Path> paths(b[5]);
    View<Tile> tiles(b[6]);
    Bump* bump = (Bump*)b[7].p;
    View<uint32_t> ptcl(b[8]);
    {
        uint32_t failed = bump->failed & (STAGE_BINNING | STAGE_TILE_ALLOC | STAGE_FLATTEN);
        if (bump->seg_counts > cfg.seg_counts_size) failed |= STAGE_PATH_COUNT;
        if (failed != 0u) { bump->failed |= failed; return; }
    }
    const uint32_t BLEND_CLIP = (128u << 8) | 0u;  // MIX_CLIP<<8 | COMPOSE_SRC_OVER(=0 in Jello)
    uint32_t width_in_bins = (cfg.width_in_tiles + 15u) / 16u;
    uint32_t n_partitions = (cfg.n_drawobj + 255u) / 256u;
    // one bin (= one WGSL workgroup) with the allocator `bump` (the shared one in the serial form); dry = count only
    auto one_bin = [&](uint32_t wx, uint32_t wy, Bump* bump, View<uint32_t> ptcl, bool dry) {
            std::vector<uint32_t> drawobjs;
            uint32_t bin_ix = width_in_bins * wy + wx;
            uint32_t bin_tile_x = 16u * wx, bin_tile_y = 16u * wy;
            // merged, draw-ordered element list of this bin
            drawobjs.clear();
            for (uint32_t part = 0; part < n_partitions; part++) {
                BinHeader h = bin_headers.rd((size_t)part * 256u + bin_ix);
                for (uint32_t j = 0; j < h.element_count; j++)
                    drawobjs.push_back(info_bin_data.rd((size_t)cfg.bin_data_start + h.chunk_offset + j));
            }
            for (uint32_t local = 0; local < 256u; local++) {
                uint32_t tile_x = local % 16u, tile_y = local / 16u;
                uint32_t this_tile_ix = (bin_tile_y + tile_y) * cfg.width_in_tiles + bin_tile_x + tile_x;
                co::Ctx c{&cfg, bump, ptcl, tiles, 0, 0};
                c.dry = dry;
                c.cmd_offset = this_tile_ix * 64u;
                c.cmd_limit = c.cmd_offset + (64u - 2u);
                uint32_t clip_zero_depth = 0u, clip_depth = 0u, render_blend_depth = 0u, max_blend_depth = 0u;
                uint32_t blend_offset = c.cmd_offset;
                c.cmd_offset += 1u;
                for (uint32_t drawobj_ix : drawobjs) {
                    uint32_t tag = scene.rd((size_t)cfg.drawtag_base + drawobj_ix);
                    if (tag == 0u) continue;
                    DrawMonoid dm = draw_monoids.rd(drawobj_ix);
                    Path path = paths.rd(dm.path_ix);
                    uint32_t stride = path.bbox[2] - path.bbox[0];
                    int32_t dx = (int32_t)path.bbox[0] - (int32_t)bin_tile_x;
                    int32_t dy = (int32_t)path.bbox[1] - (int32_t)bin_tile_y;
                    int32_t x0 = iclamp_(dx, 0, 16);
                    int32_t y0 = iclamp_(dy, 0, 16);
                    int32_t x1 = iclamp_((int32_t)path.bbox[2] - (int32_t)bin_tile_x, 0, 16);
                    int32_t y1 = iclamp_((int32_t)path.bbox[3] - (int32_t)bin_tile_y, 0, 16);
                    if (!((int32_t)tile_x >= x0 && (int32_t)tile_x < x1 && (int32_t)tile_y >= y0 && (int32_t)tile_y < y1)) continue;
                    uint32_t base = path.tiles - (uint32_t)(dy * (int32_t)stride + dx);
                    uint32_t tile_ix = base + stride * tile_y + tile_x;
                    Tile tile = tiles.rd(tile_ix);
                    bool is_clip = (tag & 1u) != 0u;
                    bool is_blend = false;
                    uint32_t dd = cfg.drawdata_base + dm.scene_offset;
                    if (is_clip) {
                        uint32_t blend = scene.rd(dd);
                        is_blend = blend != BLEND_CLIP;
                    }
                    uint32_t di = dm.info_offset;
                    uint32_t draw_flags = info_bin_data.rd(di);
                    bool even_odd = (draw_flags & 1u) != 0u;
                    uint32_t n_segs = tile.segment_count_or_ix;
                    int32_t bd = tile.backdrop;
                    int32_t absbd = bd < 0 ? (int32_t)(0u - (uint32_t)bd) : bd;
                    bool backdrop_clear = (even_odd ? (absbd & 1) : bd) == 0;
                    bool include_tile = n_segs != 0u || (backdrop_clear == is_clip) || is_blend;
                    if (!include_tile) continue;
                    // per-tile command emission (coarse.wgsl:350-442)
                    if (clip_zero_depth == 0u) {
                        switch (tag) {
                            case 0x50u: {
                                co::write_path(c, tile, tile_ix, draw_flags);
                                co::alloc_cmd(c, 5u);
                                ptcl.wr(c.cmd_offset, 5u);
                                for (uint32_t k = 0; k < 4; k++) ptcl.wr((size_t)c.cmd_offset + 1u + k, scene.rd((size_t)dd + k));
                                c.cmd_offset += 5u;
                                break;
                            }
                            case 0x114u: case 0x29cu: case 0x254u: {
                                co::write_path(c, tile, tile_ix, draw_flags);
                                uint32_t ty = tag == 0x114u ? 6u : (tag == 0x29cu ? 7u : 8u);
                                co::alloc_cmd(c, 3u);
                                ptcl.wr(c.cmd_offset, ty);
                                ptcl.wr((size_t)c.cmd_offset + 1u, scene.rd(dd));
                                ptcl.wr((size_t)c.cmd_offset + 2u, di + 1u);
                                c.cmd_offset += 3u;
                                break;
                            }
                            case 0x248u: {
                                co::write_path(c, tile, tile_ix, draw_flags);
                                co::alloc_cmd(c, 2u);
                                ptcl.wr(c.cmd_offset, 9u);
                                ptcl.wr((size_t)c.cmd_offset + 1u, di + 1u);
                                c.cmd_offset += 2u;
                                break;
                            }
                            case 0x9u: {
                                if (tile.segment_count_or_ix == 0u && tile.backdrop == 0) {
                                    clip_zero_depth = clip_depth + 1u;
                                } else {
                                    co::alloc_cmd(c, 1u);
                                    ptcl.wr(c.cmd_offset, 10u);
                                    c.cmd_offset += 1u;
                                    render_blend_depth += 1u;
                                    max_blend_depth = umax_(max_blend_depth, render_blend_depth);
                                }
                                clip_depth += 1u;
                                break;
                            }
                            case 0x21u: {
                                clip_depth -= 1u;
                                co::write_path(c, tile, tile_ix, 0u);
                                uint32_t blend = scene.rd(dd);
                                uint32_t alpha = scene.rd((size_t)dd + 1u);
                                co::alloc_cmd(c, 3u);
                                ptcl.wr(c.cmd_offset, 11u);
                                ptcl.wr((size_t)c.cmd_offset + 1u, blend);
                                ptcl.wr((size_t)c.cmd_offset + 2u, alpha);
                                c.cmd_offset += 3u;
                                render_blend_depth -= 1u;
                                break;
                            }
                            default: break;
                        }
                    } else {
                        if (tag == 0x9u) {
                            clip_depth += 1u;
                        } else if (tag == 0x21u) {
                            if (clip_depth == clip_zero_depth) clip_zero_depth = 0u;
                            clip_depth -= 1u;
                        }
                    }
                }
                if (bin_tile_x + tile_x < cfg.width_in_tiles && bin_tile_y + tile_y < cfg.height_in_tiles) {
                    ptcl.wr(c.cmd_offset, 0u);
                    uint32_t blend_ix = 0u;
                    if (max_blend_depth > 4u) {
                        uint32_t scratch_size = (max_blend_depth - 4u) * 256u;
                        blend_ix = bump->blend;
                        bump->blend += scratch_size;
                        if (blend_ix + scratch_size > cfg.blend_size) bump->failed |= STAGE_COARSE;
                    }
                    ptcl.wr(blend_offset, blend_ix);
                }
            }
    };
    if (!(g_oracle_parallel_alloc && g_oracle_threads > 1)) {
        for (uint32_t wy = 0; wy < n_wg_y; wy++)
            for (uint32_t wx = 0; wx < n_wg_x; wx++) one_bin(wx, wy, bump, ptcl, false);
        return;
    }
    // count -> scan -> write over the bins in their canonical (row-major) order: PTCL chunks, segments and blend space
    const uint32_t n_bins = n_wg_x * n_wg_y;
    std::vector<Bump> need(n_bins + 1u);
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_oracle_threads)
    for (uint32_t bi = 0; bi < n_bins; bi++) {
        Bump local = {};
        View<uint32_t> none = ptcl;
        none.n = 0;  // nothing is written
        one_bin(bi % n_wg_x, bi / n_wg_x, &local, none, true);
        need[bi + 1u] = local;
    }
    need[0] = *bump;
    for (uint32_t bi = 0; bi < n_bins; bi++) {
        need[bi + 1u].ptcl += need[bi].ptcl;
        need[bi + 1u].segments += need[bi].segments;
        need[bi + 1u].blend += need[bi].blend;
    }
    uint32_t failed = 0u;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_oracle_threads) reduction(| : failed)
    for (uint32_t bi = 0; bi < n_bins; bi++) {
        Bump local = {};
        local.ptcl = need[bi].ptcl; local.segments = need[bi].segments; local.blend = need[bi].blend;
        one_bin(bi % n_wg_x, bi / n_wg_x, &local, ptcl, false);
        failed |= local.failed;
    }
    bump->ptcl = need[n_bins].ptcl;
    bump->segments = need[n_bins].segments;
    bump->blend = need[n_bins].blend;
    bump->failed |= failed;
}

// path_tiling_setup.wgsl:20-32 -- [bump, indirect, ptcl]
static void path_tiling_setup(OBuf* b) {
    Bump* bump = (Bump*)b[0].p;
    Indirect* ind = (Indirect*)b[1].p;
    View<uint32_t> ptcl(b[2]);
    if (bump->failed != 0u) { ind->x = 0u; ptcl.wr(0, ~0u); }
    else ind->x = (bump->seg_counts + (WG - 1u)) / WG;
    ind->y = 1u; ind->z = 1u;
}


// path_tiling.wgsl:39-173 -- [bump, seg_counts, lines, paths, tiles, segments]
static void path_tiling(uint32_t n_wg, OBuf* b) {
    Bump* bump = (Bump*)b[0].p;
    View<SegmentCount> seg_counts(b[1]);
    View<LineSoup> lines(b[2]);
    View<Path> paths(b[3]);
    View<Tile> tiles(b[4]);
    View<Segment> segments(b[5]);
    uint32_t n_segments = bump->seg_counts;
#pragma omp parallel for schedule(static) num_threads(g_oracle_threads)
    for (uint32_t gid = 0; gid < n_wg * WG; gid++) {
        if (!(gid < n_segments)) continue;
        SegmentCount sc = seg_counts.rd(gid);
        LineSoup line = lines.rd(sc.line_ix);
        uint32_t seg_within_slice = sc.counts >> 16;
        uint32_t seg_within_line = sc.counts & 0xffffu;
        V2 lp0 = v2(line.p0[0], line.p0[1]), lp1 = v2(line.p1[0], line.p1[1]);
        bool is_down = lp1.y >= lp0.y;
        V2 xy0 = is_down ? lp0 : lp1;
        V2 xy1 = is_down ? lp1 : lp0;
        V2 s0 = xy0 * TILE_SCALE;
        V2 s1 = xy1 * TILE_SCALE;
        uint32_t count_x = span(s0.x, s1.x) - 1u;
        uint32_t count = count_x + span(s0.y, s1.y);
        float dx = abs_(s1.x - s0.x);
        float dy = s1.y - s0.y;
        float idxdy = 1.0f / (dx + dy);
        float a = dx * idxdy;
        bool is_positive_slope = s1.x >= s0.x;
        float x_sign = is_positive_slope ? 1.0f : -1.0f;
        float xt0 = floor_(s0.x * x_sign);
        float c = s0.x * x_sign - xt0;
        float y0i = floor_(s0.y);
        float ytop = (s0.y == s1.y) ? ceil_(s0.y) : (y0i + 1.0f);
        float bb = fmin_((dy * c + dx * (ytop - s0.y)) * idxdy, ONE_MINUS_ULP);
        float robust_err = floor_(a * ((float)count - 1.0f) + bb) - (float)count_x;
        if (robust_err != 0.0f) a -= ROBUST_EPSILON * sign_(robust_err);
        int32_t x0i = to_i32(xt0 * x_sign + 0.5f * (x_sign - 1.0f));
        float z = floor_(a * (float)seg_within_line + bb);
        int32_t x = x0i + to_i32(x_sign * z);
        int32_t y = to_i32(y0i + (float)seg_within_line - z);
        Path path = paths.rd(line.path_ix);
        int32_t bbox[4] = {(int32_t)path.bbox[0], (int32_t)path.bbox[1], (int32_t)path.bbox[2], (int32_t)path.bbox[3]};
        int32_t stride = bbox[2] - bbox[0];
        int32_t tile_ix = (int32_t)path.tiles + (y - bbox[1]) * stride + x - bbox[0];
        Tile tile = tiles.rd((size_t)(uint32_t)tile_ix);
        uint32_t seg_start = ~tile.segment_count_or_ix;
        if ((int32_t)seg_start < 0) continue;
        V2 tile_xy = v2((float)x * 16.0f, (float)y * 16.0f);
        V2 tile_xy1 = tile_xy + v2(16.0f, 16.0f);
        if (seg_within_line > 0u) {
            float z_prev = floor_(a * ((float)seg_within_line - 1.0f) + bb);
            if (z == z_prev) {
                float xt = xy0.x + (xy1.x - xy0.x) * (tile_xy.y - xy0.y) / (xy1.y - xy0.y);
                xt = clamp_(xt, tile_xy.x + 1e-3f, tile_xy1.x);
                xy0 = v2(xt, tile_xy.y);
            } else {
                float x_clip = is_positive_slope ? tile_xy.x : tile_xy1.x;
                float yt = xy0.y + (xy1.y - xy0.y) * (x_clip - xy0.x) / (xy1.x - xy0.x);
                yt = clamp_(yt, tile_xy.y + 1e-3f, tile_xy1.y);
                xy0 = v2(x_clip, yt);
            }
        }
        if (seg_within_line < count - 1u) {
            float z_next = floor_(a * ((float)seg_within_line + 1.0f) + bb);
            if (z == z_next) {
                float xt = xy0.x + (xy1.x - xy0.x) * (tile_xy1.y - xy0.y) / (xy1.y - xy0.y);
                xt = clamp_(xt, tile_xy.x + 1e-3f, tile_xy1.x);
                xy1 = v2(xt, tile_xy1.y);
            } else {
                float x_clip = is_positive_slope ? tile_xy1.x : tile_xy.x;
                float yt = xy0.y + (xy1.y - xy0.y) * (x_clip - xy0.x) / (xy1.x - xy0.x);
                yt = clamp_(yt, tile_xy.y + 1e-3f, tile_xy1.y);
                xy1 = v2(x_clip, yt);
            }
        }
        float y_edge = 1e9f;
        V2 p0 = xy0 - tile_xy;
        V2 p1 = xy1 - tile_xy;
        const float EPSILON = 1e-6f;
        if (p0.x == 0.0f) {
            if (p1.x == 0.0f) {
                p0.x = EPSILON;
                if (p0.y == 0.0f) {
                    p1.x = EPSILON;
                    p1.y = 16.0f;
                } else {
                    p1.x = 2.0f * EPSILON;
                    p1.y = p0.y;
                }
            } else if (p0.y == 0.0f) {
                p0.x = EPSILON;
            } else {
                y_edge = p0.y;
            }
        } else if (p1.x == 0.0f) {
            if (p1.y == 0.0f) p1.x = EPSILON; else y_edge = p1.y;
        }
        if (p0.x == floor_(p0.x) && p0.x != 0.0f) p0.x -= EPSILON;
        if (p1.x == floor_(p1.x) && p1.x != 0.0f) p1.x -= EPSILON;
        if (!is_down) { V2 tmp = p0; p0 = p1; p1 = tmp; }
        Segment seg;
        seg.p0[0] = p0.x; seg.p0[1] = p0.y; seg.p1[0] = p1.x; seg.p1[1] = p1.y; seg.y_edge = y_edge; seg.pad = 0;
        segments.wr((size_t)seg_start + seg_within_slice, seg);
    }
}

// ---------------------------------------------------------------------------------------------
// shared/blend.wgsl
// ---------------------------------------------------------------------------------------------
namespace bl {
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 screen(V3 cb, V3 cs) { return v3(cb.x + cs.x - (cb.x * cs.x), cb.y + cs.y - (cb.y * cs.y), cb.z + cs.z - (cb.z * cs.z)); }
static inline float color_dodge(float cb, float cs) {
    if (cb == 0.0f) return 0.0f; else if (cs == 1.0f) return 1.0f; else return fmin_(1.0f, cb / (1.0f - cs));
}
static inline float color_burn(float cb, float cs) {
    if (cb == 1.0f) return 1.0f; else if (cs == 0.0f) return 0.0f; else return 1.0f - fmin_(1.0f, (1.0f - cb) / cs);
}
static inline float hard_light1(float cb, float cs) {
    float scr_cs = 2.0f * cs - 1.0f;
    float a = cb + scr_cs - (cb * scr_cs);
    float bb = cb * 2.0f * cs;
    return (cs <= 0.5f) ? bb : a;
}
static inline V3 hard_light(V3 cb, V3 cs) { return v3(hard_light1(cb.x, cs.x), hard_light1(cb.y, cs.y), hard_light1(cb.z, cs.z)); }
static inline float soft_light1(float cb, float cs) {
    float d = (cb <= 0.25f) ? (((16.0f * cb - 12.0f) * cb + 4.0f) * cb) : sqrt_(cb);
    float t = cb + (2.0f * cs - 1.0f) * (d - cb);
    float f = cb - (1.0f - 2.0f * cs) * cb * (1.0f - cb);
    return (cs <= 0.5f) ? f : t;
}
static inline V3 soft_light(V3 cb, V3 cs) { return v3(soft_light1(cb.x, cs.x), soft_light1(cb.y, cs.y), soft_light1(cb.z, cs.z)); }
static inline float sat(V3 c) { return fmax_(c.x, fmax_(c.y, c.z)) - fmin_(c.x, fmin_(c.y, c.z)); }
static inline float lum(V3 c) { return c.x * 0.3f + c.y * 0.59f + c.z * 0.11f; }
static inline V3 clip_color(V3 c) {
    float l = lum(c);
    float n = fmin_(c.x, fmin_(c.y, c.z));
    float x = fmax_(c.x, fmax_(c.y, c.z));
    if (n < 0.0f) c = v3(l + (((c.x - l) * l) / (l - n)), l + (((c.y - l) * l) / (l - n)), l + (((c.z - l) * l) / (l - n)));
    if (x > 1.0f) c = v3(l + (((c.x - l) * (1.0f - l)) / (x - l)), l + (((c.y - l) * (1.0f - l)) / (x - l)), l + (((c.z - l) * (1.0f - l)) / (x - l)));
    return c;
}
static inline V3 set_lum(V3 c, float l) { float d = l - lum(c); return clip_color(v3(c.x + d, c.y + d, c.z + d)); }
static inline void set_sat_inner(float* cmin, float* cmid, float* cmax, float s) {
    if (*cmax > *cmin) { *cmid = ((*cmid - *cmin) * s) / (*cmax - *cmin); *cmax = s; }
    else { *cmid = 0.0f; *cmax = 0.0f; }
    *cmin = 0.0f;
}
static inline V3 set_sat(V3 c, float s) {
    float r = c.x, g = c.y, b = c.z;
    if (r <= g) {
        if (g <= b) set_sat_inner(&r, &g, &b, s);
        else { if (r <= b) set_sat_inner(&r, &b, &g, s); else set_sat_inner(&b, &r, &g, s); }
    } else {
        if (r <= b) set_sat_inner(&g, &r, &b, s);
        else { if (g <= b) set_sat_inner(&g, &b, &r, s); else set_sat_inner(&b, &g, &r, s); }
    }
    return v3(r, g, b);
}
static V3 blend_mix(V3 cb, V3 cs, uint32_t mode) {  // blend.wgsl:142-195
    switch (mode) {
        case 1: return v3(cb.x * cs.x, cb.y * cs.y, cb.z * cs.z);
        case 2: return screen(cb, cs);
        case 3: return hard_light(cs, cb);
        case 4: return v3(fmin_(cb.x, cs.x), fmin_(cb.y, cs.y), fmin_(cb.z, cs.z));
        case 5: return v3(fmax_(cb.x, cs.x), fmax_(cb.y, cs.y), fmax_(cb.z, cs.z));
        case 6: return v3(color_dodge(cb.x, cs.x), color_dodge(cb.y, cs.y), color_dodge(cb.z, cs.z));
        case 7: return v3(color_burn(cb.x, cs.x), color_burn(cb.y, cs.y), color_burn(cb.z, cs.z));
        case 8: return hard_light(cb, cs);
        case 9: return soft_light(cb, cs);
        case 10: return v3(abs_(cb.x - cs.x), abs_(cb.y - cs.y), abs_(cb.z - cs.z));
        case 11: return v3(cb.x + cs.x - 2.0f * cb.x * cs.x, cb.y + cs.y - 2.0f * cb.y * cs.y, cb.z + cs.z - 2.0f * cb.z * cs.z);
        case 12: return set_lum(set_sat(cs, sat(cb)), lum(cb));
        case 13: return set_lum(set_sat(cb, sat(cs)), lum(cb));
        case 14: return set_lum(cs, lum(cb));
        case 15: return set_lum(cb, lum(cs));
        default: return cs;
    }
}
static V4 blend_compose(V3 cb, V3 cs, float ab, float as_, uint32_t mode) {  // blend.wgsl:216-284
    float fa = 0.0f, fb = 0.0f;
    switch (mode) {
        case 1: fa = 1.0f; fb = 0.0f; break;
        case 2: fa = 0.0f; fb = 1.0f; break;
        case 0: fa = 1.0f; fb = 1.0f - as_; break;
        case 4: fa = 1.0f - ab; fb = 1.0f; break;
        case 5: fa = ab; fb = 0.0f; break;
        case 6: fa = 0.0f; fb = as_; break;
        case 7: fa = 1.0f - ab; fb = 0.0f; break;
        case 8: fa = 0.0f; fb = 1.0f - as_; break;
        case 9: fa = ab; fb = 1.0f - as_; break;
        case 10: fa = 1.0f - ab; fb = as_; break;
        case 11: fa = 1.0f - ab; fb = 1.0f - as_; break;
        case 12: fa = 1.0f; fb = 1.0f; break;
        case 13:
            return V4{fmin_(1.0f, as_ * cs.x + ab * cb.x), fmin_(1.0f, as_ * cs.y + ab * cb.y), fmin_(1.0f, as_ * cs.z + ab * cb.z),
                      fmin_(1.0f, as_ + ab)};
        default: break;
    }
    float as_fa = as_ * fa;
    float ab_fb = ab * fb;
    return V4{as_fa * cs.x + ab_fb * cb.x, as_fa * cs.y + ab_fb * cb.y, as_fa * cs.z + ab_fb * cb.z, fmin_(as_fa + ab_fb, 1.0f)};
}
static V4 blend_mix_compose(V4 backdrop, V4 src, uint32_t mode) {  // blend.wgsl:288-310
    const float EPSILON = 1e-15f;
    if ((mode & 0x7fffu) == 0u) {
        float k = 1.0f - src.w;
        return V4{backdrop.x * k + src.x, backdrop.y * k + src.y, backdrop.z * k + src.z, backdrop.w * k + src.w};
    }
    float inv_src_a = 1.0f / fmax_(src.w, EPSILON);
    V3 cs = v3(src.x * inv_src_a, src.y * inv_src_a, src.z * inv_src_a);
    float inv_backdrop_a = 1.0f / fmax_(backdrop.w, EPSILON);
    V3 cb = v3(backdrop.x * inv_backdrop_a, backdrop.y * inv_backdrop_a, backdrop.z * inv_backdrop_a);
    uint32_t mix_mode = mode >> 8;
    V3 mixed = blend_mix(cb, cs, mix_mode);
    cs = v3(mix_(cs.x, mixed.x, backdrop.w), mix_(cs.y, mixed.y, backdrop.w), mix_(cs.z, mixed.z, backdrop.w));
    uint32_t compose_mode = mode & 0xffu;
    if (compose_mode == 0u) {
        return V4{mix_(backdrop.x, cs.x, src.w), mix_(backdrop.y, cs.y, src.w), mix_(backdrop.z, cs.z, src.w),
                  src.w + backdrop.w * (1.0f - src.w)};
    }
    return blend_compose(cb, cs, backdrop.w, src.w, compose_mode);
}
}  // namespace bl

// ---------------------------------------------------------------------------------------------
// fine.wgsl (area AA, `full` permutation): :824-878 fill_path, :883-1103 main
// [config, segments, ptcl, info, blend_spill, output(rgba16f, W*H*4 u16), gradients(rgba16f 512xH), image_atlas?]
// Each WGSL invocation owns 4 horizontally adjacent pixels; arithmetic is restated per invocation.
// ---------------------------------------------------------------------------------------------
static float extend_mode(float t, uint32_t mode) {  // fine.wgsl:800-812
    switch (mode) {
        case 0: return clamp_(t, 0.0f, 1.0f);
        case 1: return fract_(t);
        default: return abs_(t - 2.0f * round_(0.5f * t));
    }
}
#include "srgb_lut.h"
struct ImageDesc { uint64_t offset_px; uint32_t width, height; };  // oracle-side image table entry; height bit 31 = sRGB texels


// ------------------------------------------------------------------------------------------------
// fill_path_ms / fill_path_ms_evenodd (fine.wgsl:148-711): multisampled coverage of one 16x16 tile for one
// CMD_FILL, evaluated for the whole workgroup at once (the WGSL is workgroup-cooperative; its result for an
// invocation depends only on the fill and the segments).  SWAR integer arithmetic; every atomic is a
// commutative add/xor, so the sequential order here gives the same words.  WGSL rules applied: shifts take
// the amount modulo 32, out-of-range workgroup/storage indices read zero and drop writes, float->int
// conversions saturate.  SAMPLES is 8 or 16; mask_lut per renderer/mask.go:43-105.
// ------------------------------------------------------------------------------------------------
static inline uint32_t shl32(uint32_t v, uint32_t s) { return v << (s & 31u); }
static inline uint32_t shr32(uint32_t v, uint32_t s) { return v >> (s & 31u); }

struct MsTile {
    uint32_t sh_count[64];
    uint32_t sh_winding_y[4], sh_winding_y_prefix[4];
    uint32_t sh_winding[64];
    uint32_t sh_samples[1024];
};

static void fill_path_ms_tile(int SAMPLES, uint32_t size_and_rule, uint32_t seg_data, int32_t backdrop, const View<Segment>& segments,
                              const View<uint32_t>& mask_lut, float* area /*[256], pixel = y*16+x*/) {
    const bool even_odd = (size_and_rule & 1u) != 0u;
    const uint32_t n_segs = size_and_rule >> 1;
    const uint32_t MASK_WIDTH = SAMPLES == 8 ? 32u : 64u, MASK_HEIGHT = MASK_WIDTH;
    const uint32_t WORDS = even_odd ? 1u : (SAMPLES == 8 ? 2u : 4u);  // sh_samples words per pixel
    const uint32_t SH_SAMPLES_SIZE = SAMPLES == 8 ? 512u : 1024u;
    const uint32_t FULL = SAMPLES == 8 ? 0xffu : 0xffffu;
    MsTile T;
    if (even_odd) {
        T.sh_winding_y[0] = 0u;
        for (int i = 0; i < 16; i++) T.sh_winding[i] = 0u;
        for (uint32_t i = 0; i < 256u; i++) T.sh_samples[i] = 0u;
    } else {
        for (int i = 0; i < 4; i++) T.sh_winding_y[i] = 0x80808080u;
        for (int i = 0; i < 64; i++) T.sh_winding[i] = 0x80808080u;
        for (uint32_t i = 0; i < 256u * WORDS; i++) T.sh_samples[i] = 0x80808080u;
    }
    auto samples_add = [&](uint32_t ix, uint32_t v) { if (ix < SH_SAMPLES_SIZE) T.sh_samples[ix] += v; };
    auto samples_xor = [&](uint32_t ix, uint32_t v) { if (ix < SH_SAMPLES_SIZE) T.sh_samples[ix] ^= v; };
    const uint32_t n_batch = (n_segs + 63u) / 64u;
    for (uint32_t batch = 0; batch < n_batch; batch++) {
        const uint32_t slice_size = std::min(n_segs - batch * 64u, 64u);
        for (uint32_t th = 0; th < 64u; th++) {  // fine.wgsl:176-203 / :532-555
            uint32_t count = 0u;
            if (th < slice_size) {
                Segment seg = segments.rd((size_t)seg_data + batch * 64u + th);
                float x0 = seg.p0[0], y0 = seg.p0[1], x1 = seg.p1[0], y1 = seg.p1[1];
                float y_edge_f = 16.0f;
                int32_t delta = (x1 <= x0) ? 1 : -1;
                if (x0 == 0.0f) y_edge_f = y0;
                else if (x1 == 0.0f) y_edge_f = y1;
                if (!(y0 == y1 && y0 == floor_(y0))) count = span(x0, x1) + span(y0, y1) - 1u;
                uint32_t y_edge = to_u32(ceil_(y_edge_f));
                if (y_edge < 16u) {
                    if (even_odd) T.sh_winding_y[0] ^= shl32(1u, y_edge);
                    else T.sh_winding_y[y_edge >> 2] += shl32((uint32_t)delta, (y_edge & 3u) << 3);
                }
            }
            T.sh_count[th] = count;
        }
        for (uint32_t th = 1; th < slice_size; th++) T.sh_count[th] += T.sh_count[th - 1u];  // inclusive prefix, :205-215
        const uint32_t total = T.sh_count[slice_size - 1u];
        for (uint32_t i = 0; i < total; i++) {  // :217-383 / :566-675
            uint32_t lo = 0u, hi = slice_size;
            while (hi > lo + 1u) {
                uint32_t mid = (lo + hi) >> 1;
                if (i >= T.sh_count[mid - 1u]) lo = mid; else hi = mid;
            }
            const uint32_t el_ix = lo;
            const bool last_pixel = i + 1u == T.sh_count[el_ix];
            const uint32_t sub_ix = i - (el_ix > 0u ? T.sh_count[el_ix - 1u] : 0u);
            Segment seg = segments.rd((size_t)seg_data + batch * 64u + el_ix);
            V2 in0{seg.p0[0], seg.p0[1]}, in1{seg.p1[0], seg.p1[1]};
            const bool is_down = in1.y >= in0.y;
            const V2 xy0 = is_down ? in0 : in1, xy1 = is_down ? in1 : in0;
            const float dx = abs_(xy1.x - xy0.x);
            const float dy = xy1.y - xy0.y;
            const float idxdy = 1.0f / (dx + dy);
            float a = dx * idxdy;
            const bool is_positive_slope = xy1.x >= xy0.x;
            const float x_sign = is_positive_slope ? 1.0f : -1.0f;
            const float xt0 = floor_(xy0.x * x_sign);
            const float c = xy0.x * x_sign - xt0;
            const float y0i = floor_(xy0.y);
            const float ytop = y0i + 1.0f;
            const float b = fmin_((dy * c + dx * (ytop - xy0.y)) * idxdy, ONE_MINUS_ULP);
            const uint32_t count_x = span(xy0.x, xy1.x) - 1u;
            const uint32_t count = count_x + span(xy0.y, xy1.y);
            const float robust_err = floor_(a * ((float)count - 1.0f) + b) - (float)count_x;
            if (robust_err != 0.0f) a -= ROBUST_EPSILON * sign_(robust_err);
            const int32_t x0i = to_i32(xt0 * x_sign + 0.5f * (x_sign - 1.0f));
            const float zf = a * (float)sub_ix + b;
            const float z = floor_(zf);
            const int32_t x = x0i + to_i32(x_sign * z);
            const int32_t y = (int32_t)((uint32_t)to_i32(y0i) + sub_ix - (uint32_t)to_i32(z));
            bool is_delta, is_bump = false;
            const float zp = floor_(a * (float)(sub_ix - 1u) + b);
            if (sub_ix == 0u) {
                is_delta = y0i == xy0.y;
                is_bump = even_odd ? (xy0.x == 0.0f) : (xy0.x == 0.0f && y0i != xy0.y);
            } else {
                is_delta = z == zp;
                is_bump = is_positive_slope && !is_delta;
            }
            const uint32_t pix_ix = (uint32_t)y * 16u + (uint32_t)x;
            if ((uint32_t)x < 15u && (uint32_t)y < 16u) {
                if (is_delta) {
                    if (even_odd) {
                        T.sh_winding[y] ^= shl32(2u, (uint32_t)x);
                    } else {
                        const uint32_t delta_pix = pix_ix + 1u;
                        T.sh_winding[delta_pix >> 2] += shl32(is_down ? 1u : 0xffffffffu, (delta_pix & 3u) << 3);
                    }
                }
            }
            const uint32_t mask_block = (is_positive_slope ? 1u : 0u) * (MASK_WIDTH * MASK_HEIGHT / 2u);
            const float half_height = (float)(MASK_HEIGHT / 2u);
            const float mask_row = floor_(fmin_(a * half_height, half_height - 1.0f)) * (float)MASK_WIDTH;
            const float mask_col = floor_((zf - z) * (float)MASK_WIDTH);
            const uint32_t mask_ix = mask_block + to_u32(mask_row + mask_col);
            uint32_t mask;
            if (SAMPLES == 8) mask = shr32(mask_lut.rd(mask_ix / 4u), (mask_ix % 4u) * 8u) & 0xffu;
            else mask = shr32(mask_lut.rd(mask_ix / 2u), (mask_ix % 2u) * 16u) & 0xffffu;
            const float sf = (float)SAMPLES;
            if (sub_ix == 0u && !is_bump) mask &= shl32(FULL, to_u32(round_(sf * (xy0.y - (float)y))));
            if (last_pixel && xy1.x != 0.0f) mask &= ~shl32(FULL, to_u32(round_(sf * (xy1.y - (float)y))));
            if (even_odd) {
                if (is_bump) mask ^= FULL;
                samples_xor(pix_ix, mask);
            } else {
                const uint32_t bump_delta = is_down ? 0x1010101u : (uint32_t)-0x1010101;
                for (uint32_t half = 0; half < (SAMPLES == 8 ? 1u : 2u); half++) {
                    const uint32_t m8 = (mask >> (8u * half)) & 0xffu;
                    const uint32_t ma = m8 ^ (m8 << 7);
                    const uint32_t mb = ma ^ (ma << 14);
                    const uint32_t e0 = mb & 0x1010101u, e1 = (mb >> 4) & 0x1010101u;
                    uint32_t s0 = is_down ? (uint32_t)(-(int32_t)e0) : e0, s1 = is_down ? (uint32_t)(-(int32_t)e1) : e1;
                    if (is_bump) { s0 += bump_delta; s1 += bump_delta; }
                    samples_add(pix_ix * WORDS + 2u * half, s0);
                    samples_add(pix_ix * WORDS + 2u * half + 1u, s1);
                }
            }
        }
    }
    // resolve (:386-501 / :677-710)
    if (even_odd) {
        uint32_t scan_y = T.sh_winding_y[0];
        scan_y ^= scan_y << 1; scan_y ^= scan_y << 2; scan_y ^= scan_y << 4; scan_y ^= scan_y << 8;
        for (uint32_t ly = 0; ly < 16u; ly++) {
            uint32_t scan_x = T.sh_winding[ly];
            scan_x ^= scan_x << 1; scan_x ^= scan_x << 2; scan_x ^= scan_x << 4; scan_x ^= scan_x << 8;
            const uint32_t row_parity = (scan_y >> ly) ^ (uint32_t)backdrop;
            for (uint32_t px = 0; px < 16u; px++) {
                const uint32_t pix_ix = ly * 16u + px;
                const uint32_t samples = T.sh_samples[pix_ix];
                const uint32_t pix_parity = row_parity ^ (scan_x >> (pix_ix % 16u));
                const uint32_t pix_mask = (uint32_t)(-(int32_t)(pix_parity & 1u));
                area[pix_ix] = (float)__builtin_popcount((samples ^ pix_mask) & FULL) * (SAMPLES == 8 ? 0.125f : 0.0625f);
            }
        }
        return;
    }
    uint32_t packed_w_th[64], wind_y_th[64];
    for (uint32_t th = 0; th < 64u; th++) {
        const uint32_t lx = th & 3u, ly = th >> 2;
        uint32_t packed_w = T.sh_winding[th];  // major == th
        packed_w += (packed_w - 0x808080u) << 8;
        packed_w += (packed_w - 0x8080u) << 16;
        uint32_t packed_y = T.sh_winding_y[ly >> 2];
        packed_y += (packed_y - 0x808080u) << 8;
        packed_y += (packed_y - 0x8080u) << 16;
        const uint32_t wind_y = (packed_y >> ((ly & 3u) << 3)) - 0x80u;
        if ((ly & 3u) == 3u && lx == 0u) T.sh_winding_y_prefix[ly >> 2] = wind_y;
        packed_w_th[th] = packed_w;
        wind_y_th[th] = wind_y;
    }
    uint32_t prefix_x_th[64];
    for (uint32_t th = 0; th < 64u; th++) prefix_x_th[th] = ((packed_w_th[th] >> 24) - 0x80u) * 0x1010101u;  // stored to sh_winding[major]
    for (uint32_t th = 0; th < 64u; th++) {
        const uint32_t ly = th >> 2;
        uint32_t packed_w = packed_w_th[th];
        for (uint32_t i = (th & ~3u); i < th; i++) packed_w += prefix_x_th[i];
        uint32_t wind_y = wind_y_th[th];
        for (uint32_t i = 0; i < (ly >> 2); i++) wind_y += T.sh_winding_y_prefix[i];
        for (uint32_t i = 0; i < 4u; i++) {
            const uint32_t pix_ix = th * 4u + i;
            const uint32_t expected_zero = (((packed_w >> (i * 8u)) + wind_y) & 0xffu) - (uint32_t)backdrop;
            if (expected_zero >= 256u) {
                area[pix_ix] = 1.0f;
            } else if (SAMPLES == 8) {
                const uint32_t samples0 = T.sh_samples[pix_ix * 2u], samples1 = T.sh_samples[pix_ix * 2u + 1u];
                const uint32_t xored0 = (expected_zero * 0x1010101u) ^ samples0;
                const uint32_t xored0_2 = xored0 | (xored0 * 2u);
                const uint32_t xored1 = (expected_zero * 0x1010101u) ^ samples1;
                const uint32_t xored1_2 = xored1 | (xored1 >> 1);
                const uint32_t xored2 = (xored0_2 & 0xAAAAAAAAu) | (xored1_2 & 0x55555555u);
                const uint32_t xored4 = xored2 | (xored2 * 4u);
                const uint32_t xored8 = xored4 | (xored4 * 16u);
                area[pix_ix] = (float)__builtin_popcount(xored8 & 0xC0C0C0C0u) * 0.125f;
            } else {
                const uint32_t e = expected_zero * 0x1010101u;
                const uint32_t xored0 = e ^ T.sh_samples[pix_ix * 4u], xored1 = e ^ T.sh_samples[pix_ix * 4u + 1u];
                const uint32_t xored2 = e ^ T.sh_samples[pix_ix * 4u + 2u], xored3 = e ^ T.sh_samples[pix_ix * 4u + 3u];
                const uint32_t xored0_2 = xored0 | (xored0 * 2u), xored1_2 = xored1 | (xored1 >> 1);
                const uint32_t xored01 = (xored0_2 & 0xAAAAAAAAu) | (xored1_2 & 0x55555555u);
                const uint32_t xored01_4 = xored01 | (xored01 * 4u);
                const uint32_t xored2_2 = xored2 | (xored2 * 2u), xored3_2 = xored3 | (xored3 >> 1);
                const uint32_t xored23 = (xored2_2 & 0xAAAAAAAAu) | (xored3_2 & 0x55555555u);
                const uint32_t xored23_4 = xored23 | (xored23 >> 2);
                const uint32_t xored4 = (xored01_4 & 0xCCCCCCCCu) | (xored23_4 & 0x33333333u);
                const uint32_t xored8 = xored4 | (xored4 * 16u);
                area[pix_ix] = (float)__builtin_popcount(xored8 & 0xF0F0F0F0u) * 0.0625f;
            }
        }
    }
}

extern "C" void oracle_set_threads(int n) { g_oracle_threads = n < 1 ? 1 : n; }
// EXPERIMENT switch (tools/fine_order_ulp.py; never set by a test or by bench.py): the association of fill_path's sum.
// 0: the WGSL's (the oracle proper).  1: the terms of a fill are summed on their own, the backdrop is added last -- what a
// per-(fill,row) segmented sum would compute.  2: the segments of a fill in reverse order.  Used to MEASURE what
// north_star's "within 1 ULP" would leave of the image if the f32 order of fine.wgsl:832-864 were given up.
static int g_oracle_fine_order = 0;
extern "C" void oracle_set_fine_order(int o) { g_oracle_fine_order = o; }
extern "C" void oracle_set_parallel_alloc(int on) { g_oracle_parallel_alloc = on != 0; }

// aa = 0: analytic area (fine_area); 8 / 16: fine_msaa8 / fine_msaa16 with the mask LUT as last binding
static void fine_area(uint32_t n_wg_x, uint32_t n_wg_y, OBuf* b, int nb, int aa = 0) {
    using bl::V4;
    const Config& cfg = *(Config*)b[0].p;
    View<Segment> segments(b[1]);
    View<uint32_t> ptcl(b[2]);
    View<uint32_t> info(b[3]);
    View<V4> blend_spill(b[4]);
    uint16_t* output = (uint16_t*)b[5].p;
    size_t output_px = (size_t)(b[5].n / 8);
    const uint16_t* gradients = nb > 6 ? (const uint16_t*)b[6].p : nullptr;
    size_t grad_px = nb > 6 ? (size_t)(b[6].n / 8) : 0;
    // images: b[7] = table of ImageDesc, b[8] = rgba8 texels (already linear, premul applied in shader)
    View<ImageDesc> img_table(nb > 8 ? b[7] : OBuf{nullptr, 0});
    const uint8_t* img_px = nb > 8 ? (const uint8_t*)b[8].p : nullptr;
    size_t img_n = nb > 8 ? (size_t)(b[8].n / 4) : 0;
    View<uint32_t> mask_lut((aa != 0 && nb > 9) ? b[9] : OBuf{nullptr, 0});
    if (ptcl.rd(0) == ~0u) return;
    auto load_grad = [&](int32_t x, int32_t y) {
        size_t ix = (size_t)y * 512u + (size_t)x;
        if (x < 0 || x >= 512 || y < 0 || ix >= grad_px) return V4{0, 0, 0, 0};
        const uint16_t* t = gradients + ix * 4;
        return V4{f16_to_f32(t[0]), f16_to_f32(t[1]), f16_to_f32(t[2]), f16_to_f32(t[3])};
    };
    // Tiles are independent (disjoint pixels and blend_spill slices): rows of tiles may run on several host
    // threads (oracle_set_threads; default 1).  Results do not depend on the thread count.
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_oracle_threads)
    for (uint32_t wy = 0; wy < n_wg_y; wy++)
        for (uint32_t wx = 0; wx < n_wg_x; wx++) {
            uint32_t tile_ix = wy * cfg.width_in_tiles + wx;
            std::vector<float> ms_area;    // per tile: cached fill_path_ms results, 256 floats per CMD_FILL in stream order
            std::vector<uint32_t> ms_cmd;  // ... and the command index each belongs to
            for (uint32_t ly = 0; ly < 16u; ly++)
                for (uint32_t lx = 0; lx < 4u; lx++) {
                    uint32_t gx = wx * 4u + lx, gy = wy * 16u + ly;
                    float xyx = (float)(gx * 4u), xyy = (float)gy;
                    float lxyx = (float)(lx * 4u), lxyy = (float)ly;
                    V4 rgba[4];
                    for (int i = 0; i < 4; i++) rgba[i] = V4{cfg.base_color[0], cfg.base_color[1], cfg.base_color[2], cfg.base_color[3]};
                    V4 blend_stack[4][4];
                    std::memset(blend_stack, 0, sizeof blend_stack);
                    uint32_t clip_depth = 0u;
                    float area[4] = {0, 0, 0, 0};
                    uint32_t cmd_ix = tile_ix * 64u;
                    uint32_t blend_offset = ptcl.rd(cmd_ix);
                    cmd_ix += 1u;
                    uint32_t guard = 0;
                    for (;;) {
                        uint32_t tag = ptcl.rd(cmd_ix);
                        if (tag == 0u) break;
                        if (++guard > (1u << 24)) break;  // malformed stream (WGSL would spin)
                        switch (tag) {
                            case 1u: {  // CMD_FILL -> fill_path
                                uint32_t size_and_rule = ptcl.rd((size_t)cmd_ix + 1u);
                                uint32_t seg_data = ptcl.rd((size_t)cmd_ix + 2u);
                                int32_t backdrop = (int32_t)ptcl.rd((size_t)cmd_ix + 3u);
                                uint32_t n_segs = size_and_rule >> 1;
                                bool even_odd = (size_and_rule & 1u) != 0u;
                                if (aa != 0) {  // fine.wgsl:916-917: workgroup-cooperative fill, evaluated once per tile and command
                                    size_t slot = 0;
                                    while (slot < ms_cmd.size() && ms_cmd[slot] != cmd_ix) slot++;
                                    if (slot == ms_cmd.size()) {
                                        ms_cmd.push_back(cmd_ix);
                                        ms_area.resize(ms_area.size() + 256u);
                                        fill_path_ms_tile(aa, size_and_rule, seg_data, backdrop, segments, mask_lut, ms_area.data() + slot * 256u);
                                    }
                                    for (int i = 0; i < 4; i++) area[i] = ms_area[slot * 256u + ly * 16u + lx * 4u + (uint32_t)i];
                                    cmd_ix += 4u;
                                    break;
                                }
                                float backdrop_f = (float)backdrop;
                                const int order = g_oracle_fine_order;
                                for (int i = 0; i < 4; i++) area[i] = order == 1 ? 0.0f : backdrop_f;
                                for (uint32_t s_ = 0; s_ < n_segs; s_++) {
                                    const uint32_t s = order == 2 ? n_segs - 1u - s_ : s_;
                                    Segment seg = segments.rd((size_t)seg_data + s);
                                    float y = seg.p0[1] - lxyy;
                                    float dlx = seg.p1[0] - seg.p0[0], dly = seg.p1[1] - seg.p0[1];
                                    float y0 = clamp_(y, 0.0f, 1.0f);
                                    float y1 = clamp_(y + dly, 0.0f, 1.0f);
                                    float dy = y0 - y1;
                                    if (dy != 0.0f) {
                                        float vec_y_recip = 1.0f / dly;
                                        float t0 = (y0 - y) * vec_y_recip;
                                        float t1 = (y1 - y) * vec_y_recip;
                                        float startx = seg.p0[0] - lxyx;
                                        float x0 = startx + t0 * dlx;
                                        float x1 = startx + t1 * dlx;
                                        float xmin0 = fmin_(x0, x1);
                                        float xmax0 = fmax_(x0, x1);
                                        for (int i = 0; i < 4; i++) {
                                            float i_f = (float)i;
                                            float xmin = fmin_(xmin0 - i_f, 1.0f) - 1.0e-6f;
                                            float xmax = xmax0 - i_f;
                                            float bb = fmin_(xmax, 1.0f);
                                            float cc = fmax_(bb, 0.0f);
                                            float d = fmax_(xmin, 0.0f);
                                            float a = (bb + 0.5f * (d * d - cc * cc) - xmin) / (xmax - xmin);
                                            area[i] += a * dy;
                                        }
                                    }
                                    float y_edge = sign_(dlx) * clamp_(lxyy - seg.y_edge + 1.0f, 0.0f, 1.0f);
                                    for (int i = 0; i < 4; i++) area[i] += y_edge;
                                }
                                if (order == 1) for (int i = 0; i < 4; i++) area[i] = backdrop_f + area[i];
                                if (even_odd) {
                                    for (int i = 0; i < 4; i++) { float a = area[i]; area[i] = abs_(a - 2.0f * round_(0.5f * a)); }
                                } else {
                                    for (int i = 0; i < 4; i++) area[i] = fmin_(abs_(area[i]), 1.0f);
                                }
                                cmd_ix += 4u;
                                break;
                            }
                            case 3u: for (int i = 0; i < 4; i++) area[i] = 1.0f; cmd_ix += 1u; break;
                            case 5u: {
                                V4 fg{u2f(ptcl.rd((size_t)cmd_ix + 1u)), u2f(ptcl.rd((size_t)cmd_ix + 2u)), u2f(ptcl.rd((size_t)cmd_ix + 3u)),
                                      u2f(ptcl.rd((size_t)cmd_ix + 4u))};
                                for (int i = 0; i < 4; i++) {
                                    V4 fg_i{fg.x * area[i], fg.y * area[i], fg.z * area[i], fg.w * area[i]};
                                    float k = 1.0f - fg_i.w;
                                    rgba[i] = V4{rgba[i].x * k + fg_i.x, rgba[i].y * k + fg_i.y, rgba[i].z * k + fg_i.z, rgba[i].w * k + fg_i.w};
                                }
                                cmd_ix += 5u;
                                break;
                            }
                            case 10u: {
                                if (clip_depth < 4u) {
                                    for (int i = 0; i < 4; i++) { blend_stack[clip_depth][i] = rgba[i]; rgba[i] = V4{0, 0, 0, 0}; }
                                } else {
                                    uint32_t blend_in_scratch = clip_depth - 4u;
                                    uint32_t local_tile_ix = lx * 4u + ly * 16u;
                                    uint32_t local_blend_start = blend_offset + blend_in_scratch * 256u + local_tile_ix;
                                    for (int i = 0; i < 4; i++) { blend_spill.wr((size_t)local_blend_start + i, rgba[i]); rgba[i] = V4{0, 0, 0, 0}; }
                                }
                                clip_depth += 1u;
                                cmd_ix += 1u;
                                break;
                            }
                            case 11u: {
                                uint32_t blend = ptcl.rd((size_t)cmd_ix + 1u);
                                float alpha = u2f(ptcl.rd((size_t)cmd_ix + 2u));
                                clip_depth -= 1u;
                                for (int i = 0; i < 4; i++) {
                                    V4 bg;
                                    if (clip_depth < 4u) {
                                        bg = blend_stack[clip_depth][i];
                                    } else {
                                        uint32_t blend_in_scratch = clip_depth - 4u;
                                        uint32_t local_tile_ix = lx * 4u + ly * 16u;
                                        uint32_t local_blend_start = blend_offset + blend_in_scratch * 256u + local_tile_ix;
                                        bg = blend_spill.rd((size_t)local_blend_start + i);
                                    }
                                    V4 fg{rgba[i].x * area[i] * alpha, rgba[i].y * area[i] * alpha, rgba[i].z * area[i] * alpha,
                                          rgba[i].w * area[i] * alpha};
                                    rgba[i] = bl::blend_mix_compose(bg, fg, blend);
                                }
                                cmd_ix += 3u;
                                break;
                            }
                            case 12u: cmd_ix = ptcl.rd((size_t)cmd_ix + 1u); break;
                            case 6u: {  // CMD_LIN_GRAD
                                uint32_t index_mode = ptcl.rd((size_t)cmd_ix + 1u);
                                uint32_t index = index_mode >> 2, ext = index_mode & 3u;
                                uint32_t io = ptcl.rd((size_t)cmd_ix + 2u);
                                float line_x = u2f(info.rd(io)), line_y = u2f(info.rd((size_t)io + 1u)), line_c = u2f(info.rd((size_t)io + 2u));
                                float d = line_x * xyx + line_y * xyy + line_c;
                                for (int i = 0; i < 4; i++) {
                                    float my_d = d + line_x * (float)i;
                                    int32_t x = to_i32(round_(extend_mode(my_d, ext) * 511.0f));
                                    V4 fg = load_grad(x, (int32_t)index);
                                    V4 fg_i{fg.x * area[i], fg.y * area[i], fg.z * area[i], fg.w * area[i]};
                                    float k = 1.0f - fg_i.w;
                                    rgba[i] = V4{rgba[i].x * k + fg_i.x, rgba[i].y * k + fg_i.y, rgba[i].z * k + fg_i.z, rgba[i].w * k + fg_i.w};
                                }
                                cmd_ix += 3u;
                                break;
                            }
                            case 7u: {  // CMD_RAD_GRAD
                                uint32_t index_mode = ptcl.rd((size_t)cmd_ix + 1u);
                                uint32_t index = index_mode >> 2, ext = index_mode & 3u;
                                uint32_t io = ptcl.rd((size_t)cmd_ix + 2u);
                                float m[4];
                                for (int k = 0; k < 4; k++) m[k] = u2f(info.rd((size_t)io + k));
                                float xl0 = u2f(info.rd((size_t)io + 4u)), xl1 = u2f(info.rd((size_t)io + 5u));
                                float focal_x = u2f(info.rd((size_t)io + 6u));
                                float radius = u2f(info.rd((size_t)io + 7u));
                                uint32_t flags_kind = info.rd((size_t)io + 8u);
                                uint32_t flags = flags_kind >> 3, kind = flags_kind & 7u;
                                bool is_strip = kind == 2u, is_circular = kind == 1u, is_focal_on_circle = kind == 3u;
                                bool is_swapped = (flags & 1u) != 0u;
                                float r1_recip = is_circular ? 0.0f : (1.0f / radius);
                                float less_scale = (is_swapped || (1.0f - focal_x) < 0.0f) ? -1.0f : 1.0f;
                                float t_sign = sign_(1.0f - focal_x);
                                for (int i = 0; i < 4; i++) {
                                    float mx = xyx + (float)i, my = xyy;
                                    float x = m[0] * mx + m[2] * my + xl0;
                                    float y = m[1] * mx + m[3] * my + xl1;
                                    float xx = x * x, yy = y * y;
                                    float t = 0.0f;
                                    bool is_valid = true;
                                    if (is_strip) {
                                        float a = radius - yy;
                                        t = sqrt_(a) + x;
                                        is_valid = a >= 0.0f;
                                    } else if (is_focal_on_circle) {
                                        t = (xx + yy) / x;
                                        is_valid = t >= 0.0f && x != 0.0f;
                                    } else if (radius > 1.0f) {
                                        t = sqrt_(xx + yy) - x * r1_recip;
                                    } else {
                                        float a = xx - yy;
                                        t = less_scale * sqrt_(a) - x * r1_recip;
                                        is_valid = a >= 0.0f && t >= 0.0f;
                                    }
                                    if (is_valid) {
                                        t = extend_mode(focal_x + t_sign * t, ext);
                                        t = is_swapped ? (1.0f - t) : t;
                                        int32_t gx2 = to_i32(round_(t * 511.0f));
                                        V4 fg = load_grad(gx2, (int32_t)index);
                                        V4 fg_i{fg.x * area[i], fg.y * area[i], fg.z * area[i], fg.w * area[i]};
                                        float k = 1.0f - fg_i.w;
                                        rgba[i] = V4{rgba[i].x * k + fg_i.x, rgba[i].y * k + fg_i.y, rgba[i].z * k + fg_i.z, rgba[i].w * k + fg_i.w};
                                    }
                                }
                                cmd_ix += 3u;
                                break;
                            }
                            case 8u: {  // CMD_SWEEP_GRAD
                                uint32_t index_mode = ptcl.rd((size_t)cmd_ix + 1u);
                                uint32_t index = index_mode >> 2, ext = index_mode & 3u;
                                uint32_t io = ptcl.rd((size_t)cmd_ix + 2u);
                                float m[4];
                                for (int k = 0; k < 4; k++) m[k] = u2f(info.rd((size_t)io + k));
                                float xl0 = u2f(info.rd((size_t)io + 4u)), xl1 = u2f(info.rd((size_t)io + 5u));
                                float t0 = u2f(info.rd((size_t)io + 6u)), t1 = u2f(info.rd((size_t)io + 7u));
                                float scale = 1.0f / (t1 - t0);
                                for (int i = 0; i < 4; i++) {
                                    float mx = xyx + (float)i, my = xyy;
                                    float x = m[0] * mx + m[2] * my + xl0;
                                    float y = m[1] * mx + m[3] * my + xl1;
                                    float xabs = abs_(x), yabs = abs_(y);
                                    float slope = fmin_(xabs, yabs) / fmax_(xabs, yabs);
                                    float s = slope * slope;
                                    float phi = slope * (0.15912117063999176025390625f +
                                                         s * (-5.185396969318389892578125e-2f +
                                                              s * (2.476101927459239959716796875e-2f + s * (-7.0547382347285747528076171875e-3f))));
                                    phi = (xabs < yabs) ? (0.25f - phi) : phi;
                                    phi = (x < 0.0f) ? (0.5f - phi) : phi;
                                    phi = (y < 0.0f) ? (1.0f - phi) : phi;
                                    phi = (phi != phi) ? 0.0f : phi;
                                    phi = (phi - t0) * scale;
                                    float t = extend_mode(phi, ext);
                                    int32_t ramp_x = to_i32(round_(t * 511.0f));
                                    V4 fg = load_grad(ramp_x, (int32_t)index);
                                    V4 fg_i{fg.x * area[i], fg.y * area[i], fg.z * area[i], fg.w * area[i]};
                                    float k = 1.0f - fg_i.w;
                                    rgba[i] = V4{rgba[i].x * k + fg_i.x, rgba[i].y * k + fg_i.y, rgba[i].z * k + fg_i.z, rgba[i].w * k + fg_i.w};
                                }
                                cmd_ix += 3u;
                                break;
                            }
                            case 9u: {  // CMD_IMAGE
                                uint32_t io = ptcl.rd((size_t)cmd_ix + 1u);
                                float m[4];
                                for (int k = 0; k < 4; k++) m[k] = u2f(info.rd((size_t)io + k));
                                float xl0 = u2f(info.rd((size_t)io + 4u)), xl1 = u2f(info.rd((size_t)io + 5u));
                                uint32_t index = info.rd((size_t)io + 6u);
                                uint32_t width_height = info.rd((size_t)io + 7u);
                                float ew = (float)(width_height >> 16), eh = (float)(width_height & 0xffffu);
                                ImageDesc desc = img_table.rd(index);
                                const uint32_t img_h = desc.height & 0x7fffffffu;
                                const bool img_srgb = (desc.height >> 31) != 0u;  // render.go:137 Rgba8Srgb: decoded by the texture unit
                                auto texel = [&](int32_t tx, int32_t ty) {
                                    if (tx < 0 || ty < 0 || (uint32_t)tx >= desc.width || (uint32_t)ty >= img_h) return V4{0, 0, 0, 0};
                                    size_t ix = (size_t)desc.offset_px + (size_t)ty * desc.width + (size_t)tx;
                                    if (ix >= img_n) return V4{0, 0, 0, 0};
                                    const uint8_t* p = img_px + ix * 4;
                                    V4 c{(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f};
                                    if (img_srgb) { c.x = kSrgbToLinear[p[0]]; c.y = kSrgbToLinear[p[1]]; c.z = kSrgbToLinear[p[2]]; }
                                    return V4{c.x * c.w, c.y * c.w, c.z * c.w, c.w};  // premul_alpha
                                };
                                for (int i = 0; i < 4; i++) {
                                    float mx = xyx + (float)i, my = xyy;
                                    float u = m[0] * mx + m[2] * my + xl0;
                                    float v = m[1] * mx + m[3] * my + xl1;
                                    if (u < ew && v < eh && area[i] != 0.0f) {
                                        float fu = floor_(u), fv = floor_(v), cu = ceil_(u), cv = ceil_(v);
                                        float fru = fract_(u), frv = fract_(v);
                                        V4 a = texel(to_i32(fu), to_i32(fv));
                                        V4 bq = texel(to_i32(fu), to_i32(cv));
                                        V4 cq = texel(to_i32(cu), to_i32(fv));
                                        V4 dq = texel(to_i32(cu), to_i32(cv));
                                        auto mix4 = [&](V4 p, V4 q, float t) { return V4{mix_(p.x, q.x, t), mix_(p.y, q.y, t), mix_(p.z, q.z, t), mix_(p.w, q.w, t)}; };
                                        V4 fg = mix4(mix4(a, bq, frv), mix4(cq, dq, frv), fru);
                                        V4 fg_i{fg.x * area[i], fg.y * area[i], fg.z * area[i], fg.w * area[i]};
                                        float k = 1.0f - fg_i.w;
                                        rgba[i] = V4{rgba[i].x * k + fg_i.x, rgba[i].y * k + fg_i.y, rgba[i].z * k + fg_i.z, rgba[i].w * k + fg_i.w};
                                    }
                                }
                                cmd_ix += 2u;
                                break;
                            }
                            default: guard = (1u << 24); break;  // unknown tag: WGSL never advances
                        }
                    }
                    for (uint32_t i = 0; i < 4u; i++) {
                        uint32_t cx = gx * 4u + i, cy = gy;
                        if (cx < cfg.target_width && cy < cfg.target_height) {
                            V4 fg = rgba[i];
                            float a_inv = 1.0f / fmax_(fg.w, 1e-6f);
                            size_t px = (size_t)cy * cfg.target_width + cx;
                            if (px < output_px) {
                                uint16_t* o = output + px * 4;
                                o[0] = f32_to_f16_rtne(fg.x * a_inv);
                                o[1] = f32_to_f16_rtne(fg.y * a_inv);
                                o[2] = f32_to_f16_rtne(fg.z * a_inv);
                                o[3] = f32_to_f16_rtne(fg.w);
                            }
                        }
                    }
                }
        }
}

// ---------------------------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------------------------
extern "C" {

// Stage ids follow renderer.FullShaders field order (renderer/render.go:17-43).
enum {
    ST_PATHTAG_REDUCE = 0, ST_PATHTAG_REDUCE2, ST_PATHTAG_SCAN1, ST_PATHTAG_SCAN_SMALL, ST_PATHTAG_SCAN_LARGE,
    ST_BBOX_CLEAR, ST_FLATTEN, ST_DRAW_REDUCE, ST_DRAW_LEAF, ST_CLIP_REDUCE, ST_CLIP_LEAF, ST_BINNING,
    ST_TILE_ALLOC, ST_BACKDROP_DYN, ST_PATH_COUNT_SETUP, ST_PATH_COUNT, ST_COARSE, ST_PATH_TILING_SETUP,
    ST_PATH_TILING, ST_FINE_AREA, ST_FINE_MSAA8, ST_FINE_MSAA16
};

int oracle_dispatch(int stage, uint32_t gx, uint32_t gy, uint32_t gz, OBuf* b, int nb) {
    (void)gz;
    switch (stage) {
        case ST_PATHTAG_REDUCE: pathtag_reduce(gx, b); break;
        case ST_PATHTAG_REDUCE2: pathtag_reduce2(gx, b); break;
        case ST_PATHTAG_SCAN1: pathtag_scan1(gx, b); break;
        case ST_PATHTAG_SCAN_SMALL: pathtag_scan(gx, b, true); break;
        case ST_PATHTAG_SCAN_LARGE: pathtag_scan(gx, b, false); break;
        case ST_BBOX_CLEAR: bbox_clear(gx, b); break;
        case ST_FLATTEN: flatten(gx, b); break;
        case ST_DRAW_REDUCE: draw_reduce(gx, b); break;
        case ST_DRAW_LEAF: draw_leaf(gx, b); break;
        case ST_CLIP_REDUCE: clip_reduce(gx, b); break;
        case ST_CLIP_LEAF: clip_leaf(gx, b); break;
        case ST_BINNING: binning(gx, b); break;
        case ST_TILE_ALLOC: tile_alloc(gx, b); break;
        case ST_BACKDROP_DYN: backdrop_dyn(gx, b); break;
        case ST_PATH_COUNT_SETUP: path_count_setup(b); break;
        case ST_PATH_COUNT: path_count(gx, b); break;
        case ST_COARSE: coarse(gx, gy, b); break;
        case ST_PATH_TILING_SETUP: path_tiling_setup(b); break;
        case ST_PATH_TILING: path_tiling(gx, b); break;
        case ST_FINE_AREA: fine_area(gx, gy, b, nb); break;
        case ST_FINE_MSAA8: fine_area(gx, gy, b, nb, 8); break;
        case ST_FINE_MSAA16: fine_area(gx, gy, b, nb, 16); break;
        default: return -1;
    }
    return 0;
}

// scalar entry points for tests/test_oracle_math.py
float oracle_sin(float x) { return sin_(x); }
float oracle_cos(float x) { return cos_(x); }
float oracle_atan2(float y, float x) { return atan2_(y, x); }
float oracle_acos(float x) { return acos_(x); }
float oracle_asin(float x) { return asin_(x); }
float oracle_pow23(float x) { return pow23_abs_(x); }
float oracle_round(float x) { return round_(x); }
void oracle_vec_minmaxclamp(const float* a, const float* b, float* mn, float* mx, float* cl, int n) {
    for (int i = 0; i < n; i++) { mn[i] = fmin_(a[i], b[i]); mx[i] = fmax_(a[i], b[i]); cl[i] = clamp_(a[i], 0.0f, 1.0f); }
}
uint32_t oracle_to_u32(float x) { return to_u32(x); }
int32_t oracle_to_i32(float x) { return to_i32(x); }
uint16_t oracle_f32_to_f16(float x) { return f32_to_f16_rtne(x); }
float oracle_f16_to_f32(uint16_t x) { return f16_to_f32(x); }
uint32_t oracle_span(float a, float b) { return span(a, b); }
void oracle_reduce_tag(uint32_t w, uint32_t* out5) { TagMonoid m = reduce_tag(w); std::memcpy(out5, &m, 20); }
void oracle_map_draw_tag(uint32_t t, uint32_t* out4) { DrawMonoid m = map_draw_tag(t); std::memcpy(out4, &m, 16); }
void oracle_blend_mix_compose(const float* bg, const float* fg, uint32_t mode, float* out) {
    bl::V4 r = bl::blend_mix_compose(bl::V4{bg[0], bg[1], bg[2], bg[3]}, bl::V4{fg[0], fg[1], fg[2], fg[3]}, mode);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void oracle_vec_sin(const float* x, float* y, int n) { for (int i = 0; i < n; i++) y[i] = sin_(x[i]); }
void oracle_vec_cos(const float* x, float* y, int n) { for (int i = 0; i < n; i++) y[i] = cos_(x[i]); }
void oracle_vec_atan2(const float* a, const float* b, float* y, int n) { for (int i = 0; i < n; i++) y[i] = atan2_(a[i], b[i]); }
void oracle_vec_acos(const float* x, float* y, int n) { for (int i = 0; i < n; i++) y[i] = acos_(x[i]); }
void oracle_vec_asin(const float* x, float* y, int n) { for (int i = 0; i < n; i++) y[i] = asin_(x[i]); }
void oracle_vec_pow23(const float* x, float* y, int n) { for (int i = 0; i < n; i++) y[i] = pow23_abs_(x[i]); }

}  // extern "C"
