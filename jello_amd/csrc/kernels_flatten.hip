// kernels_flatten.hip -- K6 flatten (orig/flatten.wgsl:46-901): one thread per path-tag byte;
// Euler-spiral subdivision of fills, GPU stroke expansion (offset curves, caps, joins, arcs).
//
// MI355X design: the WGSL allocates every output line with atomicAdd(bump.lines), which makes the
// LineSoup order run-dependent.  Here the stage is classify -> items -> scan -> lines -> bbox:
//   k_flatten_classify   one thread per tag byte: splits it into up to 3 work items (see below) and appends them
//                        to a heavy / light list; writes PathBbox.draw_flags/trans_ix;
//   k_flatten_items      a wave per batch of 64 items: caps / joins / lines leave one record per line; the Euler jobs of
//                        the batch are subdivided together -- the nodes of their subdivision trees sit on one LDS stack and
//                        64 are tested per step, whichever jobs they belong to -- and leave one 64-byte record per accepted
//                        piece, densely, in the order the pieces were found.  Every LINE gets a temporary slot; the slots
//                        of a batch are ONE range, laid out in the canonical order of its lines (job, piece, line), and
//                        slot_info[slot] = (record, index of the line in its piece); counts[item] = lines of the item;
//   jh_scan_u32          line base per item; the total lands in bump.lines;
//   k_flatten_lines      one thread per temporary slot: evaluates the line's end point from the piece record and writes the
//                        line to lines[bases[item] + k] (its start is the end point of the line before it);
//   k_flatten_bbox       streams the finished lines and folds their boxes by path (segmented wave scan), honouring the
//                        WGSL's per-tag extent rule -> path bounding boxes.
// Result: lines are ordered by (tag byte, emission order) -- the reference's own sequential order
// (shaders/cpu/flatten.go:664-823) -- with the subdivision arithmetic executed exactly once.
// Temporary memory: 8 bytes per slot + 64 bytes per record, both allocated EXACTLY (one returning atomic per batch and kind
// on a packed 64-bit cursor: slots in the low half, records in the high half), capacity = the line buffer's: a frame whose
// lines fit the line buffer fits the temporary.
// Algorithmic traffic: scene bytes + 20 B / tag word in, 24 B / line out (+ 64 B / piece and 8 B / line through the temp).
// k_flatten_items is VALU/latency-bound (f64 transcendentals, 1...30 subdivision attempts per job).
#include <hip/hip_runtime.h>
__shared__ double fl_atan_tab[9];  // dmath.h: atan(k/8) for the table-split arctangent, filled by atan_tab_fill()
#define JD_ATAN_TAB_LDS fl_atan_tab
#include "kcommon.h"
#define FF_INLINE __device__ __forceinline__
#include "flatten_fast.h"

using namespace jk;
using namespace jd;

namespace {

struct CubicParams { float th0, th1, chord_len, err; };
struct EulerParams { float th0, th1, k0, k1, ch; };
struct CubicPoints { V2 p0, p1, p2, p3; };
struct PointDeriv { V2 point, deriv; };
struct PathTagData { uint32_t tag_byte; MonoidK<5> monoid; };

#define DERIV_THRESH 1e-6f
#define DERIV_THRESH_SQUARED (DERIV_THRESH * DERIV_THRESH)
#define DERIV_EPS 1e-6f
#define SUBDIV_LIMIT (1.0f / 65536.0f)
#define K1_THRESH 1e-3f
#define DIST_THRESH 1e-3f
#define TANGENT_THRESH 1e-6f

#define FL_INVALID 0xffffffffu
// slot_info[t] = (record index, FL_INFO_* | lines of the piece << 8 | index of this line in the piece); every slot below the
// slot cursor is written exactly once per frame (allocation is exact), so the array needs no clearing
#define FL_INFO_PIECE 0x80000000u    // line i of an Euler piece (record: see "piece record" below)
#define FL_INFO_DIRECT 0x40000000u   // complete line (record: {item, k, path_ix, -}, {p0, p1})
#ifndef FL_REFILL_LANES
#define FL_REFILL_LANES 32u  // idle lanes that trigger a refill of the wave
#endif

// The temporary: slot_info (8 B per slot) and records (64 B), each cut into K regions of R entries with a cursor of its own.
// ONE cursor for everything is a hot word: it sustains ~60-90 returning atomics per microsecond, and the 20 000 allocations of
// a C3 frame took 210 us longer than the arithmetic; K cursors in K memory channels scale.  An allocation takes n consecutive
// entries of the caller's current region (one returning atomic); a region that cannot hold it is left for the next one, for
// good (the caller's region index is sticky), and the slots it leaves unused at the region's end are marked empty.  Capacity:
// no allocation is larger than FL_MAX_GRAB, a region wastes less than one allocation at its end, so K * R >= lines + K *
// FL_MAX_GRAB holds every frame whose lines fit the line buffer.
#define FL_MAX_REGIONS 8u
#define FL_MAX_GRAB 51200u  // = the lines of one job of the cooperative subdivision (512 pieces of 100); an arc has < 31 416
struct FlTemp {
    uint2* sinfo;
    uint4* recs;
    uint32_t* ctr;  // the stage's counters (FL_CTR_*)
    uint32_t K, R;
};
#define FL_CTR_CURSOR 512u    // word index of region 0's slot cursor; its record cursor 32 words on; next region FL_CUR_STRIDE on
#define FL_CUR_STRIDE 64u
// n consecutive slots (KIND 0) or records (KIND 1): the position, or FL_INVALID when every region is full (the frame has
// overflowed its line buffer).  `home` is the wave's region: WAVE-UNIFORM, so that the lanes that allocate in one instruction
// can share one atomic (wave_bump; one atomic per lane means 400 000 per C3 frame instead of 8 000: +370 us).  `failed` is raised when home could not serve: the wave moves
// on at its next uniform point (fl_next_home).
// (wave_bump: kcommon.h)
template <int KIND>
JD uint32_t fl_grab(const FlTemp& T, uint32_t home, uint32_t n, bool& failed) {
    // (the attempt on home stands apart from the loop over the other regions: inside the loop the region is a per-lane value)
    uint32_t p = wave_bump(T.ctr + FL_CTR_CURSOR + FL_CUR_STRIDE * home + (KIND ? 32u : 0u), n);
    if (__builtin_expect(p <= T.R && n <= T.R - p, 1)) return home * T.R + p;
    failed = true;
    uint32_t region = home;
    for (uint32_t tries = 1u;; tries++) {
        if (KIND == 0 && p < T.R)  // the region's last slots stay empty
            for (uint32_t i = p; i < T.R; i++) T.sinfo[(size_t)region * T.R + i] = make_uint2(0u, 0u);
        // (an add that found the region full already is taken back: the cursor of a full region stays within one allocation of R
        // however many waves still try it before they move on -- it can never wrap and reopen the region)
        if (p >= T.R) atomicSub(T.ctr + FL_CTR_CURSOR + FL_CUR_STRIDE * region + (KIND ? 32u : 0u), n);
        if (tries >= T.K) return FL_INVALID;
        region = region + 1u == T.K ? 0u : region + 1u;
        p = atomicAdd(T.ctr + FL_CTR_CURSOR + FL_CUR_STRIDE * region + (KIND ? 32u : 0u), n);
        if (p <= T.R && n <= T.R - p) return region * T.R + p;
    }
}
// (wave-uniform) the next region that still has room, seen from `home`; home itself if none has
JD uint32_t fl_next_home(const FlTemp& T, uint32_t home, int kind) {
    uint32_t region = home;
    for (uint32_t tries = 0u; tries < T.K; tries++) {
        const uint32_t cur = __hip_atomic_load(T.ctr + FL_CTR_CURSOR + FL_CUR_STRIDE * region + (kind ? 32u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur < T.R) return region;
        region = region + 1u == T.K ? 0u : region + 1u;
    }
    return home;
}

// Line sink of the DIRECT items (caps, joins, straight segments) and of the sequential fall-back walk.  EMIT: an allocation
// of n lines takes n temporary slots and n records (two returning atomics on different words; lanes that allocate in the same
// instruction are combined by the compiler's atomic optimizer), a line leaves its record and its slot_info; k_flatten_lines
// then moves it to lines[bases[item] + k], the canonical position.  !EMIT: only counts.
template <bool EMIT>
struct Out {
    const JlConfig* cfg;
    FlTemp T;
    uint32_t home_s, home_r;   // the wave's current slot / record region (uniform)
    bool failed_s, failed_r;   // an allocation of this lane found the region full
    uint32_t slot;
    uint32_t cursor;           // lines emitted so far by this item (local index of the next line)
    uint32_t a_first, a_spos, a_rpos;  // current allocation: first local index, its slot and its record

    JD uint32_t alloc(uint32_t n) {
        uint32_t first = cursor;
        cursor += n;
        if (EMIT) {
            a_first = first;
#if defined(FL_ISPLIT) && FL_ISPLIT == 2  // (measurement builds only: direct lines allocate nothing -- results are wrong)
            a_spos = a_rpos = FL_INVALID;
#else
            a_spos = fl_grab<0>(T, home_s, n, failed_s);
            a_rpos = fl_grab<1>(T, home_r, n, failed_r);
#endif
        }
        return first;
    }
    JD void write_line(uint32_t line_ix, uint32_t path_ix, V2 p0, V2 p1) {  // flatten.wgsl:749-756
        if (EMIT) {
            if (a_spos != FL_INVALID && a_rpos != FL_INVALID) {
                const uint32_t t = a_spos + (line_ix - a_first), r = a_rpos + (line_ix - a_first);
                T.recs[(size_t)r * 4u] = make_uint4(slot, line_ix, path_ix, 0u);
                T.recs[(size_t)r * 4u + 1u] = make_uint4(f2u(p0.x), f2u(p0.y), f2u(p1.x), f2u(p1.y));
                T.sinfo[t] = make_uint2(r, FL_INFO_DIRECT);
            } else if (a_spos != FL_INVALID) {
                T.sinfo[a_spos + (line_ix - a_first)] = make_uint2(0u, 0u);  // (a slot without a record: empty)
            }
        }
    }
    JD void write_line_t(uint32_t line_ix, uint32_t path_ix, V2 p0, V2 p1, const Xf& t) {
        if (EMIT) write_line(line_ix, path_ix, xf_apply(t, p0), xf_apply(t, p1));
    }
    JD void output_line_t(uint32_t path_ix, V2 p0, V2 p1, const Xf& t) {
        uint32_t ix = alloc(1);
        write_line_t(ix, path_ix, p0, p1, t);
    }
};

JD CubicParams cubic_from_points_derivs(V2 p0, V2 p1, V2 q0, V2 q1, float dt) {  // flatten.wgsl:94-133
    V2 chord = p1 - p0;
    float chord_squared = dot(chord, chord);
    float chord_len = sqrt_(chord_squared);
    CubicParams r;
    if (chord_squared < DERIV_THRESH_SQUARED) {
        float chord_err = sqrt_((float)(9.0 / 32.0) * (dot(q0, q0) + dot(q1, q1))) * dt;
        r.th0 = 0.0f; r.th1 = 0.0f; r.chord_len = DERIV_THRESH; r.err = chord_err;
        return r;
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
    r.th0 = th0; r.th1 = th1; r.chord_len = chord_len; r.err = err;
    return r;
}

JD EulerParams es_params_from_angles(float th0, float th1) {  // flatten.wgsl:135-158
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
    EulerParams r;
    r.th0 = th0; r.th1 = th1; r.k0 = k0; r.k1 = k1; r.ch = ch;
    return r;
}
JD float es_params_eval_th(const EulerParams& p, float t) { return (p.k0 + 0.5f * p.k1 * (t - 1.0f)) * t - p.th0; }

JD V2 integ_euler_10(float k0, float k1) {  // flatten.wgsl:165-195
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
JD V2 es_params_eval(const EulerParams& p, float t) {  // :197-209
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
JD V2 es_params_eval_with_offset(const EulerParams& p, float t, float offset) {  // :211-215
    float th = es_params_eval_th(p, t);
    V2 v = offset * v2(sin_(th), cos_(th));
    return es_params_eval(p, t) + v;
}
JD V2 es_seg_eval_with_offset(V2 es_p0, V2 es_p1, const EulerParams& p, float t, float normalized_offset) {  // :222-226
    V2 chord = es_p1 - es_p0;
    V2 xy = es_params_eval_with_offset(p, t, normalized_offset);
    return es_p0 + v2(chord.x * xy.x - chord.y * xy.y, chord.x * xy.y + chord.y * xy.x);
}
JD float pow_1_5_signed(float x) { return x * sqrt_(abs_(x)); }

#define BREAK1 0.8f
#define BREAK2 1.25f
#define BREAK3 2.1f
#define SIN_SCALE 1.0976991822760038f
#define QUAD_A1 0.6406f
#define QUAD_B1 -0.81f
#define QUAD_C1 0.9148117935952064f
#define QUAD_A2 0.5f
#define QUAD_B2 -0.156f
#define QUAD_C2 0.16145779359520596f
#define FRAC_PI_4 0.7853981633974483f
#define CBRT_9_8 1.040041911525952f

JD float espc_int_approx(float x) {  // :250-262
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
JD float espc_int_inv_approx(float x) {  // :264-278
    // const QUAD_W/V/U are f32 const-expressions in the WGSL (flatten.wgsl:241-246)
    const float QUAD_W1 = 0.5f * QUAD_B1 / QUAD_A1;
    const float QUAD_V1 = 1.0f / QUAD_A1;
    const float QUAD_U1 = QUAD_W1 * QUAD_W1 - QUAD_C1 / QUAD_A1;
    const float QUAD_W2 = 0.5f * QUAD_B2 / QUAD_A2;
    const float QUAD_V2 = 1.0f / QUAD_A2;
    const float QUAD_U2 = QUAD_W2 * QUAD_W2 - QUAD_C2 / QUAD_A2;
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
JD PointDeriv eval_cubic_and_deriv(V2 p0, V2 p1, V2 p2, V2 p3, float t) {  // :285-293
    float m = 1.0f - t;
    float mm = m * m;
    float mt = m * t;
    float tt = t * t;
    PointDeriv r;
    r.point = p0 * (mm * m) + (p1 * (3.0f * mm) + p2 * (3.0f * mt) + p3 * tt) * t;
    r.deriv = (p1 - p0) * mm + (p2 - p1) * (2.0f * mt) + (p3 - p2) * tt;
    return r;
}
JD V2 cubic_start_tangent(V2 p0, V2 p1, V2 p2, V2 p3) {  // :295-301
    const float EPS = 1e-12f;
    V2 d01 = p1 - p0, d02 = p2 - p0, d03 = p3 - p0;
    V2 inner = (dot(d02, d02) > EPS) ? d02 : d03;
    return (dot(d01, d01) > EPS) ? d01 : inner;
}
JD V2 cubic_end_tangent(V2 p0, V2 p1, V2 p2, V2 p3) {  // :303-309
    const float EPS = 1e-12f;
    V2 d23 = p3 - p2, d13 = p3 - p1, d03 = p3 - p0;
    V2 inner = (dot(d13, d13) > EPS) ? d13 : d03;
    return (dot(d23, d23) > EPS) ? d23 : inner;
}

struct Scene {
    const JlConfig* cfg;
    Buf<uint32_t> scene;
    Buf<JlTagMonoid> tag_monoids;
};

// ------------------------------------------------------------------------------------------------
// flatten_euler (flatten.wgsl:328-477) in two kernels.
//
// k_flatten_items decides the subdivision and, for every ACCEPTED piece, the number of its lines; the piece leaves a 64-byte
// record, its lines are NOT evaluated in this kernel (that loop at 3 waves/SIMD and 35 % lane use cost 240 of the stage's
// 600 us).  k_flatten_lines (after the per-item line counts are scanned): one thread per temporary slot evaluates the END point
// of its line from the piece record (the WGSL's arithmetic per point), transforms it and writes it straight into the
// canonical LineSoup position of the line AND as the START point of the following line of the item -- a line's start
// is the end of the line before it (flatten.wgsl:462-468); only the item's first line takes its start from the
// record.  Every point is computed once, with the same operations on the same values as in the sequential
// formulation, hence the same bits.  Lines emitted directly (caps, joins) wait in records of their own and are copied.
// ------------------------------------------------------------------------------------------------
JD void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// piece record: one 64-byte sector, 4 x uint4 at recs[4 * r]:
//   0: es_p0.x es_p0.y es_p1.x es_p1.y   1: th0 th1 int0 integral   2: noff item path_ix trans_ix<<6|flags
//   3: (first piece of its item ? t_start.x : index of the piece's first line in the item) t_start.y t_end.x t_end.y
// t_start is read only by the item's first line, t_end only by its last.  The number of lines n comes with slot_info.
// flags: bits 0-1 robust case, 4 = ends at t == 1 (last line ends in t_end), 8 = offset >= 0, 16 = offset == 0,
//        32 = first piece of its item.  Of the subdivision constants of flatten.wgsl:404-433, int0 and integral (two
//        espc_int_approx evaluations) travel in the record; the Euler parameters k0, k1, ch are recomputed from the two angles
//        (es_params_from_angles: 45 binary32 operations, no transcendentals), a and b by the same two products.
JD void piece_record_write(uint4* __restrict__ rec, V2 es_p0, V2 es_p1, float th0, float th1, float int0, float integral, float noff,
                           uint32_t item, uint32_t path_ix, uint32_t trans_ix, uint32_t fl, uint32_t first, V2 t_start, V2 t_end) {
    rec[0] = make_uint4(f2u(es_p0.x), f2u(es_p0.y), f2u(es_p1.x), f2u(es_p1.y));
    rec[1] = make_uint4(f2u(th0), f2u(th1), f2u(int0), f2u(integral));
    rec[2] = make_uint4(f2u(noff), item, path_ix, (trans_ix << 6) | fl);
    rec[3] = make_uint4((fl & 32u) != 0u ? f2u(t_start.x) : first, f2u(t_start.y), f2u(t_end.x), f2u(t_end.y));
}
// slot_info of the n lines of a piece whose first line has slot tpos
JD void piece_slots_write(uint2* __restrict__ sinfo, uint32_t tpos, uint32_t n_u, uint32_t r) {
#if !(defined(FL_ISPLIT) && FL_ISPLIT == 1)  // (measurement builds only: no slot_info of pieces -- results are wrong)
    for (uint32_t i = 0u; i < n_u; i++) sinfo[tpos + i] = make_uint2(r, FL_INFO_PIECE | (n_u << 8) | i);
#endif
}

struct EulerJob {
    bool valid;
    CubicPoints cubic;
    uint32_t path_ix, trans_ix;
    Xf local_to_device;
    float offset;
    V2 start_p, end_p;
};

// Resumable per-lane state of flatten_euler: lanes that finish their job early are REFILLED with the next
// work item while the others keep subdividing (the attempts per job vary from 1 to ~30, and with ~1.5 cubic
// jobs per resident lane a wave-synchronous "64 items, wait for the slowest" loop idles most lanes).
struct EulerLane {
    V2 p0, p1, p2, p3;
    float scale, offset;
    V2 t_end;
    uint32_t t0_u;
    float dt;
    V2 last_p, last_q;
    float last_t;
    V2 t_start;          // start point of the item's first line
    bool first_piece;
    uint32_t path_ix, trans_ix;
    bool done;
};

JD void euler_begin(EulerLane& e, const EulerJob& job) {  // flatten.wgsl:328-360
    e.p0 = e.p1 = e.p2 = e.p3 = v2(0, 0);
    e.scale = 1.0f;
    V2 t_start = job.start_p;
    e.t_end = job.end_p;
    e.done = !job.valid;
    e.offset = job.offset;
    e.path_ix = job.path_ix;
    e.trans_ix = job.trans_ix;
    if (job.valid) {
        if (job.offset == 0.0f) {
            e.p0 = xf_apply(job.local_to_device, job.cubic.p0);
            e.p1 = xf_apply(job.local_to_device, job.cubic.p1);
            e.p2 = xf_apply(job.local_to_device, job.cubic.p2);
            e.p3 = xf_apply(job.local_to_device, job.cubic.p3);
            e.scale = 1.0f;
            t_start = e.p0;
            e.t_end = e.p3;
        } else {
            e.p0 = job.cubic.p0; e.p1 = job.cubic.p1; e.p2 = job.cubic.p2; e.p3 = job.cubic.p3;
            const Xf& tr = job.local_to_device;
            e.scale = 0.5f * length(v2(tr.m0 + tr.m3, tr.m1 - tr.m2)) + length(v2(tr.m0 - tr.m3, tr.m1 + tr.m2));
        }
        if (veq(e.p0, e.p1) && veq(e.p0, e.p2) && veq(e.p0, e.p3)) e.done = true;
    }
    e.t0_u = 0u;
    e.dt = 1.0f;
    e.last_p = e.p0;
    e.last_q = e.p1 - e.p0;
    if (!e.done && dot(e.last_q, e.last_q) < DERIV_THRESH_SQUARED) e.last_q = eval_cubic_and_deriv(e.p0, e.p1, e.p2, e.p3, DERIV_EPS).deriv;
    e.last_t = 0.0f;
    e.t_start = t_start;
    e.first_piece = true;
}

// `refill()` is called (by the whole wave) when enough lanes are idle; it finalises finished items, gives idle
// lanes new ones (euler_begin) and returns false once no lane is active and the queue is empty.
template <class Refill>
JD void flatten_euler_wave(Out<true>& o, EulerLane& e, Refill&& refill) {
    V2 &p0 = e.p0, &p1 = e.p1, &p2 = e.p2, &p3 = e.p3;
    float& scale = e.scale;
    const float& offset = e.offset;
    V2& t_end = e.t_end;
    bool& done = e.done;
    const float tol = 0.25f;
    uint32_t& t0_u = e.t0_u;
    float& dt = e.dt;
    V2 &last_p = e.last_p, &last_q = e.last_q;
    float& last_t = e.last_t;
    for (;;) {
        {
            uint64_t idle = __builtin_amdgcn_ballot_w64(done);
            if ((uint32_t)__builtin_popcountll(idle) >= FL_REFILL_LANES)
                if (!refill()) break;
        }

        bool accept = false;
        uint32_t n_u = 0u;
        float pc_noff = 0.0f, pc_int0 = 0.0f, pc_integral = 0.0f;
        EulerParams ep;
        ep.th0 = ep.th1 = ep.k0 = ep.k1 = ep.ch = 0.0f;
        V2 es_p0 = v2(0, 0), es_p1 = v2(0, 0);
        uint32_t pc_flags = 0u;
        if (!done) {
            float t0 = (float)t0_u * dt;
            if (t0 == 1.0f) {
                done = true;
            } else {
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
                if (cp.err * scale <= tol || dt <= SUBDIV_LIMIT) {
                    ep = es_params_from_angles(cp.th0, cp.th1);
                    float k0 = ep.k0 - 0.5f * ep.k1;
                    float k1 = ep.k1;
                    float normalized_offset = offset / cp.chord_len;
                    float dist_scaled = normalized_offset * ep.ch;
                    float scale_multiplier = sqrt_(0.125f * scale * cp.chord_len / (ep.ch * tol));
                    float a = 0.0f, b = 0.0f, integral = 0.0f, int0 = 0.0f, n_frac;
                    uint32_t robust = 0u;
                    if (abs_(k1) < K1_THRESH) {
                        float k = ep.k0;
                        n_frac = sqrt_(abs_(k * (k * dist_scaled + 1.0f)));
                        robust = 1u;
                    } else if (abs_(dist_scaled) < DIST_THRESH) {
                        a = k1;
                        b = k0;
                        int0 = pow_1_5_signed(b);
                        float int1 = pow_1_5_signed(a + b);
                        integral = int1 - int0;
                        n_frac = (float)(2.0 / 3.0) * integral / a;
                        robust = 2u;
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
                    n_u = to_u32(n);
                    accept = true;
                    pc_noff = normalized_offset; pc_int0 = int0; pc_integral = integral;  // (a, b: two products, recomputed)
                    es_p0 = this_p0; es_p1 = this_pq1.point;
                    pc_flags = robust | ((t1 == 1.0f) ? 4u : 0u) | ((offset >= 0.0f) ? 8u : 0u) | ((offset == 0.0f) ? 16u : 0u);
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
        }
        if (accept) {  // (n slots and n records per piece, of which the piece uses the first: this walk is the rare fall-back)
            const uint32_t first = o.alloc(n_u);
            const uint32_t tpos = o.a_spos, r = o.a_rpos;
            if (r != FL_INVALID) {
                const uint32_t fl = pc_flags | (e.first_piece ? 32u : 0u);
                piece_record_write(o.T.recs + (size_t)r * 4u, es_p0, es_p1, ep.th0, ep.th1, pc_int0, pc_integral, pc_noff, o.slot, e.path_ix,
                                   e.trans_ix, fl, first, e.t_start, t_end);
            }
            if (tpos != FL_INVALID) {
                if (r != FL_INVALID) piece_slots_write(o.T.sinfo, tpos, n_u, r);
                else for (uint32_t i = 0u; i < n_u; i++) o.T.sinfo[tpos + i] = make_uint2(0u, 0u);
            }
            e.first_piece = false;
        }
    }
}

// flatten.wgsl:490-517
template <bool EMIT>
JD void flatten_arc(Out<EMIT>& o, uint32_t path_ix, V2 begin, V2 end, V2 center, float angle, const Xf& transform) {
    V2 p0 = xf_apply(transform, begin);
    V2 r = begin - center;
    const float MIN_THETA = 0.0001f;
    const float tol = 0.25f;
    float radius = fmax_(tol, length(p0 - xf_apply(transform, center)));
    float theta = fmax_(MIN_THETA, 2.0f * acos_(1.0f - tol / radius));
    uint32_t n_lines = umax_(1u, to_u32(ceil_(angle / theta)));
    uint32_t line_ix = o.alloc(n_lines);
    if (EMIT) {
        float cs = cos_(theta);
        float sn = sin_(theta);
        for (uint32_t i = 0; i < n_lines - 1u; i++) {
            r = v2(cs * r.x + sn * r.y, -sn * r.x + cs * r.y);
            V2 p1 = xf_apply(transform, center + r);
            o.write_line(line_ix + i, path_ix, p0, p1);
            p0 = p1;
        }
        V2 p1 = xf_apply(transform, end);
        o.write_line(line_ix + n_lines - 1u, path_ix, p0, p1);
    }
}
// flatten.wgsl:519-543
template <bool EMIT>
JD void draw_cap(Out<EMIT>& o, uint32_t path_ix, uint32_t cap_style, V2 point, V2 cap0, V2 cap1, V2 offset_tangent, const Xf& transform) {
    if (cap_style == JL_STYLE_FLAGS_CAP_ROUND) {
        flatten_arc<EMIT>(o, path_ix, cap0, cap1, point, 3.1415927f, transform);
        return;
    }
    V2 start = cap0, end = cap1;
    bool is_square = (cap_style == JL_STYLE_FLAGS_CAP_SQUARE);
    uint32_t line_ix = o.alloc(is_square ? 3u : 1u);
    if (is_square) {
        V2 v = offset_tangent;
        V2 p0 = start + v;
        V2 p1 = end + v;
        o.write_line_t(line_ix + 1u, path_ix, start, p0, transform);
        o.write_line_t(line_ix + 2u, path_ix, p1, end, transform);
        start = p0;
        end = p1;
    }
    o.write_line_t(line_ix, path_ix, start, end, transform);
}
// flatten.wgsl:545-614
template <bool EMIT>
JD void draw_join(Out<EMIT>& o, uint32_t path_ix, uint32_t style_flags, V2 p0, V2 tan_prev, V2 tan_next, V2 n_prev, V2 n_next,
                  const Xf& transform) {
    V2 front0 = p0 + n_prev;
    V2 front1 = p0 + n_next;
    V2 back0 = p0 - n_next;
    V2 back1 = p0 - n_prev;
    float cr = tan_prev.x * tan_next.y - tan_prev.y * tan_next.x;
    float d = dot(tan_prev, tan_next);
    uint32_t join = style_flags & JL_STYLE_FLAGS_JOIN_MASK;
    if (join == JL_STYLE_FLAGS_JOIN_BEVEL) {
        uint32_t line_ix = o.alloc(2u);
        o.write_line_t(line_ix, path_ix, front0, front1, transform);
        o.write_line_t(line_ix + 1u, path_ix, back0, back1, transform);
    } else if (join == JL_STYLE_FLAGS_JOIN_MITER) {
        float hypot = length(v2(cr, d));
        float miter_limit = f16_to_f32((uint16_t)(style_flags & JL_STYLE_MITER_LIMIT_MASK));
        uint32_t line_ix;
        if (2.0f * hypot < (hypot + d) * miter_limit * miter_limit && cr != 0.0f) {
            bool is_backside = cr > 0.0f;
            V2 fp_last = is_backside ? back1 : front0;
            V2 fp_this = is_backside ? back0 : front1;
            V2 p = is_backside ? back0 : front0;
            V2 v = fp_this - fp_last;
            float h = (tan_prev.x * v.y - tan_prev.y * v.x) / cr;
            V2 miter_pt = fp_this - tan_next * h;
            line_ix = o.alloc(3u);
            o.write_line_t(line_ix, path_ix, p, miter_pt, transform);
            line_ix += 1u;
            if (is_backside) back0 = miter_pt; else front0 = miter_pt;
        } else {
            line_ix = o.alloc(2u);
        }
        o.write_line_t(line_ix, path_ix, front0, front1, transform);
        o.write_line_t(line_ix + 1u, path_ix, back0, back1, transform);
    } else if (join == JL_STYLE_FLAGS_JOIN_ROUND) {
        V2 arc0, arc1, other0, other1;
        if (cr > 0.0f) { arc0 = back0; arc1 = back1; other0 = front0; other1 = front1; }
        else { arc0 = front0; arc1 = front1; other0 = back0; other1 = back1; }
        flatten_arc<EMIT>(o, path_ix, arc0, arc1, p0, abs_(atan2_(cr, d)), transform);
        o.output_line_t(path_ix, other0, other1, transform);
    }
}


JD V2 read_f32_point(const Scene& s, uint32_t ix) {
    uint32_t b = s.cfg->layout.pathdata_base + ix;
    return v2(u2f(s.scene.rd(b)), u2f(s.scene.rd(b + 1u)));
}
JD V2 read_i16_point(const Scene& s, uint32_t ix) {
    uint32_t raw = s.scene.rd(s.cfg->layout.pathdata_base + ix);
    float x = (float)((int32_t)(raw << 16) >> 16);
    float y = (float)((int32_t)raw >> 16);
    return v2(x, y);
}
JD PathTagData compute_tag_monoid(const Scene& s, uint32_t ix) {  // flatten.wgsl:668-682
    uint32_t tag_word = s.scene.rd(s.cfg->layout.pathtag_base + (ix >> 2));
    uint32_t shift = (ix & 3u) * 8u;
    MonoidK<5> tm = reduce_tag(tag_word & ((1u << shift) - 1u));
    JlTagMonoid pm = s.tag_monoids.rd(ix >> 2);
    tm.v[0] += pm.trans_ix; tm.v[1] += pm.pathseg_ix; tm.v[2] += pm.pathseg_offset; tm.v[3] += pm.style_ix; tm.v[4] += pm.path_ix;
    PathTagData r;
    r.tag_byte = (tag_word >> shift) & 0xffu;
    tm.v[0] -= 1u;
    tm.v[3] -= 2u;
    r.monoid = tm;
    return r;
}
JD CubicPoints read_path_segment(const Scene& s, const PathTagData& tag, bool is_stroke) {  // flatten.wgsl:691-747
    V2 p0 = v2(0, 0), p1 = v2(0, 0), p2 = v2(0, 0), p3 = v2(0, 0);
    uint32_t seg_type = tag.tag_byte & 3u;
    uint32_t pathseg_offset = tag.monoid.v[2];
    bool is_stroke_cap_marker = is_stroke && (tag.tag_byte & JL_PATH_TAG_SUBPATH_END) != 0u;
    bool is_open = seg_type == JL_PATH_TAG_QUADTO;
    if ((tag.tag_byte & JL_PATH_TAG_F32) != 0u) {
        p0 = read_f32_point(s, pathseg_offset);
        p1 = read_f32_point(s, pathseg_offset + 2u);
        if (seg_type >= JL_PATH_TAG_QUADTO) {
            p2 = read_f32_point(s, pathseg_offset + 4u);
            if (seg_type == JL_PATH_TAG_CUBICTO) p3 = read_f32_point(s, pathseg_offset + 6u);
        }
    } else {
        p0 = read_i16_point(s, pathseg_offset);
        p1 = read_i16_point(s, pathseg_offset + 1u);
        if (seg_type >= JL_PATH_TAG_QUADTO) {
            p2 = read_i16_point(s, pathseg_offset + 2u);
            if (seg_type == JL_PATH_TAG_CUBICTO) p3 = read_i16_point(s, pathseg_offset + 3u);
        }
    }
    if (is_stroke_cap_marker && is_open) {
        p0 = p1;
        p1 = p2;
        seg_type = JL_PATH_TAG_LINETO;
    }
    const float THIRD = (float)(1.0 / 3.0);
    if (seg_type == JL_PATH_TAG_LINETO) {
        p3 = p1;
        p2 = vmix(p3, p0, THIRD);
        p1 = vmix(p0, p3, THIRD);
    } else if (seg_type == JL_PATH_TAG_QUADTO) {
        p3 = p2;
        p2 = vmix(p1, p2, THIRD);
        p1 = vmix(p1, p0, THIRD);
    }
    CubicPoints r;
    r.p0 = p0; r.p1 = p1; r.p2 = p2; r.p3 = p3;
    return r;
}

// ------------------------------------------------------------------------------------------------
// Work decomposition.  The WGSL runs one invocation per tag byte (flatten.wgsl:809-901); in a wave of
// 64 consecutive tag bytes only a few lanes hold a curve and a stroked cubic costs several times a
// filled one, so SIMD utilisation is poor.  Here every tag byte is split into up to three independent
// ITEMS -- slot = 3*tag_ix + sub:
//     fill segment            sub 0: flatten_euler(offset 0)
//     stroke segment          sub 0: flatten_euler(+offset)   sub 1: flatten_euler(-offset)   sub 2: join or end cap
//     open-stroke cap marker  sub 0: start cap
// Slots are in the reference's emission order, so "count per slot -> exclusive scan -> emit at base" still
// yields the canonical LineSoup order.  Items are appended (wave-aggregated atomics; list order is
// irrelevant) to a HEAVY list (quad/cubic Euler flattening) or a LIGHT list (lines, caps, joins), and the
// count/emit kernels walk heavy-then-light so that waves are homogeneous.
// ------------------------------------------------------------------------------------------------
struct Seg {
    PathTagData tag;
    uint32_t style_flags;
    bool is_stroke;
};

JD Seg load_seg(const Scene& s, uint32_t ix) {
    Seg r;
    r.tag = compute_tag_monoid(s, ix);
    r.style_flags = s.scene.rd(s.cfg->layout.style_base + r.tag.monoid.v[3]);
    r.is_stroke = (r.style_flags & JL_STYLE_FLAGS_STYLE) != 0u;
    return r;
}

// One atomicAdd per workgroup and class: per-lane item counts are prefix-summed across the block, thread 0
// reserves the block's range, every lane writes its own slots.  (A per-lane atomicAdd on two hot
// words costs ~3.7 ms for 1.2 M items on MI355X; this costs ~10 us.)  List order is irrelevant for the
// result but this keeps it nearly sorted by tag, i.e. coalesced scene reads and line writes later on.
// counters: [0] heavy items, [FL_CTR_LIGHT] light items -- the two list counters are hot (every workgroup of
// k_flatten_classify waits for its two returns) and live in different memory channels --, from FL_CTR_CURSOR the cursors of
// the temporary's regions (FlTemp)
#define FL_CTR_LIGHT 256u
#define FL_CTR_WORDS (FL_CTR_CURSOR + FL_CUR_STRIDE * FL_MAX_REGIONS)
// The classification: one thread per tag WORD (four tag bytes: the word and its monoid are read once, the bytes' monoids follow from
// them; up to round 5 a thread took eight tag bytes and fetched word and monoid for each).  SCAN != 0: as an epilogue of the LAST
// pathtag scan (pathtag_scan.wgsl:24-64 / pathtag_scan_large; the engine holds that dispatch back when flatten follows it,
// jello_hip.cpp Deferred) -- the thread then HAS the word and its prefix monoid in registers instead of reading them back: one
// launch, 20 us, where pathtag_scan took 4 and the classification 23.  What the scan writes (tag_monoids) and what the
// classification writes (lists, counters, counts, draw_flags / trans_ix of the path boxes) are the same words either way.
#ifndef PSC_BLOCKS
#define PSC_BLOCKS 2u  // blocks of 256 tag words per workgroup: the two hot list counters see one atomic pair per workgroup (782 -> 391 on C3);
                       // C3, with / without the scan in it: 24.5 / 24.0 us with 1, 20.3 / 19.0 with 2, 23.5 / 21.1 with 3, 25.1 / 22.2 with 4
#endif
template <int SCAN>  // 0: tag_monoids are there; 1: + pathtag_scan_small; 2: + pathtag_scan_large
__global__ __launch_bounds__(JL_WG) void k_flatten_classify(const JlConfig* __restrict__ cfg, Buf<uint32_t> scene, Buf<JlTagMonoid> reduced,
                                                                 Buf<JlTagMonoid> tag_monoids, uint32_t n_blocks, Buf<JlPathBbox> path_bboxes,
                                                                 uint32_t* __restrict__ list, uint32_t* __restrict__ counters, uint32_t cap,
                                                                 uint32_t n_tags, uint32_t* __restrict__ counts, uint32_t absorb,
                                                                 uint32_t* __restrict__ bump_words) {
    __shared__ uint32_t sh[20];
    __shared__ uint32_t sh_base[2];
    // Commands the engine held back for this stage (jello_hip.cpp, Deferred): bbox_clear (bbox_clear.wgsl:13-24; this kernel
    // writes only the draw_flags / trans_ix words of the boxes, the min / max words are first used by k_flatten_bbox) and
    // the recording's Clear(bump) (render.go:237; nothing of flatten touches bump before k_flatten_items).
    if (absorb & JH_ABSORB_BBOX_CLEAR) {
        const uint32_t n_clear = umin_(cfg->layout.n_path, path_bboxes.n);
        for (uint32_t i = blockIdx.x * JL_WG + threadIdx.x; i < n_clear; i += gridDim.x * JL_WG) {
            path_bboxes.p[i].x0 = 0x7fffffff;
            path_bboxes.p[i].y0 = 0x7fffffff;
            path_bboxes.p[i].x1 = (int32_t)0x80000000;
            path_bboxes.p[i].y1 = (int32_t)0x80000000;
        }
    }
    if ((absorb & JH_ABSORB_BUMP_CLEAR) && blockIdx.x == 0u && threadIdx.x < 8u) bump_words[threadIdx.x] = 0u;
    uint32_t nh[PSC_BLOCKS][4], nl[PSC_BLOCKS][4];
    uint32_t eh[PSC_BLOCKS], el[PSC_BLOCKS];  // this thread's exclusive prefix of heavy / light items inside the workgroup
    uint32_t wg_h = 0u, wg_l = 0u;
#pragma unroll
    for (uint32_t blk = 0; blk < PSC_BLOCKS; blk++) {
        const uint32_t block = blockIdx.x * PSC_BLOCKS + blk;  // = the workgroup index of the recorded scan dispatch
        eh[blk] = 0u; el[blk] = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) { nh[blk][k] = 0u; nl[blk][k] = 0u; }
        if (block >= n_blocks) continue;  // uniform
        const uint32_t word_ix = block * JL_WG + threadIdx.x;
        const uint32_t tag_word = scene.rd(cfg->layout.pathtag_base + word_ix);
        MonoidK<5> pm;
        if (SCAN != 0) {  // ---- the scan: this thread's tag word ----
            MonoidK<5> prefix;
            if (SCAN == 1) prefix = parent_prefix(reduced, block, sh); else prefix = load_tm(reduced, block);
            MonoidK<5> tot5;
            if (SCAN == 1) __syncthreads();  // (parent_prefix has used sh)
            const MonoidK<5> ex = block_excl_scan_monoid<5>(reduce_tag(tag_word), sh, &tot5);
            pm = monoid_add(prefix, ex);
            if (tag_monoids.ok(word_ix)) {
                store_tm(&tag_monoids.p[word_ix], pm);
            } else {  // (robust access: a later read of a word behind the buffer sees zeros)
#pragma unroll
                for (int i = 0; i < 5; i++) pm.v[i] = 0u;
            }
            __syncthreads();  // (sh is used again below)
        } else {
            pm = load_tm(tag_monoids, word_ix);
        }
        // ---- the classification of the word's four tag bytes ----
        const uint32_t ix0 = word_ix * 4u;
        for (uint32_t q = 0u; q < 12u; q++) {
            const uint32_t sl = ix0 * 3u + q;
            if (sl < cap) counts[sl] = 0u;
        }
        uint32_t th = 0u, tl = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t ix = ix0 + k;
            if (ix >= n_tags) continue;
            // compute_tag_monoid (flatten.wgsl:668-682) with the word and its prefix at hand
            const MonoidK<5> tm = monoid_add(pm, reduce_tag(tag_word & ((1u << (k * 8u)) - 1u)));
            const uint32_t tag_byte = (tag_word >> (k * 8u)) & 0xffu;
            const uint32_t style_flags = scene.rd(cfg->layout.style_base + tm.v[3] - 2u);
            const bool is_stroke = (style_flags & JL_STYLE_FLAGS_STYLE) != 0u;
            const uint32_t path_ix = tm.v[4];
            if ((tag_byte & JL_PATH_TAG_PATH) != 0u && path_bboxes.ok(path_ix)) {  // flatten.wgsl:825-828
                path_bboxes.p[path_ix].draw_flags = ((style_flags & JL_STYLE_FLAGS_FILL) == 0u) ? 0u : 1u;
                path_bboxes.p[path_ix].trans_ix = tm.v[0] - 1u;
            }
            const uint32_t seg_type = tag_byte & JL_PATH_TAG_SEG_TYPE;
            const bool curved = seg_type != JL_PATH_TAG_LINETO;
            if (seg_type != 0u) {
                if (!is_stroke) {
                    if (curved) nh[blk][k] = 1u; else nl[blk][k] = 1u;
                } else if ((tag_byte & JL_PATH_TAG_SUBPATH_END) != 0u) {
                    if (curved) nl[blk][k] = 1u;
                } else {
                    if (curved) { nh[blk][k] = 2u; nl[blk][k] = 1u; } else { nl[blk][k] = 3u; }
                }
            }
            th += nh[blk][k];
            tl += nl[blk][k];
        }
        MonoidK<2> m, tot;
        m.v[0] = th; m.v[1] = tl;
        const MonoidK<2> e2 = block_excl_scan_monoid<2>(m, sh, &tot);
        eh[blk] = wg_h + e2.v[0];
        el[blk] = wg_l + e2.v[1];
        wg_h += tot.v[0];
        wg_l += tot.v[1];
        __syncthreads();
    }
    if (threadIdx.x == 0) {  // one atomic pair per workgroup (a hot word sustains only ~88 atomics/us)
        sh_base[0] = wg_h ? atomicAdd(&counters[0], wg_h) : 0u;
        sh_base[1] = wg_l ? atomicAdd(&counters[FL_CTR_LIGHT], wg_l) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t blk = 0; blk < PSC_BLOCKS; blk++) {
        const uint32_t ix0 = ((blockIdx.x * PSC_BLOCKS + blk) * JL_WG + threadIdx.x) * 4u;
        uint32_t ph = sh_base[0] + eh[blk], pl = sh_base[1] + el[blk];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t ix = ix0 + k;
            uint32_t sub = 0u;
            for (uint32_t q = 0; q < nh[blk][k]; q++, sub++, ph++)
                if (ph < cap) list[ph] = ix * 3u + sub;
            for (uint32_t q = 0; q < nl[blk][k]; q++, sub++, pl++)
                if (pl < cap) list[cap - 1u - pl] = ix * 3u + sub;
        }
    }
}

// Everything of an item except Euler flattening, which is returned as a job for flatten_euler_wave.
template <bool EMIT>
JD void run_item(const JlConfig* cfg, const Scene& s, Out<EMIT>& o, uint32_t slot, EulerJob& job, uint32_t& path_ix_out) {
    uint32_t ix = slot / 3u, sub = slot - ix * 3u;
    Seg g = load_seg(s, ix);
    uint32_t path_ix = g.tag.monoid.v[4];
    path_ix_out = path_ix;
    uint32_t style_ix = g.tag.monoid.v[3];
    uint32_t trans_ix = g.tag.monoid.v[0];
    uint32_t style_flags = g.style_flags;
    job.path_ix = path_ix; job.trans_ix = trans_ix;
    Xf transform;
    {
        uint32_t base = cfg->layout.transform_base + trans_ix * 6u;
        transform.m0 = u2f(s.scene.rd(base)); transform.m1 = u2f(s.scene.rd(base + 1u)); transform.m2 = u2f(s.scene.rd(base + 2u));
        transform.m3 = u2f(s.scene.rd(base + 3u)); transform.t0 = u2f(s.scene.rd(base + 4u)); transform.t1 = u2f(s.scene.rd(base + 5u));
    }
    CubicPoints pts = read_path_segment(s, g.tag, g.is_stroke);
    if (g.is_stroke) {
        float linewidth = u2f(s.scene.rd(cfg->layout.style_base + style_ix + 1u));
        float offset = 0.5f * linewidth;
        bool is_stroke_cap_marker = (g.tag.tag_byte & JL_PATH_TAG_SUBPATH_END) != 0u;
        if (is_stroke_cap_marker) {
            // open path start cap (flatten.wgsl:845-852)
            V2 tangent = cubic_start_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
            V2 offset_tangent = offset * normalize(tangent);
            V2 n = v2(offset_tangent.y * -1.0f, offset_tangent.x * 1.0f);
            draw_cap<EMIT>(o, path_ix, (style_flags & JL_STYLE_FLAGS_START_CAP_MASK) >> 2, pts.p0, pts.p0 - n, pts.p0 + n, -offset_tangent,
                           transform);
        } else {
            const float TT = TANGENT_THRESH * TANGENT_THRESH;
            V2 tan_prev = cubic_end_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
            if (dot(tan_prev, tan_prev) < TT) tan_prev = v2(TANGENT_THRESH, 0.0f);
            V2 offset_tangent = offset * normalize(tan_prev);
            V2 n_prev = v2(offset_tangent.y * -1.0f, offset_tangent.x * 1.0f);
            if (sub < 2u) {
                V2 tan_start = cubic_start_tangent(pts.p0, pts.p1, pts.p2, pts.p3);
                if (dot(tan_start, tan_start) < TT) tan_start = v2(TANGENT_THRESH, 0.0f);
                V2 n_start = offset * normalize(v2(-tan_start.y, tan_start.x));
                job.valid = true; job.cubic = pts; job.local_to_device = transform;
                if (sub == 0u) { job.offset = offset; job.start_p = pts.p0 + n_start; job.end_p = pts.p3 + n_prev; }
                else { job.offset = -offset; job.start_p = pts.p0 - n_start; job.end_p = pts.p3 - n_prev; }
            } else {
                PathTagData ntag = compute_tag_monoid(s, ix + 1u);  // read_neighboring_segment, :790-800
                CubicPoints npts = read_path_segment(s, ntag, true);
                bool n_is_closed = (ntag.tag_byte & JL_PATH_TAG_SEG_TYPE) == JL_PATH_TAG_LINETO;
                bool n_is_marker = (ntag.tag_byte & JL_PATH_TAG_SUBPATH_END) != 0u;
                bool do_join = !n_is_marker || n_is_closed;
                if (do_join) {
                    V2 tan_next = cubic_start_tangent(npts.p0, npts.p1, npts.p2, npts.p3);
                    if (dot(tan_next, tan_next) < TT) tan_next = v2(TANGENT_THRESH, 0.0f);
                    V2 tnn = normalize(tan_next);
                    V2 n_next = v2((offset * tnn.y) * -1.0f, (offset * tnn.x) * 1.0f);
                    draw_join<EMIT>(o, path_ix, style_flags, pts.p3, tan_prev, tan_next, n_prev, n_next, transform);
                } else {
                    draw_cap<EMIT>(o, path_ix, (style_flags & JL_STYLE_FLAGS_END_CAP_MASK), pts.p3, pts.p3 + n_prev, pts.p3 - n_prev, offset_tangent,
                                   transform);
                }
            }
        }
    } else {
        job.valid = true; job.cubic = pts; job.local_to_device = transform;
        job.offset = 0.0f; job.start_p = pts.p0; job.end_p = pts.p3;
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-cooperative subdivision (the Euler jobs of k_flatten_items).
//
// flatten_euler (flatten.wgsl:328-477) walks the dyadic intervals of [0,1] depth first: try [t0, t0+dt]; accept it as
// one Euler piece, or halve dt.  Whether an interval is accepted depends only on the cubic and the interval: the state
// carried from piece to piece (last_p, last_q, last_t) is the cubic evaluated at the interval's start (with the WGSL's
// own fix-up where the derivative vanishes), which every node can recompute with the same operations on the same
// values.  So the nodes of the subdivision trees of a whole batch of jobs are independent work items: the wave keeps
// them on a stack in LDS and evaluates 64 of them per step, whichever jobs they belong to, instead of one job per
// lane (1...27 dependent attempts per job: a third of the lanes busy).  A rejected node pushes its two halves, an
// accepted one becomes a piece: it reserves its temp slots and leaves its record exactly like the sequential walk.
// What the sequential walk gets for free -- the index of a piece's first line inside its item = the lines of the
// pieces before it -- is filled in when the batch has drained: every piece adds up the line counts of its job's
// pieces with a smaller t0 (a linked list per job in LDS; a job has ~5 pieces).
// Bounds: the stack is LIFO, so it holds at most 64 nodes per tree level (+128); if it or the piece list of a batch
// overflows, or a tree goes deeper than FLQ_MAX_LEVEL (dt < 2^-9; none of the test scenes goes below 2^-6), the
// unfinished jobs of the batch fall back to the sequential walk.
// ------------------------------------------------------------------------------------------------
// (capacities, overridable only so that tools/soak_flatten_fallback.sh can force the fall-back: results do not depend on them)
#ifndef FLQ_STACK
#define FLQ_STACK 448u
#endif
#ifndef FLQ_LEAVES
#define FLQ_LEAVES 640u
#endif
#ifndef FLQ_MAX_LEVEL
#define FLQ_MAX_LEVEL 9u  // deeper trees (t0 no longer fits the 9 key bits) take the sequential walk
#endif
struct FlBatch {
    uint32_t jhead[64];                // newest piece of the job + 1 (0 = none), lane = job
    uint32_t stack[FLQ_STACK];         // job | level << 6 | t0_u << 11
    uint32_t l_tpos[FLQ_LEAVES];       // the accepted node (same packing as the stack); after phase B: the piece's first line in its job
    uint16_t l_key[FLQ_LEAVES];        // t0 in units of 2^-9 << 7 | n
    uint16_t l_link[FLQ_LEAVES];       // job << 10 | next piece of the job + 1
    uint32_t unsure[128];              // nodes the transcendental-free test could not decide (flatten_fast.h): < 64 waiting + 64 new;
                                       // after phase B: [job] = lines of the batch's jobs before it
    uint32_t n_stack, n_leaves, bail, n_unsure;
};
// The job state (control points, scale, offset, ids) stays in the registers of the lane that set the job up; the lane
// that evaluates one of its nodes fetches it with ds_bpermute (no LDS storage: occupancy is bound by registers only).
JD float lanef(float v, uint32_t src) { return u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)f2u(v))); }
JD uint32_t laneu(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v); }

struct NodeResult {
    bool accept;
    CubicParams cp;
    V2 es_p0, es_p1;
    bool ends_at_one;
};
// The two ends of the interval [t0_u, t0_u + 1] * 2^-level as the sequential walk of flatten.wgsl:362-383 sees them: the
// state it carries into the interval (the end of the piece that ended at t0) and the point / derivative at t1.
struct NodeEnds {
    V2 last_p, last_q;
    float last_t;
    V2 point, deriv;
    float t1;
};
JD NodeEnds node_ends(V2 p0, V2 p1, V2 p2, V2 p3, uint32_t level, uint32_t t0_u) {
    const float dt = u2f((127u - level) << 23);  // 2^-level
    const float t0 = (float)t0_u * dt;
    NodeEnds r;
    if (t0_u == 0u) {
        r.last_p = p0;
        r.last_q = p1 - p0;
        if (dot(r.last_q, r.last_q) < DERIV_THRESH_SQUARED) r.last_q = eval_cubic_and_deriv(p0, p1, p2, p3, DERIV_EPS).deriv;
        r.last_t = 0.0f;
    } else {
        PointDeriv pq0 = eval_cubic_and_deriv(p0, p1, p2, p3, t0);
        r.last_t = t0;
        if (dot(pq0.deriv, pq0.deriv) < DERIV_THRESH_SQUARED) {  // (t0 < 1 here: the piece before took the adjusted end)
            PointDeriv n0 = eval_cubic_and_deriv(p0, p1, p2, p3, t0 - DERIV_EPS);
            pq0.deriv = n0.deriv;
            pq0.point = n0.point;
            r.last_t = t0 - DERIV_EPS;
        }
        r.last_p = pq0.point;
        r.last_q = pq0.deriv;
    }
    float t1 = t0 + dt;
    PointDeriv this_pq1 = eval_cubic_and_deriv(p0, p1, p2, p3, t1);
    if (dot(this_pq1.deriv, this_pq1.deriv) < DERIV_THRESH_SQUARED) {
        PointDeriv new_pq1 = eval_cubic_and_deriv(p0, p1, p2, p3, t1 - DERIV_EPS);
        this_pq1.deriv = new_pq1.deriv;
        if (t1 < 1.0f) {
            this_pq1.point = new_pq1.point;
            t1 = t1 - DERIV_EPS;
        }
    }
    r.point = this_pq1.point;
    r.deriv = this_pq1.deriv;
    r.t1 = t1;
    return r;
}
// One attempt of flatten.wgsl:362-403 for the interval [t0_u, t0_u + 1] * 2^-level of the cubic (p0..p3): the pinned sequence.
JD NodeResult node_test(V2 p0, V2 p1, V2 p2, V2 p3, float scale, uint32_t level, uint32_t t0_u) {
    const float tol = 0.25f;
    const float dt = u2f((127u - level) << 23);
    const NodeEnds ne = node_ends(p0, p1, p2, p3, level, t0_u);
    const float actual_dt = ne.t1 - ne.last_t;
    NodeResult r;
    r.cp = cubic_from_points_derivs(ne.last_p, ne.point, ne.last_q, ne.deriv, actual_dt);
    r.accept = r.cp.err * scale <= tol || dt <= SUBDIV_LIMIT;
    r.es_p0 = ne.last_p;
    r.es_p1 = ne.point;
    r.ends_at_one = ne.t1 == 1.0f;
    return r;
}
// The same attempt, DECIDED without transcendentals where that is provably safe (flatten_fast.h): FF_ACCEPT / FF_REJECT
// agree with node_test().accept, FF_UNSURE means "run node_test".  (dt > SUBDIV_LIMIT for every level the cooperative
// subdivision visits, FLQ_MAX_LEVEL < 16, so the test is the error test alone.)
JD int node_test_fast(V2 p0, V2 p1, V2 p2, V2 p3, float scale, uint32_t level, uint32_t t0_u, float* v_est, float* delta) {
    const float tol = 0.25f;
    const NodeEnds ne = node_ends(p0, p1, p2, p3, level, t0_u);
    const float actual_dt = ne.t1 - ne.last_t;
    // cubic_from_points_derivs (flatten.wgsl:94-114) up to d0 / d1: chord and h0 / h1 with the same operations on the same
    // values, the roots and the quotient with the 1-ulp instructions (flatten_fast.h charges them)
    const V2 chord = ne.point - ne.last_p;
    const float chord_squared = dot(chord, chord);
    const V2 q0 = ne.last_q, q1 = ne.deriv;
    *v_est = 0.0f; *delta = 0.0f;
    if (chord_squared < DERIV_THRESH_SQUARED) {  // no transcendentals in this branch: decided exactly, by the pinned operations
        const float chord_err = sqrt_((float)(9.0 / 32.0) * (dot(q0, q0) + dot(q1, q1))) * actual_dt;
        return (chord_err * scale <= tol) ? ffast::FF_ACCEPT : ffast::FF_REJECT;
    }
    const float chord_len = FF_SQRT(chord_squared);
    const float sc = actual_dt * FF_RCP(chord_squared);
    const V2 h0 = v2(q0.x * chord.x + q0.y * chord.y, q0.y * chord.x - q0.x * chord.y);
    const V2 h1 = v2(q1.x * chord.x + q1.y * chord.y, q1.x * chord.y - q1.y * chord.x);
    const float len0 = FF_SQRT(h0.x * h0.x + h0.y * h0.y), len1 = FF_SQRT(h1.x * h1.x + h1.y * h1.y);
    return ffast::ff_decide(h0.x, h0.y, len0, h1.x, h1.y, len1, len0 * sc, len1 * sc, chord_len, scale, tol, v_est, delta);
}
// th0, th1 and chord_len of an ACCEPTED interval (the pinned atan2; err is not needed any more)
JD CubicParams piece_angles(const NodeEnds& ne) {
    const V2 chord = ne.point - ne.last_p;
    const float chord_squared = dot(chord, chord);
    CubicParams r;
    r.err = 0.0f;
    if (chord_squared < DERIV_THRESH_SQUARED) {
        r.th0 = 0.0f; r.th1 = 0.0f; r.chord_len = DERIV_THRESH;
        return r;
    }
    const V2 q0 = ne.last_q, q1 = ne.deriv;
    const V2 h0 = v2(q0.x * chord.x + q0.y * chord.y, q0.y * chord.x - q0.x * chord.y);
    const V2 h1 = v2(q1.x * chord.x + q1.y * chord.y, q1.x * chord.y - q1.y * chord.x);
    r.th0 = atan2_(h0.y, h0.x);
    r.th1 = atan2_(h1.y, h1.x);
    r.chord_len = sqrt_(chord_squared);
    return r;
}

// What flatten.wgsl:404-447 computes for an accepted interval besides the lines themselves: the Euler parameters and
// the number of lines.
struct PieceParams {
    EulerParams ep;
    float n, noff, int0, integral;
    uint32_t n_u, robust;
};
JD PieceParams piece_params(const CubicParams& cp, float scale, float offset) {
    const float tol = 0.25f;
    PieceParams r;
    r.ep = es_params_from_angles(cp.th0, cp.th1);
    const EulerParams& ep = r.ep;
    float k0 = ep.k0 - 0.5f * ep.k1;
    float k1 = ep.k1;
    float normalized_offset = offset / cp.chord_len;
    float dist_scaled = normalized_offset * ep.ch;
    float scale_multiplier = sqrt_(0.125f * scale * cp.chord_len / (ep.ch * tol));
    float a = 0.0f, b = 0.0f, integral = 0.0f, int0 = 0.0f, n_frac;
    uint32_t robust = 0u;
    if (abs_(k1) < K1_THRESH) {
        float k = ep.k0;
        n_frac = sqrt_(abs_(k * (k * dist_scaled + 1.0f)));
        robust = 1u;
    } else if (abs_(dist_scaled) < DIST_THRESH) {
        a = k1;
        b = k0;
        int0 = pow_1_5_signed(b);
        float int1 = pow_1_5_signed(a + b);
        integral = int1 - int0;
        n_frac = (float)(2.0 / 3.0) * integral / a;
        robust = 2u;
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
    r.n = clamp_(ceil_(n_frac * scale_multiplier), 1.0f, 100.0f);
    r.n_u = to_u32(r.n);
    r.noff = normalized_offset;
    r.int0 = int0;
    r.integral = integral;
    r.robust = robust;
    return r;
}

#ifdef FL_FAST_CHECK
__device__ uint32_t g_ff_stats[8];  // nodes tested, undecided, contradictions, bound violations
#endif
#ifndef FL_WAVES_PER_EU
#define FL_WAVES_PER_EU 4  // 128 VGPRs with 8 spilled, reloaded in the batch set-up and the fallback (marked unlikely): 190.7 us against 194.0 with 3 waves
                           // and 149 VGPRs (same box); without the inlined fallback the loop needs 131 and runs in 182 us at 4 waves
#endif
#ifndef FL_BLOCKS_PER_CU
#define FL_BLOCKS_PER_CU 4  // = what is resident at 4 waves per SIMD (round 5, C3, same box: 162 / 139 / 129 / 136 / 138 / 147 us with 2 / 3 / 4 / 5 / 6 / 8;
                            // a fifth workgroup per CU only starts when one has finished and then has a full share of batches in front of it)
#endif
__global__ __launch_bounds__(JL_WG) __attribute__((amdgpu_waves_per_eu(FL_WAVES_PER_EU, FL_WAVES_PER_EU))) void k_flatten_items(const JlConfig* __restrict__ cfg, Buf<uint32_t> scene, Buf<JlTagMonoid> tag_monoids,
                                                         Buf<JlPathBbox> path_bboxes, const uint32_t* __restrict__ list,
                                                         uint32_t* __restrict__ counters, uint32_t cap, uint32_t* __restrict__ counts,
                                                         FlTemp T, uint32_t debug) {
    __shared__ uint32_t sh_item;  // next position of this workgroup's share of the item list
    __shared__ FlBatch sh_batch[JL_WG / 64];
    Scene s;
    s.cfg = cfg; s.scene = scene; s.tag_monoids = tag_monoids;
    uint32_t n_heavy = umin_(counters[0], cap), n_light = umin_(counters[FL_CTR_LIGHT], cap - n_heavy);
    uint32_t n = n_heavy + n_light;
    if (blockIdx.x * 64u >= n) return;  // uniform: this workgroup's share of the list is empty
    atan_tab_fill();
    if (threadIdx.x == 0) sh_item = 0u;
    __syncthreads();
    const uint32_t lane = lane_id();
    FlBatch& B = sh_batch[threadIdx.x >> 6];
    Out<true> o;
    o.cfg = cfg; o.T = T; o.home_s = o.home_r = blockIdx.x % T.K; o.failed_s = o.failed_r = false; o.slot = 0u;
    if ((debug & 1u) != 0u) o.home_s = o.home_r = 0u;  // (jh_debug_flatten_regions: regions fill up and are left behind on ordinary scenes)
#ifdef FL_SOAK_HOME0  // (tools/soak_flatten_fallback.sh: the same for every frame of the build)
    o.home_s = o.home_r = 0u;
#endif
    o.cursor = 0u; o.a_first = 0u; o.a_spos = 0u; o.a_rpos = 0u;
    // Work distribution: the item list (heavy items first) is dealt to the workgroups in chunks of 64, round robin; a
    // wave takes one chunk (= one batch) at a time through the workgroup's LDS counter.
    for (;;) {
        uint32_t base = 0u;
        if (lane == 0u) base = atomicAdd(&sh_item, 64u);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t t_first = ((base >> 6) * gridDim.x + blockIdx.x) * 64u;
        if (t_first >= n) break;
        const uint32_t t = t_first + lane;
        // (a region of the temporary that is full is left behind at this uniform point)
        if (__builtin_amdgcn_ballot_w64(o.failed_s) != 0ull) { o.home_s = fl_next_home(T, o.home_s, 0); o.failed_s = false; }
        if (__builtin_amdgcn_ballot_w64(o.failed_r) != 0ull) { o.home_r = fl_next_home(T, o.home_r, 1); o.failed_r = false; }
        // ---- set up the batch: direct items are emitted at once, Euler jobs go to LDS ----
        EulerJob job;
        job.valid = false; job.path_ix = 0u; job.trans_ix = 0u; job.offset = 0.0f;
        job.start_p = v2(0, 0); job.end_p = v2(0, 0);
        job.cubic.p0 = job.cubic.p1 = job.cubic.p2 = job.cubic.p3 = v2(0, 0);
        job.local_to_device = xf_identity();
        uint32_t slot = FL_INVALID;
        if (t < n) {
            slot = t < n_heavy ? list[t] : list[cap - 1u - (t - n_heavy)];
            o.slot = slot; o.cursor = 0u; o.a_first = 0u; o.a_spos = 0u; o.a_rpos = 0u;
            uint32_t path_ix;
            run_item<true>(cfg, s, o, slot, job, path_ix);
            if (!job.valid) counts[slot] = o.cursor;  // a direct item is complete
        }
        EulerLane e;
        euler_begin(e, job);  // (transforms the control points of a fill, scale of an offset curve, degenerate test)
        const bool active = job.valid && !e.done;
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) continue;  // uniform: nothing to subdivide in this batch
        B.jhead[lane] = 0u;
        {
            const uint64_t am = __builtin_amdgcn_ballot_w64(active);
            if (active) B.stack[(uint32_t)__builtin_popcountll(am & ((1ull << lane) - 1ull))] = lane;  // root: level 0, t0_u 0
            if (lane == 0u) { B.n_stack = (uint32_t)__builtin_popcountll(am); B.n_leaves = 0u; B.bail = 0u; B.n_unsure = 0u; }
        }
        wave_fence();
#ifdef FL_SPLIT_NO_A  // (measurement builds only, tools/flatten_split.sh: the kernel without its subdivision -- results are wrong)
        if (lane == 0u) B.n_stack = 0u;
        wave_fence();
#endif
        // ---- phase A: drain the stack -- DECISIONS only.  A node is accepted, rejected (its halves are pushed) or left
        // undecided by the transcendental-free test (flatten_fast.h; ~0.1 % of the nodes); the undecided ones wait in a list of
        // their own and get the pinned sequence in rounds of up to 64 when the stack has run dry (or 64 are waiting). ----
        for (;;) {
            const uint32_t ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)B.n_stack);
            const uint32_t nu = (uint32_t)__builtin_amdgcn_readfirstlane((int)B.n_unsure);
            if (ns == 0u && nu == 0u) break;
            const bool exact_round = ns == 0u || nu >= 64u;  // uniform
            const uint32_t take = umin_(exact_round ? nu : ns, 64u);
            const bool has = lane < take;
            const uint32_t node = has ? (exact_round ? B.unsure[nu - 1u - lane] : B.stack[ns - 1u - lane]) : 0u;
            wave_fence();
            const uint32_t j = node & 63u, level = (node >> 6) & 31u, t0_u = node >> 11;
            // the job's state from its owner lane (every lane takes part in the permutes)
            const V2 jp0 = v2(lanef(e.p0.x, j), lanef(e.p0.y, j)), jp1 = v2(lanef(e.p1.x, j), lanef(e.p1.y, j));
            const V2 jp2 = v2(lanef(e.p2.x, j), lanef(e.p2.y, j)), jp3 = v2(lanef(e.p3.x, j), lanef(e.p3.y, j));
            const float scale = lanef(e.scale, j);
            int kind = ffast::FF_UNSURE;
            if (exact_round) {
                if (has) kind = node_test(jp0, jp1, jp2, jp3, scale, level, t0_u).accept ? ffast::FF_ACCEPT : ffast::FF_REJECT;
            } else if (has) {
                float v_est, delta;
                kind = node_test_fast(jp0, jp1, jp2, jp3, scale, level, t0_u, &v_est, &delta);
#ifdef FL_FAST_CHECK  // (make VARIANT=ffcheck: both paths on every node; contradictions and bound violations are counted)
            }
            if (!exact_round) {  // (uniform; every lane takes part in the ballots: one atomic per wave and counter, not one per node)
                bool c_unsure = false, c_contra = false, c_viol = false;
                if (has) {
                    float v2_, d2_;
                    const int k2 = node_test_fast(jp0, jp1, jp2, jp3, scale, level, t0_u, &v2_, &d2_);
                    const NodeResult rx = node_test(jp0, jp1, jp2, jp3, scale, level, t0_u);
                    const float vx = rx.cp.err * scale;
                    c_unsure = k2 == ffast::FF_UNSURE;
                    c_contra = (k2 == ffast::FF_ACCEPT && !rx.accept) || (k2 == ffast::FF_REJECT && rx.accept);
                    c_viol = (d2_ > 0.0f && !(abs_(v2_ - vx) <= d2_)) || (d2_ == 0.0f && k2 != ffast::FF_UNSURE && v2_ != 0.0f);
                }
                const uint32_t n0 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(has));
                const uint32_t n1 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c_unsure));
                const uint32_t n2 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c_contra));
                const uint32_t n3 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c_viol));
                if (lane == 0u) {
                    atomicAdd(&g_ff_stats[0], n0);
                    if (n1) atomicAdd(&g_ff_stats[1], n1);
                    if (n2) atomicAdd(&g_ff_stats[2], n2);
                    if (n3) atomicAdd(&g_ff_stats[3], n3);
                }
#endif
            }
            const bool acc = has && kind == ffast::FF_ACCEPT, rej = has && kind == ffast::FF_REJECT, uns = has && kind == ffast::FF_UNSURE;
            const uint64_t accm = __builtin_amdgcn_ballot_w64(acc), rejm = __builtin_amdgcn_ballot_w64(rej), unsm = __builtin_amdgcn_ballot_w64(uns);
            const uint32_t n_acc = (uint32_t)__builtin_popcountll(accm), n_rej = (uint32_t)__builtin_popcountll(rejm);
            const uint32_t n_uns = (uint32_t)__builtin_popcountll(unsm);
            const uint32_t nl = (uint32_t)__builtin_amdgcn_readfirstlane((int)B.n_leaves);
            const uint32_t ns_left = exact_round ? ns : ns - take, nu_left = exact_round ? nu - take : nu;
            const bool too_deep = __builtin_amdgcn_ballot_w64(rej && level + 1u > FLQ_MAX_LEVEL) != 0ull;
            if (too_deep || ns_left + 2u * n_rej > FLQ_STACK || nl + n_acc > FLQ_LEAVES) {  // uniform: give up on the unfinished jobs
                if (lane == 0u) B.bail = 1u;
                wave_fence();
                break;
            }
            const uint64_t below = (1ull << lane) - 1ull;
            if (rej) {
                const uint32_t pos = ns_left + 2u * (uint32_t)__builtin_popcountll(rejm & below);
                B.stack[pos] = j | ((level + 1u) << 6) | ((2u * t0_u + 1u) << 11);
                B.stack[pos + 1u] = j | ((level + 1u) << 6) | ((2u * t0_u) << 11);  // the left half on top: popped first
            }
            if (acc) {  // a piece: its record is written in phase B
                const uint32_t li = nl + (uint32_t)__builtin_popcountll(accm & below);
                B.l_tpos[li] = node;
                B.l_key[li] = (uint16_t)((t0_u << (FLQ_MAX_LEVEL - level)) << 7);
                const uint32_t prev = atomicExch(&B.jhead[j], li + 1u);
                B.l_link[li] = (uint16_t)((j << 10) | prev);
            }
            if (uns) B.unsure[nu_left + (uint32_t)__builtin_popcountll(unsm & below)] = node;  // (nu_left < 64 in a fast round)
            if (lane == 0u) { B.n_stack = ns_left + 2u * n_rej; B.n_leaves = nl + n_acc; B.n_unsure = nu_left + n_uns; }
            wave_fence();
        }
        // ---- phase B: the pieces, 64 at a time whichever jobs they belong to: the pinned angles (two atan2), the Euler
        // parameters and the line count (flatten.wgsl:404-447), the record.  The records of a batch are one dense range in the
        // order of the piece list (one returning atomic per batch; its round trip passes under the first pass's arithmetic).
        // A batch that gave up (bail) leaves no pieces: all its jobs take the sequential walk below. ----
        const bool bail = (uint32_t)__builtin_amdgcn_readfirstlane((int)B.bail) != 0u;
        const uint32_t nl = bail ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)B.n_leaves);
        uint32_t rec_base = 0u;
#if defined(FL_ISPLIT) && FL_ISPLIT == 3  // (measurement builds only: the batches allocate nothing -- results are wrong)
        if (nl != 0u && lane == 0u) rec_base = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 640u % (T.K * T.R - 640u);
#else
        if (nl != 0u && lane == 0u) rec_base = fl_grab<1>(T, o.home_r, nl, o.failed_r);
#endif
        rec_base = (uint32_t)__builtin_amdgcn_readlane((int)rec_base, 0);
#ifdef FL_SPLIT_NO_B  // (measurement builds only: no piece is written)
        for (uint32_t lbase = nl; lbase < nl; lbase += 64u) {
#else
        for (uint32_t lbase = 0u; lbase < nl; lbase += 64u) {
#endif
            const uint32_t li = lbase + lane;
            const bool has = li < nl;
            const uint32_t node = has ? B.l_tpos[li] : 0u;
            const uint32_t j = node & 63u, level = (node >> 6) & 31u, t0_u = node >> 11;
            const V2 jp0 = v2(lanef(e.p0.x, j), lanef(e.p0.y, j)), jp1 = v2(lanef(e.p1.x, j), lanef(e.p1.y, j));
            const V2 jp2 = v2(lanef(e.p2.x, j), lanef(e.p2.y, j)), jp3 = v2(lanef(e.p3.x, j), lanef(e.p3.y, j));
            const float scale = lanef(e.scale, j), offset = lanef(e.offset, j);
            // a piece needs the ids of its job and, if it is the item's first or last, the item's end points
            const uint32_t j_slot = laneu(slot, j), j_path = laneu(e.path_ix, j), j_trans = laneu(e.trans_ix, j);
            const float j_tsx = lanef(e.t_start.x, j), j_tsy = lanef(e.t_start.y, j), j_tex = lanef(e.t_end.x, j), j_tey = lanef(e.t_end.y, j);
            if (has) {
                const NodeEnds ne = node_ends(jp0, jp1, jp2, jp3, level, t0_u);
                const CubicParams cp = piece_angles(ne);
                const PieceParams pp = piece_params(cp, scale, offset);
                const uint32_t fl = pp.robust | ((ne.t1 == 1.0f) ? 4u : 0u) | ((offset >= 0.0f) ? 8u : 0u) | ((offset == 0.0f) ? 16u : 0u) |
                                    ((t0_u == 0u) ? 32u : 0u);
                const uint32_t r = rec_base + li;
                if (rec_base != FL_INVALID)  // (word 12 of a piece that is not its item's first -- the index of its first line -- is written below)
                    piece_record_write(T.recs + (size_t)r * 4u, ne.last_p, ne.point, pp.ep.th0, pp.ep.th1, pp.int0, pp.integral, pp.noff, j_slot, j_path,
                                       j_trans, fl, 0u, v2(j_tsx, j_tsy), v2(j_tex, j_tey));
                B.l_key[li] = (uint16_t)(B.l_key[li] | pp.n_u);
            }
        }
        wave_fence();
        if (nl != 0u) {  // uniform
            // ---- the lines of a job; the slots of the batch: ONE range (one returning atomic), the jobs in lane order ----
            uint32_t total = 0u;
            if (active)
                for (uint32_t q = B.jhead[lane]; q != 0u; q = B.l_link[q - 1u] & 1023u) total += B.l_key[q - 1u] & 127u;
            if (active) counts[slot] = total;
            const uint32_t incl = wave_incl_scan_u32(total);
            const uint32_t batch_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            // (a batch of more than FL_MAX_GRAB lines -- 64 jobs of 800 -- allocates job by job: no allocation may be larger)
            const bool together = batch_total <= ((debug & 4u) != 0u ? 48u : FL_MAX_GRAB);  // uniform (jh_debug_flatten_regions bit 2: tests)
            uint32_t grab = 0u;
            if (together) {
#if defined(FL_ISPLIT) && FL_ISPLIT == 3
                if (lane == 0u) grab = FL_INVALID;
#else
                if (lane == 0u) grab = fl_grab<0>(T, o.home_s, batch_total, o.failed_s);
#endif
            } else if (active && total != 0u) {
                grab = fl_grab<0>(T, o.home_s, total, o.failed_s);
            }
            // ---- a piece's first line = the lines of its job's pieces before it (in the meantime the atomic returns) ----
            for (uint32_t li = lane; li < nl; li += 64u) {
                const uint32_t j = B.l_link[li] >> 10, key = B.l_key[li];
                uint32_t first = 0u;
                for (uint32_t q = B.jhead[j]; q != 0u; q = B.l_link[q - 1u] & 1023u) {
                    const uint32_t k2 = B.l_key[q - 1u];
                    if ((k2 >> 7) < (key >> 7)) first += k2 & 127u;
                }
                B.l_tpos[li] = first;
            }
            {   // the first slot of every job
                const uint32_t base0 = (uint32_t)__builtin_amdgcn_readlane((int)grab, 0);
                B.unsure[lane] = together ? (base0 == FL_INVALID ? FL_INVALID : base0 + (incl - total)) : grab;
            }
            wave_fence();
            // ---- the slots: in the canonical order of the batch's lines (job, piece, line), so that k_flatten_lines reads
            // consecutive slots and writes consecutive lines ----
            for (uint32_t li = lane; li < nl; li += 64u) {
                const uint32_t j = B.l_link[li] >> 10, key = B.l_key[li], first = B.l_tpos[li];
                const uint32_t r = rec_base + li, jb = B.unsure[j];
                if (rec_base != FL_INVALID && (key >> 7) != 0u) ((uint32_t*)T.recs)[(size_t)r * 16u + 12u] = first;  // (t0 != 0: not the item's first piece)
                if (jb == FL_INVALID) continue;
                if (rec_base != FL_INVALID) piece_slots_write(T.sinfo, jb + first, key & 127u, r);
                else for (uint32_t i = 0u; i < (key & 127u); i++) T.sinfo[jb + first + i] = make_uint2(0u, 0u);  // (slots without a record: empty)
            }
            wave_fence();
        }
        if (__builtin_expect(bail, 0)) {  // uniform, rare: the sequential walk for all jobs of the batch
            o.slot = slot; o.cursor = 0u; o.a_first = 0u; o.a_spos = 0u; o.a_rpos = 0u;
            bool have = active;
            auto finish = [&]() -> bool {
                if (e.done && have) {
                    counts[o.slot] = o.cursor;
                    have = false;
                }
                return __builtin_amdgcn_ballot_w64(!e.done) != 0ull;
            };
            flatten_euler_wave(o, e, finish);
        }
    }
}

// One thread per temporary slot: the Euler line (or the directly emitted line) that lives there, moved to
// lines[bases[item] + k], the canonical (tag byte, emission order) LineSoup position.
#ifndef FL_LINES_WAVES_PER_EU
#define FL_LINES_WAVES_PER_EU 4  // C3: 152 / 121 / 106 / 120 / 159 us at 2 / 3 / 4 / 5 / 8 (tools/sweep_flatten.sh)
#endif
__global__ __launch_bounds__(JL_WG) __attribute__((amdgpu_waves_per_eu(FL_LINES_WAVES_PER_EU, FL_LINES_WAVES_PER_EU))) void k_flatten_lines(const JlConfig* __restrict__ cfg, Buf<uint32_t> scene, const uint32_t* __restrict__ counters,
                                                         FlTemp T, const uint32_t* __restrict__ bases, uint32_t n_slots, Buf<JlLineSoup> lines) {
    atan_tab_fill();
    const uint32_t lines_lim = umin_(cfg->lines_size, lines.n);
    const uint2* __restrict__ sinfo = T.sinfo;
    const uint4* __restrict__ recs = T.recs;
    const uint32_t n_r = T.K * T.R;
    // Work units = JL_WG consecutive slots of a region, below the region's cursor.  Every such slot was written by
    // k_flatten_items (an allocation is exact; the slots a region could not give away are marked empty), and slot_info says
    // which record a slot belongs to and which of the record's lines it is: no search, no neighbours.
    uint32_t ru[FL_MAX_REGIONS], used[FL_MAX_REGIONS];  // units of the regions before region c, slots in use of region c (uniform; unrolled loops: registers)
    uint32_t units = 0u;
#pragma unroll
    for (uint32_t c = 0u; c < FL_MAX_REGIONS; c++) {
        ru[c] = units;
        used[c] = c < T.K ? umin_(counters[FL_CTR_CURSOR + FL_CUR_STRIDE * c], T.R) : 0u;
        units += (used[c] + JL_WG - 1u) / JL_WG;
    }
    auto fetch_info = [&](uint32_t u) -> uint2 {
        if (u >= units) return make_uint2(0u, 0u);
        uint32_t c = 0u, first_unit = 0u, in_use = used[0];
#pragma unroll
        for (uint32_t q = 1u; q < FL_MAX_REGIONS; q++)
            if (u >= ru[q] && used[q] != 0u) { c = q; first_unit = ru[q]; in_use = used[q]; }
        const uint32_t in_region = (u - first_unit) * JL_WG + threadIdx.x;
        return in_region < in_use ? sinfo[(size_t)c * T.R + in_region] : make_uint2(0u, 0u);
    };
    // state of a slot: 0 nothing to do, 1 complete line (copied), 2 line i of a piece
    auto state_of = [&](uint2 si) -> uint32_t {
        if (si.x >= n_r) return 0u;
        return (si.y & FL_INFO_PIECE) != 0u ? 2u : ((si.y & FL_INFO_DIRECT) != 0u ? 1u : 0u);
    };
    auto fetch_rec = [&](uint32_t state, uint32_t r, uint4& r0, uint4& r1, uint4& r2, uint4& r3) {
        r0 = r1 = r2 = r3 = make_uint4(0u, 0u, 0u, 0u);
        const uint4* rec = recs + (size_t)r * 4u;
        if (state != 0u) { r0 = rec[0]; r1 = rec[1]; }
        if (state == 2u) { r2 = rec[2]; r3 = rec[3]; }
    };
    // Software pipeline over the work units of a workgroup, two deep: while unit u is evaluated, the records of unit u + G
    // (G = gridDim.x) are on their way and so is the slot_info of unit u + 2 G.
    const uint32_t G = gridDim.x;
    uint2 si_a = fetch_info(blockIdx.x), si_b = fetch_info(blockIdx.x + G);
    uint32_t st_a = state_of(si_a);
    uint4 a0, a1, a2, a3;
    fetch_rec(st_a, si_a.x, a0, a1, a2, a3);
    for (uint32_t u = blockIdx.x; u < units; u += G) {  // uniform per workgroup
        // the current unit: its records were requested one trip ago
        const uint32_t st = st_a;
        const uint2 si = si_a;
        const uint4 r0 = a0, r1 = a1, r2 = a2, r3 = a3;
        // the next unit: its slot_info was requested one trip ago; request its records
        si_a = si_b;
        st_a = state_of(si_a);
        fetch_rec(st_a, si_a.x, a0, a1, a2, a3);
        // the unit after that: slot_info
        si_b = fetch_info(u + 2u * G);
        // What a slot writes: up to three 8-byte words of the line buffer (a LineSoup is {path, pad | p0 | p1} = 3 words).
        // A piece line stores its end point as p1 of its own record and, with the header, as p0 of the next one: the words
        // 3 dst + 2 ... 3 dst + 4, 24 contiguous bytes.  They are stored straight from the lanes.  (Rounds 3 and 4 staged them
        // through LDS so that a store instruction wrote 64 consecutive words: the slots were in allocation order then, a wave's
        // lines lay scattered piece by piece and the kernel was bound by its write requests -- 117 -> 207 us with every store issued
        // twice.  With the slots in canonical order consecutive lanes write consecutive lines; the staging cost more than it
        // saved: 76.4 -> 74.8 us without it, and 9 KB of LDS per workgroup less.)
        uint32_t widx[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
        uint2 wdat[3] = {make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u)};
        if (st == 1u) {  // complete line: copy
            const uint32_t item = r0.x, k = r0.y;
            if (item < n_slots) {
                const uint32_t dst = bases[item] + k;
                if (dst < lines_lim) {
                    widx[0] = dst * 3u; wdat[0] = make_uint2(r0.z, 0u);
                    widx[1] = dst * 3u + 1u; wdat[1] = make_uint2(r1.x, r1.y);
                    widx[2] = dst * 3u + 2u; wdat[2] = make_uint2(r1.z, r1.w);
                }
            }
        } else if (st == 2u) {
            const uint32_t n_u = (si.y >> 8) & 127u, i = si.y & 127u;
            const uint32_t flags = r2.w & 63u;
            const bool last_of_item = i + 1u == n_u && (flags & 4u) != 0u;
            const uint32_t slot = r2.y, k = ((flags & 32u) != 0u ? 0u : r3.x) + i;
            // (the transform and the line base only depend on the record: requested here, in front of the Euler evaluation, their
            // round trips pass under its ~500 instructions instead of following them)
            Xf tr;
            if ((flags & 16u) != 0u) {
                tr = xf_identity();
            } else {
                uint32_t tb = cfg->layout.transform_base + (r2.w >> 6) * 6u;
                tr.m0 = u2f(scene.rd(tb)); tr.m1 = u2f(scene.rd(tb + 1u)); tr.m2 = u2f(scene.rd(tb + 2u));
                tr.m3 = u2f(scene.rd(tb + 3u)); tr.t0 = u2f(scene.rd(tb + 4u)); tr.t1 = u2f(scene.rd(tb + 5u));
            }
            const uint32_t slot_base = slot < n_slots ? bases[slot] : 0u;
            V2 lp1;
            if (last_of_item) {
                lp1 = v2(u2f(r3.z), u2f(r3.w));
#if defined(FL_LSPLIT) && FL_LSPLIT == 1  // (measurement builds only, tools/lines_split.sh: k_flatten_lines without the Euler evaluation -- results are wrong)
            } else if (true) {
                lp1 = v2(u2f(r0.x) + (float)i, u2f(r0.w) + u2f(r1.x) + u2f(r1.y) + u2f(r1.z) + u2f(r1.w) + u2f(r2.x));
#endif
            } else {  // flatten.wgsl:404-461
                const EulerParams ep = es_params_from_angles(u2f(r1.x), u2f(r1.y));
                const float noff = u2f(r2.x), n = (float)n_u;
                const float tt = (float)(i + 1u) / n;
                float sarg = tt;
                const uint32_t robust = flags & 3u;
                if (robust != 1u) {
                    const float k0 = ep.k0 - 0.5f * ep.k1, k1 = ep.k1;
                    const float dist_scaled = noff * ep.ch;
                    const float int0 = u2f(r1.z), integral = u2f(r1.w);  // as k_flatten_items computed them
                    float a, b;
                    if (robust == 2u) {
                        a = k1;
                        b = k0;
                    } else {
                        a = -2.0f * dist_scaled * k1;
                        b = -1.0f - 2.0f * dist_scaled * k0;
                    }
                    float uu = integral * tt + int0;
                    float inv;
                    if (robust == 2u) inv = pow23_abs_(uu) * sign_(uu); else inv = espc_int_inv_approx(uu);
                    sarg = (inv - b) / a;
                }
                lp1 = es_seg_eval_with_offset(v2(u2f(r0.x), u2f(r0.y)), v2(u2f(r0.z), u2f(r0.w)), ep, sarg, noff);
            }
            const V2 q = xf_apply(tr, lp1);
            if (slot < n_slots) {
                const uint32_t dst = slot_base + k;
                const bool fwd = (flags & 8u) != 0u;  // offset >= 0: (start, end); else the line runs (end, start)
                const bool has_next = !last_of_item && dst + 1u < lines_lim;
                const uint2 hdr = make_uint2(r2.z, 0u), pt = make_uint2(f2u(q.x), f2u(q.y));
                if (dst < lines_lim) {
                    if (k == 0u) {  // the item's first line: header and start point (the job's start point), once per item
                        uint2* w = (uint2*)lines.p;
                        const V2 qs = xf_apply(tr, v2(u2f(r3.x), u2f(r3.y)));
                        w[(size_t)dst * 3u] = hdr;
                        w[(size_t)dst * 3u + (fwd ? 1u : 2u)] = make_uint2(f2u(qs.x), f2u(qs.y));
                    }
                    // fwd: [p1 of this line | header of the next | p0 of the next]; else p0 of this line, header and p1 of the next
                    widx[0] = dst * 3u + (fwd ? 2u : 1u); wdat[0] = pt;
                    if (has_next) {
                        widx[1] = dst * 3u + 3u; wdat[1] = hdr;
                        widx[2] = dst * 3u + (fwd ? 4u : 5u); wdat[2] = pt;
                    }
                }
            }
        }
        {   // (FL_LSPLIT == 2, measurement builds only: the words are not stored -- results are wrong)
            uint2* w = (uint2*)lines.p;
#pragma unroll
            for (int j = 0; j < 3; j++)
#if defined(FL_LSPLIT) && FL_LSPLIT == 2
                if (widx[j] == 0xfffffffeu) w[widx[j]] = wdat[j];
#else
                if (widx[j] != 0xffffffffu) w[widx[j]] = wdat[j];
#endif
        }
    }
}

// order-preserving float <-> uint key (integer LDS atomic min/max on floats)
JD uint32_t fkey(float f) { uint32_t b = f2u(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
JD float fkey_inv(uint32_t k) { return u2f((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// Path bounding boxes (flatten.wgsl:807, :893-899).  The WGSL keeps one box per invocation (tag byte), grown by every
// line it emits from (1e31, 1e31, -1e31, -1e31), and merges floor/ceil of it into the path's box only if it has an
// extent.  floor, ceil and the saturating conversion are monotone, so the merged result is the min/max over the tag's
// lines of their own integer boxes; and a line that has an extent itself proves that its tag's box has one.  The
// kernel therefore streams the lines in their final order, a wave per 1...8 x 64 consecutive lines, and folds
// them by path index (non-decreasing along the lines) with a segmented DPP scan.  Only a line WITHOUT extent (both
// end points equal: none in ordinary scenes) has to look at its tag: it finds the tag's line range in `bases` (a tag's
// three work items are adjacent in the canonical order, so its lines are [bases[3g], bases[3g+3])) and folds that
// range as the WGSL does.  A path folded by one wave is merged with a plain read-modify-write; only a path that
// crosses a range boundary by 64 lines or more is shared between waves and merged with integer atomics (order-free).
JD bool fb_tag_has_extent(const uint32_t* __restrict__ bases, uint32_t n_tags, const JlLineSoup* lines, uint32_t total, uint32_t pos) {
    uint32_t g = 0u, ge = n_tags;  // the last tag whose first line is at or before pos
    while (ge - g > 1u) {
        const uint32_t mid = g + (ge - g) / 2u;
        if (umin_(bases[3u * mid], total) <= pos) g = mid; else ge = mid;
    }
    const uint32_t lo = umin_(bases[3u * g], total);
    const uint32_t hi = (g + 1u < n_tags) ? umin_(bases[3u * g + 3u], total) : total;
    float bx0 = 1e31f, by0 = 1e31f, bx1 = -1e31f, by1 = -1e31f;
    for (uint32_t i = lo; i < hi; i++) {
        const JlLineSoup l = lines[i];
        bx0 = fmin_(bx0, fmin_(l.p0[0], l.p1[0])); by0 = fmin_(by0, fmin_(l.p0[1], l.p1[1]));
        bx1 = fmax_(bx1, fmax_(l.p0[0], l.p1[0])); by1 = fmax_(by1, fmax_(l.p0[1], l.p1[1]));
    }
    return bx1 > bx0 || by1 > by0;
}

JD void fb_merge(Buf<JlPathBbox> path_bboxes, uint32_t path_ix, int32_t x0, int32_t y0, int32_t x1, int32_t y1, bool atomic) {
    if (!path_bboxes.ok(path_ix)) return;
    if (x0 == 0x7fffffff && y0 == 0x7fffffff && x1 == (int32_t)0x80000000 && y1 == (int32_t)0x80000000) return;  // nothing merged
    JlPathBbox* out = &path_bboxes.p[path_ix];
    if (!atomic) {
        out->x0 = imin_(out->x0, x0); out->y0 = imin_(out->y0, y0);
        out->x1 = imax_(out->x1, x1); out->y1 = imax_(out->y1, y1);
    } else {
        // a path spread over many waves (one outline of 200 k segments): the box only grows, so a wave whose box is
        // already inside what it reads (possibly stale, i.e. smaller) has nothing to add
        const int32_t cx0 = __hip_atomic_load(&out->x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int32_t cy0 = __hip_atomic_load(&out->y0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int32_t cx1 = __hip_atomic_load(&out->x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int32_t cy1 = __hip_atomic_load(&out->y1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x0 < cx0) atomicMin(&out->x0, x0);
        if (y0 < cy0) atomicMin(&out->y0, y0);
        if (x1 > cx1) atomicMax(&out->x1, x1);
        if (y1 > cy1) atomicMax(&out->y1, y1);
    }
}

#define FB_NONE 0xffffffffu
#ifndef FB_TARGET_WAVES
#define FB_TARGET_WAVES 2048u  // ranges grow from one batch to eight once the scene has more batches than this
#endif
__global__ __launch_bounds__(JL_WG) void k_flatten_bbox(const JlConfig* __restrict__ cfg, const JlBump* __restrict__ bump,
                                                        const uint32_t* __restrict__ bases, uint32_t n_slots, Buf<JlLineSoup> lines,
                                                        Buf<JlPathBbox> path_bboxes, uint32_t* __restrict__ zero, uint32_t zero_n) {
    // (the stage's counters, which no kernel needs any more, go back to zero for the next frame: kcommon.h, JH_CLEAN_*)
    for (uint32_t i = blockIdx.x * JL_WG + threadIdx.x; i < zero_n; i += gridDim.x * JL_WG) zero[i] = 0u;
    const uint32_t n_tags = n_slots / 3u;
    const uint32_t total = umin_(umin_(bump->lines, cfg->lines_size), lines.n);
    const uint32_t lane = lane_id();
    const uint32_t n_waves = (gridDim.x * JL_WG) >> 6;
    // The host sizes the grid for the buffer's capacity (capped); the line count is only known here.  A small scene is
    // folded a batch per wave (a handful of waves running eight batches one after the other took 40 us longer), a
    // large one in ranges of up to eight batches; a wave takes every n_waves-th range if the grid is too small.
    const uint32_t batches = umin_(umax_(((total + 63u) / 64u) / FB_TARGET_WAVES, 1u), 8u);
    for (uint64_t start64 = (uint64_t)((blockIdx.x * JL_WG + threadIdx.x) >> 6) * batches * 64u; start64 < total;
         start64 += (uint64_t)n_waves * batches * 64u) {  // uniform per wave
        const uint32_t start = (uint32_t)start64;
        const uint32_t end = (uint32_t)(start64 + (uint64_t)batches * 64u < total ? start64 + (uint64_t)batches * 64u : total);
        // Who folds a path that crosses a range boundary?  If it ends within the 64 lines behind the boundary, the wave
        // of the earlier range takes those lines too (one more batch, `ext`) and the later one skips them: the path
        // has one owner and is merged without atomics.  Only a path that runs on for 64 lines or more is shared, and
        // then both sides see that from the same 64 lines and use atomics for it.
        uint32_t p_before = FB_NONE;  // the path of the line before the range ...
        if (start > 0u) p_before = lines.p[start - 1u].path_ix;
        bool skip_before = false;     // ... whose lines at the head of the range belong to the previous wave
        uint32_t p_shared = FB_NONE;  // or are shared with it
        // the segment that is still open at the end of a batch travels on in uniform registers
        uint32_t c_path = FB_NONE;
        int32_t cx0 = 0x7fffffff, cy0 = 0x7fffffff, cx1 = (int32_t)0x80000000, cy1 = (int32_t)0x80000000;
        JlLineSoup l_next = {};
        if (start + lane < total) l_next = lines.p[start + lane];
        for (uint32_t base = start; base < total; base += 64u) {
            const bool ext = base >= end;  // the batch behind the range
            if (ext && c_path == FB_NONE) break;
            const uint32_t pos = base + lane;
            const JlLineSoup l = l_next;
            if (!ext && pos + 64u < total) l_next = lines.p[pos + 64u];  // the next batch's lines travel while this one is folded
            bool valid = pos < (ext ? total : end);
            uint32_t pix = valid ? l.path_ix : FB_NONE;
            const uint32_t p63 = (uint32_t)__builtin_amdgcn_readlane((int)pix, 63);
            if (base == start && p_before != FB_NONE) {
                // (in a range shorter than a batch lane 63 is invalid: the previous wave saw the same in its `ext` batch)
                if (p63 == p_before) p_shared = p_before; else skip_before = true;
            }
            if (ext) {
                if (p63 == c_path) {  // the open path runs on: shared
                    if (lane == 0u) fb_merge(path_bboxes, c_path, cx0, cy0, cx1, cy1, true);
                    c_path = FB_NONE;
                    break;
                }
                valid = valid && pix == c_path;  // only its remaining lines (a prefix of the batch)
                pix = valid ? pix : FB_NONE;
            }
            int32_t x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = (int32_t)0x80000000, y1 = (int32_t)0x80000000;
            const bool skipped = skip_before && base == start && pix == p_before;
            if (valid && !skipped) {
                const float lx0 = fmin_(1e31f, fmin_(l.p0[0], l.p1[0])), ly0 = fmin_(1e31f, fmin_(l.p0[1], l.p1[1]));
                const float lx1 = fmax_(-1e31f, fmax_(l.p0[0], l.p1[0])), ly1 = fmax_(-1e31f, fmax_(l.p0[1], l.p1[1]));
                bool counts = lx1 > lx0 || ly1 > ly0;
                if (!counts) counts = fb_tag_has_extent(bases, n_tags, lines.p, total, pos);
                if (counts) { x0 = to_i32(floor_(lx0)); y0 = to_i32(floor_(ly0)); x1 = to_i32(ceil_(lx1)); y1 = to_i32(ceil_(ly1)); }
            }
            // a carried segment that does not continue in this batch is complete
            const uint32_t p_lane0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pix);
            if (c_path != FB_NONE && c_path != p_lane0) {
                if (lane == 0u) fb_merge(path_bboxes, c_path, cx0, cy0, cx1, cy1, c_path == p_shared);
                c_path = FB_NONE;
            }
#define FB_SEG_STEP(CTRL, ROWS)                                                                                        \
    {                                                                                                                  \
        const bool take = (uint32_t)__builtin_amdgcn_update_dpp((int)~pix, (int)pix, CTRL, ROWS, 0xf, false) == pix;   \
        const int32_t a = imin_(x0, __builtin_amdgcn_update_dpp(x0, x0, CTRL, ROWS, 0xf, false));                      \
        const int32_t b = imin_(y0, __builtin_amdgcn_update_dpp(y0, y0, CTRL, ROWS, 0xf, false));                      \
        const int32_t c = imax_(x1, __builtin_amdgcn_update_dpp(x1, x1, CTRL, ROWS, 0xf, false));                      \
        const int32_t d = imax_(y1, __builtin_amdgcn_update_dpp(y1, y1, CTRL, ROWS, 0xf, false));                      \
        if (take) { x0 = a; y0 = b; x1 = c; y1 = d; }                                                                   \
    }
            FB_SEG_STEP(JK_DPP_ROW_SHR(1), 0xf)
            FB_SEG_STEP(JK_DPP_ROW_SHR(2), 0xf)
            FB_SEG_STEP(JK_DPP_ROW_SHR(4), 0xf)
            FB_SEG_STEP(JK_DPP_ROW_SHR(8), 0xf)
            FB_SEG_STEP(JK_DPP_ROW_BCAST15, 0xa)
            FB_SEG_STEP(JK_DPP_ROW_BCAST31, 0xc)
#undef FB_SEG_STEP
            if (pix == c_path) {  // (only the values of the segment's last lane are used)
                x0 = imin_(x0, cx0); y0 = imin_(y0, cy0); x1 = imax_(x1, cx1); y1 = imax_(y1, cy1);
            }
            // lane i reads the path of lane i + 1 (wave_shl:1); lane 63 has no source: its segment stays open
            const uint32_t pix_next = (uint32_t)__builtin_amdgcn_update_dpp((int)pix, (int)pix, 0x130, 0xf, 0xf, false);
            if (valid && !skipped && pix_next != pix) fb_merge(path_bboxes, pix, x0, y0, x1, y1, pix == p_shared);
            c_path = (uint32_t)__builtin_amdgcn_readlane((int)pix, 63);
            cx0 = __builtin_amdgcn_readlane(x0, 63); cy0 = __builtin_amdgcn_readlane(y0, 63);
            cx1 = __builtin_amdgcn_readlane(x1, 63); cy1 = __builtin_amdgcn_readlane(y1, 63);
            if (ext) break;
        }
        // still open: the lines ended with it (a shared path if it also began before the range)
        if (c_path != FB_NONE && lane == 0u) fb_merge(path_bboxes, c_path, cx0, cy0, cx1, cy1, c_path == p_shared);
    }
}

}  // namespace

#ifdef FL_FAST_CHECK
// (check build only: tools/soak_flatten_fast.py) out[0..3] = nodes tested by the fast path, undecided, contradictions with the
// pinned sequence, violations of the proven bound -- accumulated since the last reset
extern "C" int jh_debug_flatten_fast_stats(uint32_t* out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ff_stats), 8 * sizeof(uint32_t)) != hipSuccess) return -1;
    if (reset) {
        const uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ff_stats), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// [config, scene, tag_monoids, path_bboxes, bump, lines]
int jh_launch_flatten(const JhLaunch& L) {
    if (L.nb < 6) return -1;
    if (L.gx == 0) return 0;
    uint32_t n_tags = L.gx * JL_WG;
    uint64_t n_slots64 = (uint64_t)n_tags * 3;
    if (n_slots64 > 0xfffffff0ull) return -1;
    uint32_t n_slots = (uint32_t)n_slots64;
    auto cfg = (const JlConfig*)L.b[0].ptr;
    auto scene = mkbuf<uint32_t>(L.b[1].ptr, L.b[1].size);
    auto tm = mkbuf<JlTagMonoid>(L.b[2].ptr, L.b[2].size);
    auto pb = mkbuf<JlPathBbox>(L.b[3].ptr, L.b[3].size);
    JlBump* bump = (JlBump*)L.b[4].ptr;
    auto lines = mkbuf<JlLineSoup>(L.b[5].ptr, L.b[5].size);
    uint32_t cap_blocks = (uint32_t)(L.num_cus > 0 ? L.num_cus : 256) * FL_BLOCKS_PER_CU;
    uint32_t g = (n_slots + JL_WG - 1) / JL_WG;
    if (g > cap_blocks) g = cap_blocks;
    // The temporary (FlTemp): a slot per line and at most a record per line, K regions of R each; K * R = the line buffer's
    // capacity + what the regions' ends can waste -- a frame that overflows the temporary has overflowed `lines` and fails as
    // the reference's.  (At least 4096 lines: a frame that overflows a tiny line buffer by less still has every line of the
    // buffer's range written, as the reference's `line_ix < lines_size` guard leaves them -- kat_words.json: lines_overflow_guard)
    const uint64_t line_cap = std::min<uint64_t>(std::max<uint64_t>(lines.n, 4096), 0xf0000000ull);
    FlTemp T;
    T.K = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(line_cap >> 19, 1), FL_MAX_REGIONS);  // (C3: 6 regions, ~3 500 atomics per cursor and frame)
    if ((L.debug_flatten & 2u) != 0u) T.K = FL_MAX_REGIONS;
#ifdef FL_SOAK_HOME0
    T.K = FL_MAX_REGIONS;
#endif
    T.R = (uint32_t)((line_cap + T.K - 1) / T.K) + FL_MAX_GRAB;
    const uint64_t tcap = (uint64_t)T.K * T.R;
    if (tcap > 0xfffffff0ull) return -1;
    // (scratch slots are shared with the later stages: the two large arrays sit where path_count keeps its largest ones)
    uint32_t* counts = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_A, (uint64_t)n_slots * 4);
    uint32_t* bases = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_B, (uint64_t)n_slots * 4);
    uint32_t* list = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_F, (uint64_t)n_slots * 4);
    T.sinfo = (uint2*)jh_scratch_get(L.scratch, JH_SCR_C, tcap * sizeof(uint2));
    T.recs = (uint4*)jh_scratch_get(L.scratch, JH_SCR_D, tcap * 64);
    uint32_t* counters = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_FL_CTR, FL_CTR_WORDS * 4);
    if (!counts || !bases || !list || !counters || !T.sinfo || !T.recs) return -5;
    T.ctr = counters;
    uint32_t* clean = jh_scratch_flags(L.scratch);
    if ((*clean & JH_CLEAN_FL_CTR) == 0u) (void)hipMemsetAsync(counters, 0, FL_CTR_WORDS * 4, L.stream);
    *clean &= ~(uint32_t)JH_CLEAN_FL_CTR;
    {
        const uint32_t n_words = (n_tags + 3u) / 4u;
        uint32_t n_blocks = (n_words + JL_WG - 1u) / JL_WG;  // blocks of 256 tag words
        auto red = mkbuf<JlTagMonoid>(nullptr, 0);
        int scan = 0;
        if ((L.absorb & JH_ABSORB_PATHTAG_SCAN) != 0u) {
            // the held-back last pathtag scan rides in the classification (kcommon.h: extra = its `reduced`, width = its workgroups)
            red = mkbuf<JlTagMonoid>(L.extra.ptr, L.extra.size);
            n_blocks = L.extra.width;
            scan = L.extra.height != 0u ? 1 : 2;
        }
        const dim3 gsc((n_blocks + PSC_BLOCKS - 1u) / PSC_BLOCKS);
        if (scan == 1)
            hipLaunchKernelGGL(k_flatten_classify<1>, gsc, dim3(JL_WG), 0, L.stream, cfg, scene, red, tm, n_blocks, pb, list, counters, n_slots, n_tags,
                               counts, L.absorb, (uint32_t*)bump);
        else if (scan == 2)
            hipLaunchKernelGGL(k_flatten_classify<2>, gsc, dim3(JL_WG), 0, L.stream, cfg, scene, red, tm, n_blocks, pb, list, counters, n_slots, n_tags,
                               counts, L.absorb, (uint32_t*)bump);
        else
            hipLaunchKernelGGL(k_flatten_classify<0>, gsc, dim3(JL_WG), 0, L.stream, cfg, scene, red, tm, n_blocks, pb, list, counters, n_slots, n_tags,
                               counts, L.absorb, (uint32_t*)bump);
    }
    hipLaunchKernelGGL(k_flatten_items, dim3(g), dim3(JL_WG), 0, L.stream, cfg, scene, tm, pb, (const uint32_t*)list, counters, n_slots, counts,
                       T, L.debug_flatten);
    int rc = jh_scan_u32(L, counts, 1, bases, n_slots, nullptr, &bump->lines);
    if (rc) return rc;
    uint32_t gp = (uint32_t)((tcap + JL_WG - 1) / JL_WG);
    uint32_t gp_cap = (uint32_t)(L.num_cus > 0 ? L.num_cus : 256) * 8u;
    // k_flatten_lines strides over its work units: exactly the workgroups that are resident together (more would run
    // as a second, partly filled round behind the first)
    uint32_t gl_cap = (uint32_t)(L.num_cus > 0 ? L.num_cus : 256) * FL_LINES_WAVES_PER_EU;
    if (gp > gl_cap) gp = gl_cap;
    hipLaunchKernelGGL(k_flatten_lines, dim3(gp), dim3(JL_WG), 0, L.stream, cfg, scene, (const uint32_t*)counters, T, (const uint32_t*)bases, n_slots, lines);
    // a wave per 64 lines of the buffer's capacity (the line count is only known on the device), at most 16 waves
    // per SIMD: the kernel lengthens the ranges to match
    uint64_t gb64 = ((uint64_t)lines.n + 255u) / 256u;  // four waves per workgroup
#ifdef FB_MAX_BLOCKS  // (tools/soak_flatten_fallback.sh: a tiny grid, so that every wave strides over many ranges)
    const uint32_t gb_max = FB_MAX_BLOCKS;
#else
    const uint32_t gb_max = 4u * gp_cap;
#endif
    uint32_t gb = gb64 > gb_max ? gb_max : (uint32_t)(gb64 < 1u ? 1u : gb64);
    hipLaunchKernelGGL(k_flatten_bbox, dim3(gb), dim3(JL_WG), 0, L.stream, cfg, (const JlBump*)bump, (const uint32_t*)bases, n_slots, lines, pb,
                       counters, FL_CTR_WORDS);
    *clean |= JH_CLEAN_FL_CTR;
    return 0;
}
