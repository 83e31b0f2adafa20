// kernels_fine.hip -- K19 fine_area (orig/fine.wgsl:824-878 fill_path, :883-1103 main,
// shared/blend.wgsl): per 16x16 tile, interpret the PTCL, accumulate analytic-area coverage from
// the tile's segments, composite colours / gradients / images / clip-blend groups, and store
// un-premultiplied RGBA16F.
//
// MI355X design (round 3): one wave64 per tile (64 lanes x 4 horizontally adjacent pixels, exactly the WGSL's (4,16)
// workgroup), ONE tile-wave per workgroup (the CU takes 25 of them for the lean instantiation -- LDS comes in blocks of 1 280 bytes:
// five per tile-wave, FillLdsT below --; two or four waves per workgroup measured 4 % / 6 % slower).  The PTCL stream and the segment records are the same for all 64 lanes, so each
// datum is fetched ONCE per tile with wide coalesced loads (the algorithmic-bytes model of the roofline) and shared on-chip:
//   * PTCL: a REGISTER window -- lane k of a VGPR holds word k of the stream (one 256-byte load per 64 words, the next
//     window requested one ahead, re-based with ds_bpermute); a command's words are read off the lanes with v_readlane
//     (control words) or taken through the LDS crossbar (the colour of a FILL + COLOR pair);
//   * segments: coarse allocates a tile's segment slices back to back; they are evaluated in batches of up to 63
//     (segment,row) pairs by the wave-level pipeline described above fill_path below; the next batch's segment window is
//     requested straight into LDS (global_load_lds) while the current batch is evaluated, and EVERYTHING a batch leaves
//     behind lives in LDS (struct FillLds), so the command loop carries no per-lane batch state around its back edge.
// Pixels leave as two 16-byte stores per lane (4 px x RGBA16F = 32 B; 4 lanes cover one 128-B row).
// Clip / blend stack: level 0 in wave-private LDS, levels 1-3 in a per-tile slice of a global scratch array, deeper levels in
// blend_spill exactly like the WGSL; layers are lazy (see pushed_depth).  Instantiations: coverage mode (area / msaa8 /
// msaa16) x with/without the clip stack x with/without gradient+image code; the launcher picks by ConfigUniform.n_clip
// and by whether any ramp/image is bound.  Lean area variant: 72 VGPRs, no scratch, 6.1 KB of LDS per tile-wave.
// Command loop: loops per PATTERN in front of a general decoder.  FILL followed by COLOR (the pair a plain scene consists of) runs in a
// loop of its own, and so do the empty layers of clip scenes (BEGIN_CLIP ... [SOLID] END_CLIP closed by the shortcut): one definition of the
// sixteen colour registers around one back edge each.  As arms of one decoder loop with many exits they carried its flag variables, state
// copies and branch chain (C3: 0.402 -> 0.369 ms; C4: 31 k -> 19 k vector instructions per tile).
// What bounds it (profiles/r03_fine_experiments.md, r03_fine_split.json, r03_ubench_issue_rates.txt): vector issue.  A C3 tile is 2710 VALU +
// 1550 SALU + 300 LDS wave-instructions; priced with the measured issue costs (3.8 cycles per vector instruction of this mix, 4.1 per scalar
// one) the vector pipe of a SIMD is busy 74 % of the kernel's time, the scalar pipe 46 %, the LDS ~40 %.
#include <cstring>

#include "kcommon.h"
#include <algorithm>
#include "srgb_lut.h"

using namespace jk;
using namespace jd;

namespace {

#ifndef FINE_BLEND_UNIFORM_DISPATCH
#define FINE_BLEND_UNIFORM_DISPATCH 1
#endif

struct V4 {
    float x, y, z, w;
};
struct V3 {
    float x, y, z;
};
JD V4 v4(float x, float y, float z, float w) { V4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
JD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }

// ---- shared/blend.wgsl ----
JD V3 screen(V3 cb, V3 cs) { return v3(cb.x + cs.x - (cb.x * cs.x), cb.y + cs.y - (cb.y * cs.y), cb.z + cs.z - (cb.z * cs.z)); }
JD float color_dodge(float cb, float cs) {
    if (cb == 0.0f) return 0.0f; else if (cs == 1.0f) return 1.0f; else return fmin_(1.0f, cb / (1.0f - cs));
}
JD float color_burn(float cb, float cs) {
    if (cb == 1.0f) return 1.0f; else if (cs == 0.0f) return 0.0f; else return 1.0f - fmin_(1.0f, (1.0f - cb) / cs);
}
JD float hard_light1(float cb, float cs) {
    float scr_cs = 2.0f * cs - 1.0f;
    float a = cb + scr_cs - (cb * scr_cs);
    float b = cb * 2.0f * cs;
    return (cs <= 0.5f) ? b : a;
}
JD V3 hard_light(V3 cb, V3 cs) { return v3(hard_light1(cb.x, cs.x), hard_light1(cb.y, cs.y), hard_light1(cb.z, cs.z)); }
JD float soft_light1(float cb, float cs) {
    float d = (cb <= 0.25f) ? (((16.0f * cb - 12.0f) * cb + 4.0f) * cb) : sqrt_(cb);
    float t = cb + (2.0f * cs - 1.0f) * (d - cb);
    float f = cb - (1.0f - 2.0f * cs) * cb * (1.0f - cb);
    return (cs <= 0.5f) ? f : t;
}
JD V3 soft_light(V3 cb, V3 cs) { return v3(soft_light1(cb.x, cs.x), soft_light1(cb.y, cs.y), soft_light1(cb.z, cs.z)); }
JD float sat(V3 c) { return fmax_(c.x, fmax_(c.y, c.z)) - fmin_(c.x, fmin_(c.y, c.z)); }
JD float lum(V3 c) { return c.x * 0.3f + c.y * 0.59f + c.z * 0.11f; }
JD V3 clip_color(V3 c) {
    float l = lum(c);
    float n = fmin_(c.x, fmin_(c.y, c.z));
    float x = fmax_(c.x, fmax_(c.y, c.z));
    if (n < 0.0f) c = v3(l + (((c.x - l) * l) / (l - n)), l + (((c.y - l) * l) / (l - n)), l + (((c.z - l) * l) / (l - n)));
    if (x > 1.0f) c = v3(l + (((c.x - l) * (1.0f - l)) / (x - l)), l + (((c.y - l) * (1.0f - l)) / (x - l)), l + (((c.z - l) * (1.0f - l)) / (x - l)));
    return c;
}
JD V3 set_lum(V3 c, float l) { float d = l - lum(c); return clip_color(v3(c.x + d, c.y + d, c.z + d)); }
JD void set_sat_inner(float& cmin, float& cmid, float& cmax, float s) {
    if (cmax > cmin) { cmid = ((cmid - cmin) * s) / (cmax - cmin); cmax = s; }
    else { cmid = 0.0f; cmax = 0.0f; }
    cmin = 0.0f;
}
JD V3 set_sat(V3 c, float s) {
    float r = c.x, g = c.y, b = c.z;
    if (r <= g) {
        if (g <= b) set_sat_inner(r, g, b, s);
        else { if (r <= b) set_sat_inner(r, b, g, s); else set_sat_inner(b, r, g, s); }
    } else {
        if (r <= b) set_sat_inner(g, r, b, s);
        else { if (g <= b) set_sat_inner(g, b, r, s); else set_sat_inner(b, g, r, s); }
    }
    return v3(r, g, b);
}
JD V3 blend_mix(V3 cb, V3 cs, uint32_t mode) {  // blend.wgsl:142-195
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
JD V4 blend_compose(V3 cb, V3 cs, float ab, float as_, uint32_t mode) {  // blend.wgsl:216-284
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
            return v4(fmin_(1.0f, as_ * cs.x + ab * cb.x), fmin_(1.0f, as_ * cs.y + ab * cb.y), fmin_(1.0f, as_ * cs.z + ab * cb.z),
                      fmin_(1.0f, as_ + ab));
        default: break;
    }
    float as_fa = as_ * fa;
    float ab_fb = ab * fb;
    return v4(as_fa * cs.x + ab_fb * cb.x, as_fa * cs.y + ab_fb * cb.y, as_fa * cs.z + ab_fb * cb.z, fmin_(as_fa + ab_fb, 1.0f));
}
// Not inlined: with the sixteen mix modes and fourteen compose operators expanded for each of a lane's four pixels the
// clip + paint instantiation was 68 KB of code -- more than the 64 KB instruction cache -- and with lazy layers the full
// formula is the rare case.
__device__ __attribute__((noinline)) V4 blend_mix_compose(V4 backdrop, V4 src, uint32_t mode) {  // blend.wgsl:288-310
    const float EPSILON = 1e-15f;
#if FINE_BLEND_UNIFORM_DISPATCH
    // `mode` is a PTCL word: the same in every lane.  As a function argument it arrives in a vector register, and the two switches
    // below became trees of v_cmp / s_and_saveexec / s_cbranch_execz -- ~25 vector + scalar instructions per call in front of the
    // arithmetic (round 6: nested C4 spends 55 % of its fine kernel in here, 85 calls x 4 pixels per tile).  As a scalar the switches
    // are compare-and-branch on the scalar pipe.
    mode = (uint32_t)__builtin_amdgcn_readfirstlane((int)mode);
#endif
    if ((mode & 0x7fffu) == 0u) {
        float k = 1.0f - src.w;
        return v4(backdrop.x * k + src.x, backdrop.y * k + src.y, backdrop.z * k + src.z, backdrop.w * k + src.w);
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
        return v4(mix_(backdrop.x, cs.x, src.w), mix_(backdrop.y, cs.y, src.w), mix_(backdrop.z, cs.z, src.w), src.w + backdrop.w * (1.0f - src.w));
    }
    return blend_compose(cb, cs, backdrop.w, src.w, compose_mode);
}

JD float extend_mode(float t, uint32_t mode) {  // fine.wgsl:800-812
    switch (mode) {
        case 0: return clamp_(t, 0.0f, 1.0f);
        case 1: return fract_(t);
        default: return abs_(t - 2.0f * round_(0.5f * t));
    }
}

// The image array of the binding (the reference binds up to 2048 textures, wgpu.go:278).  Up to FINE_MAX_IMAGES
// descriptors travel in the kernel arguments (scalar registers: no load on the IMAGE path); a larger array is indexed
// through a device table of JhImageDesc built by the dispatcher (jello_hip.cpp), `table` != nullptr then.
#define FINE_MAX_IMAGES JH_FINE_INLINE_IMAGES
struct FineImages {
    const uint8_t* px[FINE_MAX_IMAGES];
    uint32_t w[FINE_MAX_IMAGES], h[FINE_MAX_IMAGES];
    uint32_t srgb_mask;  // bit q: image q is JL_RGBA8_SRGB (texels decode to linear like an rgba8unorm-srgb texture)
    int n;
    const JhImageDesc* table;  // all n descriptors when n > FINE_MAX_IMAGES
};

struct FineCfg {  // ConfigUniform fields of the host shadow, by value (valid = 0: read them from the device copy)
    uint32_t valid, width_in_tiles;
    float base_color[4];
};
// LDS byte addresses as integers (stage 4 of fill_path walks addresses)
#define JK_LDS __attribute__((address_space(3)))
JD uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(JK_LDS const void*)p; }
typedef float jk_v4f __attribute__((ext_vector_type(4)));
typedef float jk_v2f __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
JD float4 lds_ld_f4(uint32_t a) { const jk_v4f v = *(const JK_LDS jk_v4f*)a; return make_float4(v.x, v.y, v.z, v.w); }
JD void lds_st_u16(uint32_t a, uint16_t v) { *(JK_LDS uint16_t*)a = v; }
JD void lds_st_u8(uint32_t a, uint8_t v) { *(JK_LDS uint8_t*)a = v; }
JD float lds_ld_f32(uint32_t a) { return *(const JK_LDS float*)a; }
#else  // (the host pass only parses the kernels)
JD float4 lds_ld_f4(uint32_t) { return make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
JD void lds_st_u16(uint32_t, uint16_t) {}
JD void lds_st_u8(uint32_t, uint8_t) {}
JD float lds_ld_f32(uint32_t) { return 0.0f; }
#endif

// rgba = rgba * (1 - fg.a*area) + fg*area  (fine.wgsl:923-926 and the gradient/image arms)
JD V4 over(V4 bg, V4 fg, float area) {
    V4 fg_i = v4(fg.x * area, fg.y * area, fg.z * area, fg.w * area);
    float k = 1.0f - fg_i.w;
    return v4(bg.x * k + fg_i.x, bg.y * k + fg_i.y, bg.z * k + fg_i.z, bg.w * k + fg_i.w);
}

// ------------------------------------------------------------------------------------------------
// fill_path (fine.wgsl:824-878) as a wave-level pipeline.
//
// A segment clipped to a 16x16 tile is short (C3: 3.5 px, 3 rows); evaluating it for all 256 pixels
// wastes ~97 % of the arithmetic, and the IEEE division of the trapezoid area is 40 % of that.  The
// coverage a segment adds to a pixel of row r is  a*dy (+ y_edge)  where, with the WGSL's own operations,
//   a == 1 exactly  when the pixel lies to the right of the segment's x-span in that row (xmax <= 0),
//   a == +0 exactly when it lies to the left (xmin0 - i >= 1; the constant numerator is exactly 0),
// and only the few pixels the span actually crosses need the full formula.  So, per BATCH of segments
// (as many consecutive segments of the tile's slice as give <= 63 (segment,row) pairs):
//   stage 1  lane = segment : the rows it can cross (conservative superset), sign(dx), y_edge  -> prefix sum
//   stage 2  lane = (segment,row) pair : the WGSL's y-part; the row's 16 pixels are classified ONCE from the x-span as
//            group 0 computes it, with a margin of 1e-3 that covers the per-group roundings (everything uncertain counts
//            as crossing); the pair's 16 contributions (dy right of the span, +-0 left of it) go to its ENTRY, and the
//            entries are SORTED BY PIXEL ROW (segment order inside a row): a per-row bit mask of the batch's pairs, built
//            with LDS atomics (commutative), gives every pair its rank in its row
//   stage 3  lane = crossing pixel : the WGSL's trapezoid formula incl. the division, written into the entry.  The
//            pixel's owner pair = running maximum over marks the pairs leave at their first crossing pixel
//   stage 4  lane = pixel quad of row r : walks the entries of ITS row only, in order, two per trip (EXEC-masked, in
//            assembly): the terms of the WGSL's loop that are not +-0 for this row, in the WGSL's order (f32 addition is
//            not associative; skipped terms are +-0, which cannot change a sum that is never -0).  A segment with a y_edge
//            term (it touches the tile's left edge; uniform) ends a run for all rows: the term is added, the walk goes on.
//            C3: 3.5 trips per fill instead of one per segment (5.2) with all 64 lanes.
// Batches run ahead of the command stream (a tile's segment slices are contiguous), so the per-batch
// stages run on full waves although a single CMD_FILL has ~5 segments.
// ------------------------------------------------------------------------------------------------
#define FB_SPEC 128u
#ifndef FINE_LEAN_WAVES_PER_EU
#define FINE_LEAN_WAVES_PER_EU 7  // (72 registers; 6 208 bytes of LDS per tile-wave allow 26 per CU)
#endif
#ifndef FINE_CLIP_WAVES_PER_EU
#define FINE_CLIP_WAVES_PER_EU 4  // 128 VGPRs (6 spilled in the clip + paint instantiation); LDS (4 KiB of stack + 5.9 KiB per tile-wave = eight blocks of 1 280 bytes) allows 16 waves per CU.
                                  // (3, ~140 VGPRs without spills: C4 fine 1.24 instead of 1.19 ms, nested 3.35 instead of 3.15, once the command loops were split)
#endif
#ifndef FINE_CLIP_MS_WAVES_PER_EU
#define FINE_CLIP_MS_WAVES_PER_EU 3
#endif
#ifndef FINE_WAVES
#define FINE_WAVES 1  // tile-waves per workgroup (round 3, C3: 0.405 ms with 1, 0.421 with 2, 0.428 with 4)
#endif
#define FB_PLANE 65
#ifndef FINE_FINAL_ASM
#define FINE_FINAL_ASM 1  // (C3 fine 351.3 -> 347.0 us on the same box)
#endif
#ifndef FINE_COLOR_BPERM
#define FINE_COLOR_BPERM 1  // (C3 fine 346.0 -> 340.7 / 346.9 -> 343.5 us on one box, two rounds)
#endif
#ifndef FINE_LAYER_FILL
#define FINE_LAYER_FILL 1
#endif
#ifndef FINE_VECTOR_LAYERS
#define FINE_VECTOR_LAYERS 1  // (0: the scalar counting loop of round 4)
#endif
#ifndef FINE_CROSS_INLANE
#define FINE_CROSS_INLANE 1  // (0: every crossing pixel through the lane = crossing pixel passes, as up to round 4: C3 fine 362.6 -> 354.5 us with 1)
#endif
// FINE_SKIP (differential builds, `make VARIANT=... EXTRA=-DFINE_SKIP=n`; results are WRONG, only counters and times of
// such a library are of interest -- tools/fine_split.sh): 1 no crossing-pixel formula (stage 3), 2 no row walk / y_edge
// terms (stage 4), 3 no pair evaluation (stages 2 + 3), 4 no batches at all, 5 no compositing of solid colours,
// 6 the FLOOR build (round 5): the real PTCL, the real segment windows, the real number of pairs and crossing pixels, but of
// the coverage pipeline only the arithmetic the output is made of -- the WGSL's y-part once per (segment,row) pair, its
// trapezoid formula (with the IEEE division) once per crossing pixel, two packed additions per segment of a fill (the WGSL's
// own `area += a * dy`), finalisation, composite, store -- and none of what moves it between lanes (pair -> segment mapping,
// row sort, entries, marks, owner scan, row walk).  What stage 1 and the classification cost is in it: without them the
// number of pairs and crossing pixels is not known.
// 7 (round 6, timing only) stage 4 TRANSPOSED: a lane = (fill slot, pixel row) of the batch -- four slots of equal shares of the batch's
// segments stand in for its fills, the real row masks give the trip counts -- walks its row's entries of its slot with all sixteen
// pixels in registers (four 16-byte reads and eight packed adds per entry), the areas go back through the entry planes, and a FILL
// reads its four with one 16-byte load: what VERDICT r05 asked to be measured instead of estimated.
#ifndef FINE_SKIP
#define FINE_SKIP 0
#endif
// FINE_WHATIF (timing-only variant builds, results WRONG; round 6: what do the LDS bank conflicts cost?): bit 0 the row walk reads
// its entries at conflict-free linear addresses (same trips, same instructions), bit 1 stage 2 writes its entries at [quad][lane]
// instead of [quad][row-sorted position], bit 2 the crossing pixels' single floats go to [quad][lane] as well.
#ifndef FINE_WHATIF
#define FINE_WHATIF 0
#endif
#if FINE_WHATIF & 1
#define FINE_WALK_RD "%[rd]"
#define FINE_WALK_RD_OPERAND , [rd] "v"(lds_addr(&F.ent[0][0]) + lane * 16u)
#else
#define FINE_WALK_RD "%[cur]"
#define FINE_WALK_RD_OPERAND
#endif
#if (FINE_SKIP != 0 || FINE_WHATIF != 0) && !defined(JH_VARIANT_BUILD)
#error "FINE_SKIP changes results: build it as a variant library (make VARIANT=name EXTRA='-DJH_VARIANT_BUILD -DFINE_SKIP=n')"
#endif
#define RK_NONEG 1u
#define RK_RANGE 2u
#define RK_LUM 4u
#define FINE_TRIP_WORDS 13u  // PTCL words one trip of the command loop may consume
// Everything a batch leaves behind for stage 4 lives in LDS, not in registers: the command loop then carries no per-lane
// batch state around its back edge (as registers the nine values cost ~30 moves per command: the compiler keeps a second
// copy of every loop-carried value that a nested loop redefines).
template <bool LEAN> struct FillLdsT;
template <> struct FillLdsT<false> {  // the clip instantiations: the lanes' walk state in LDS (they have no registers to spare)
    alignas(16) float4 pre[64];    // the NEXT window's segments (p0x p0y p1x p1y), written by global_load_lds (no registers)
    alignas(16) float4 ent[4][63]; // the batch's (segment,row) pairs SORTED BY ROW (segment order inside a row): [pixel quad]
                                   // [position] = the pair's a*dy for the quad's 4 pixels of its row (a batch has at most 63 pairs).
                                   // One plane per quad: lane = pair writes and lane = pixel quad reads touch consecutive 16-byte slots
    float pre_ye[64];              // ... and their y_edge
    float edge_y[64];              // window segments: y_edge ...
    int8_t edge_s[64];             // ... and sign(dx)  (read by stage 4 at a uniform index: a broadcast)
    uint64_t rowmask[16];          // per pixel row: bit j = pair j of the batch lies in this row
    uint32_t lanest[64];           // lane = pixel quad: LDS address of my row's first entry (my quad of it) | entries consumed << 16;
                                   // lane = window segment: | first pair of the segment << 24 (segments behind the batch: the number of pairs)
    union {
        uint8_t pairflag[64];        // batch set-up: pair -> (window segment + 1) at the first pair of each segment, else 0
        uint8_t specmark[64];        // stage 3: crossing pixel k of the pass is the first one of pair specmark[k] - 1 (0: of none)
    };
    // 6 080 bytes + the 4 KiB of blend-stack level 0 = 10 176 per tile-wave: SIXTEEN per CU (the four waves per SIMD the registers allow);
    // at 10 752 (float2 edges, word marks, a byte array of first pairs, 65-slot planes, 128 marks) it was fifteen, and the sixteenth
    // is worth 6 % on nested C4 and 3 % on C4 (round 6, a what-if sweep: 2 938 us at 10 320 bytes and above, 2 759 at 9 936).
};

// The instantiations without clip layers keep the walk state of the lanes in registers (lanest / first unused) and mark pairs with
// bytes: 6 208 bytes per tile-wave instead of 6 656.  That is the difference between 24 and 26 tile-waves per CU, and on this part
// the 25th and 26th are worth 6 % of the kernel (round 6, a what-if sweep over the LDS size at 72 registers: 341 us at 6 464 bytes
// and above, 321 at 6 208 and below -- profiles/r06_fine_lds_sweep.txt); the kernel fits 72 registers (seven waves per SIMD) once the
// walk's fixed temporaries sit at v64-v71 instead of v72-v79.
template <> struct FillLdsT<true> {
    alignas(16) float4 pre[64];
    alignas(16) float4 ent[4][FB_PLANE];
    float pre_ye[64];
    float2 edge[64];
    uint64_t rowmask[16];
    uint32_t lanest[1];  // (not used)
    uint8_t first[4];    // (not used)
    union {
        uint8_t pairflag[64];
        uint8_t specmark[FB_SPEC];
    };
};
typedef FillLdsT<false> FillLds;
// The load is unconditional (index clamped) so that it can stay in flight as a prefetch; out-of-range segments are
// zeroed when the registers are consumed (robust-access rule) -- a predicated load would be waited for at once.
JD void load_segraw_clamped(const float* __restrict__ segments, uint32_t segments_n, uint32_t so, float& p0x, float& p0y, float& p1x,
                            float& p1y, float& ye) {
    // segments_n == 0: the launcher passes the (always readable) config buffer instead of a null pointer
    const float2* sp = (const float2*)(segments + (size_t)umin_(so, umax_(segments_n, 1u) - 1u) * 6);
    float2 a = sp[0], b = sp[1], c = sp[2];
    p0x = a.x; p0y = a.y; p1x = b.x; p1y = b.y; ye = c.x;
}

// ------------------------------------------------------------------------------------------------
// Multisampled coverage, 8 or 16 samples per pixel (fine_msaa8 / fine_msaa16; the arithmetic is fine.wgsl:148-711's, the sample
// masks are the LUT of renderer/mask.go:43-105 that the caller binds).
//
// What the reference computes per FILL: every segment walks the pixels it touches (a conservative DDA: count = columns + rows
// spanned - 1); each TOUCHED PIXEL takes a half-plane sample mask from the LUT (slope row, offset column), trimmed at the
// segment's two ends, and adds +-1 to the winding counter of every sample the mask covers; a segment that crosses the top of
// a pixel leaves a +-1 for all pixels to its right (an x-prefix afterwards), one that touches the tile's left edge a +-1 for all
// rows below (a y-prefix).  A pixel's coverage is the number of its samples whose counter differs from zero (non-zero rule) or is
// odd (even-odd).  Everything is integer: every decomposition of the sums gives the same words.
//
// MI355X design (round 6; rounds 2-5 ran a restatement of the WGSL's workgroup program, which is gone).  The WGSL starts from
// scratch for every FILL -- 5 segments and ~30 touched pixels on the headline scene, 13 fills per tile -- so each fill pays a
// count pass on 5 lanes, a prefix sum, a binary search over it in workgroup memory per touched pixel, the DDA set-up with its
// division per touched pixel, a dependent LUT fetch, the atomics, and three barriers: a chain of ~15 dependent round trips on a
// wave that is a tenth full.  Here the touched pixels are produced per BATCH of consecutive segments of the tile's segment
// stream (its slices are contiguous: a batch spans ~8 fills), on full waves, ahead of the command stream:
//   set-up   lane = segment : DDA constants (the division once per SEGMENT), number of touched pixels, the left-edge term;
//            a wave prefix sum cuts the batch at MS_CAP touched pixels
//   pixels   lane = touched pixel, 64 per pass: its segment = running maximum (DPP) over marks the segments leave at their first
//            pixel; the pixel, its trimmed sample mask (the LUT fetches of all passes are in flight together) and its flags are
//            packed into ONE word (8 samples; two for 16) in a list in segment order -- rule-agnostic: the two rules differ in one
//            flag, both are kept
//   per FILL the entries of its segments are a contiguous range of that list: lanes take them 64 at a time and add them to the
//            tile's sample words with LDS atomics (commutative), the segments' left-edge terms likewise; the x-prefix runs in
//            registers (SWAR inside the lane's word, DPP across the four lanes of a pixel row), the y-prefix on the four uniform
//            words; the lane resolves its own four pixels.
// A fill therefore costs: clear, one list read, the atomics, one read-back -- no search, no division, no global access.
// WGSL rules kept: shift amounts modulo 32, saturating float -> int conversions, pixels outside the tile drop their writes
// (their index is formed with the WGSL's own u32 arithmetic first), segments behind the buffer read as zeros -- and a zero
// segment touches the tile corner: it does count.
// ------------------------------------------------------------------------------------------------
// FINE_MS_SKIP (timing-only variant builds, results WRONG): bit 0 no entries are applied at the fills, bit 1 no resolve arithmetic,
// bit 2 no touched-pixel passes in the batch build, bit 3 no clearing of the accumulators
#ifndef FINE_MS_SKIP
#define FINE_MS_SKIP 0
#endif
#if FINE_MS_SKIP != 0 && !defined(JH_VARIANT_BUILD)
#error "FINE_MS_SKIP changes results: build it as a variant library"
#endif
// Touched pixels per batch (a sane segment has at most 31; one with more is walked at the fill: MsState::direct).  192 with 8
// samples: 4 944 bytes of LDS per tile-wave then -- LDS is handed out in blocks of 1 280 bytes on this part (two what-if sweeps found
// the steps, profiles/r06_fine_lds_sweep.txt / r06_fine_clip_lds.txt), so 5 120 is the line between 25 and 32 tile-waves per CU.
#ifdef MS_CAP_OVERRIDE  // (soak builds: a multiple of 64)
#define MS_CAP(SAMPLES) MS_CAP_OVERRIDE
#else
#define MS_CAP(SAMPLES) ((SAMPLES) == 8 ? 192u : 256u)
#endif
// An entry of the list, ONE word: sample mask (8 or 16 bits) | pixel << SAMPLES | flags << (SAMPLES + 8): 21 / 29 bits
#define MS_F_DOWN 1u      // the segment runs downwards as given (sign of its winding contribution)
#define MS_F_BUMP_NZ 2u   // the whole pixel takes the contribution too (left-edge crossing): non-zero rule
#define MS_F_BUMP_EO 4u   // ... even-odd rule (differs at the tile's left edge only)
#define MS_F_CARRY 8u     // crosses the pixel's top: +-1 for the pixels to the right
#define MS_F_LIVE 16u     // the pixel lies inside the tile
#define MS_S_DOWN 1u
#define MS_S_RIGHT 2u     // x does not decrease along the (downward) segment
#define MS_S_TOP_ON_EDGE 4u   // the upper end point has x == 0
#define MS_S_BOT_OFF_EDGE 8u  // the lower end point has x != 0
#define MS_S_BY_RULE 16u      // the first pixel's mask depends on the fill rule (see ms_setup): such a segment is walked at the fill
struct MsSeg {  // what a touched pixel needs of its segment (written by lane = segment, read by lane = touched pixel): 28 bytes
    float a, b;          // z = floor(a * k + b): columns crossed after k steps of the DDA
    int32_t x0i;         // column of the first pixel
    float top_y;         // y of the upper end point
    float lut_row;       // LUT row of the slope, times the row length
    uint32_t bits;       // MS_S_* | trim of the first pixel's mask << 5 | touched pixels << 10
    uint32_t first;      // index of the first touched pixel in the batch's list | the bits the last pixel's mask keeps << 16
};
template <int SAMPLES> struct MsLds {
    union {
        MsSeg seg[64];    // while a batch is built
        float4 pre[64];   // between builds: the next window's end points, in flight (global_load_lds)
    };
    alignas(16) uint32_t samples[SAMPLES == 8 ? 512 : 1024];  // [pixel][word]: SWAR winding counters, four samples per word (non-zero); [pixel]: parity bits (even-odd)
    uint32_t ent[MS_CAP(SAMPLES)];                                  // the batch's touched pixels, in segment order
    uint32_t carry_x[64];  // non-zero: [pixel / 4] a byte per pixel; even-odd: [row] a bit per pixel
    alignas(16) uint32_t carry_y[4];  // non-zero: a byte per row; even-odd: word 0, a bit per row
    uint8_t mark[64];
};
JD uint32_t shl32(uint32_t v, uint32_t s) { return v << (s & 31u); }
JD uint32_t shr32(uint32_t v, uint32_t s) { return v >> (s & 31u); }
JD uint32_t ms_span(float a, float b) { return to_u32(fmax_(ceil_(fmax_(a, b)) - floor_(fmin_(a, b)), 1.0f)); }

// lane = segment (or every lane the same segment: ms_direct): fine.wgsl:180-203 / :237-262.  Returns the number of touched pixels;
// `edge` = the left-edge term: row | 16 if it counts upwards | 32 if there is one.
template <int SAMPLES>
JD uint32_t ms_setup(float x0, float y0, float x1, float y1, MsSeg& K, uint32_t& edge) {
    const float LUT_W = SAMPLES == 8 ? 32.0f : 64.0f, HALF_H = SAMPLES == 8 ? 16.0f : 32.0f;
    uint32_t touched = 0u;
    if (!(y0 == y1 && y0 == floor_(y0))) touched = ms_span(x0, x1) + ms_span(y0, y1) - 1u;  // (a horizontal line on the pixel grid touches nothing)
    float edge_y = 16.0f;
    if (x0 == 0.0f) edge_y = y0;
    else if (x1 == 0.0f) edge_y = y1;
    const uint32_t edge_row = to_u32(ceil_(edge_y));
    edge = edge_row < 16u ? (edge_row | (x1 <= x0 ? 16u : 0u) | 32u) : 0u;
    const bool down = y1 >= y0;
    const float tx = down ? x0 : x1, ty = down ? y0 : y1, bx = down ? x1 : x0, by = down ? y1 : y0;  // top / bottom end
    const float dx = abs_(bx - tx), dy = by - ty;
    const float inv = 1.0f / (dx + dy);
    float a = dx * inv;
    const bool right = bx >= tx;
    const float sgn = right ? 1.0f : -1.0f;
    const float xt = floor_(tx * sgn);
    const float frac = tx * sgn - xt;
    const float row0 = floor_(ty);
    const float b = fmin_((dy * frac + dx * ((row0 + 1.0f) - ty)) * inv, 0.99999994f);
    const uint32_t cols = ms_span(tx, bx) - 1u;
    const uint32_t steps = cols + ms_span(ty, by);
    const float err = floor_(a * ((float)steps - 1.0f) + b) - (float)cols;
    if (err != 0.0f) a -= 2e-7f * sign_(err);
    K.a = a; K.b = b;
    K.x0i = to_i32(xt * sgn + 0.5f * (sgn - 1.0f));
    K.top_y = ty;
    K.lut_row = floor_(fmin_(a * HALF_H, HALF_H - 1.0f)) * LUT_W;
    // The sample masks of the first and the last touched pixel are trimmed at the end points (fine.wgsl:356-365): both depend on the
    // segment alone -- the pixel rows come out of the expressions ms_pixel evaluates at k = 0 and k = touched - 1 -- so they are
    // worked out here, once per segment, not by every touched pixel.
    const uint32_t FULL = SAMPLES == 8 ? 0xffu : 0xffffu;
    const float sf = (float)SAMPLES;
    const uint32_t r0 = (uint32_t)to_i32(row0);
    const int32_t y_head = (int32_t)(r0 - (uint32_t)to_i32(floor_(a * 0.0f + b)));
    const uint32_t trim = umin_(to_u32(round_(sf * (ty - (float)y_head))) & 31u, (uint32_t)SAMPLES);  // (a shift by >= SAMPLES clears the mask; the WGSL's shift is modulo 32)
    const uint32_t kl = touched - 1u;
    const int32_t y_tail = (int32_t)(r0 + kl - (uint32_t)to_i32(floor_(a * (float)kl + b)));
    const uint32_t keep = bx != 0.0f ? (FULL & ~shl32(FULL, to_u32(round_(sf * (by - (float)y_tail))))) : FULL;
    // The head trim applies unless the first pixel is "bumped", and at the tile's left edge the two rules bump differently: non-zero
    // only when the start point is not on a pixel row (fine.wgsl:306-310), even-odd always (:598).  They then differ exactly when
    // the start point lies on the edge AND on a row -- where the trim is 0 samples, a no-op, for every finite segment (the first
    // pixel's row IS the start point's).  A segment for which it is not (infinite coordinates) cannot go into a rule-agnostic
    // list: MS_S_BY_RULE sends it through the walk at the fill (ms_fill's direct route), which knows the rule.
    const bool by_rule = tx == 0.0f && row0 == ty && trim != 0u;
    K.bits = (down ? MS_S_DOWN : 0u) | (right ? MS_S_RIGHT : 0u) | (tx == 0.0f ? MS_S_TOP_ON_EDGE : 0u) | (bx != 0.0f ? MS_S_BOT_OFF_EDGE : 0u) | (by_rule ? MS_S_BY_RULE : 0u) | (trim << 5);
    K.first = keep << 16;
    return touched;
}
// lane = touched pixel k of a segment with `touched` of them (fine.wgsl:264-340): pixel | flags << 8 of its entry (0: outside the
// tile), the LUT index of its sample mask, the bits the trims at the segment's two ends leave of it -- `keep` with the head trim of
// the non-zero rule; `keep_eo` with the even-odd rule's (the same except for MS_S_BY_RULE segments).
template <int SAMPLES>
JD uint32_t ms_pixel(const MsSeg& K, uint32_t k, uint32_t touched, uint32_t& lut_ix, uint32_t& keep, uint32_t& keep_eo) {
    const uint32_t FULL = SAMPLES == 8 ? 0xffu : 0xffffu;
    const float LUT_W = SAMPLES == 8 ? 32.0f : 64.0f;
    const bool right = (K.bits & MS_S_RIGHT) != 0u;
    const float sgn = right ? 1.0f : -1.0f;
    const float zf = K.a * (float)k + K.b;
    const float z = floor_(zf);
    const int32_t x = K.x0i + to_i32(sgn * z);
    const float row0 = floor_(K.top_y);
    const int32_t y = (int32_t)((uint32_t)to_i32(row0) + k - (uint32_t)to_i32(z));
    const float z_before = floor_(K.a * (float)(k - 1u) + K.b);
    bool top, bump_nz, bump_eo;
    if (k == 0u) {
        top = row0 == K.top_y;
        bump_eo = (K.bits & MS_S_TOP_ON_EDGE) != 0u;
        bump_nz = bump_eo && row0 != K.top_y;
    } else {
        top = z == z_before;
        bump_nz = bump_eo = right && !top;
    }
    const uint32_t pix = (uint32_t)y * 16u + (uint32_t)x;
    const bool carry = (uint32_t)x < 15u && (uint32_t)y < 16u && top;
    lut_ix = (right ? (SAMPLES == 8 ? 512u : 2048u) : 0u) + to_u32(K.lut_row + floor_((zf - z) * LUT_W));
    const uint32_t tail = k + 1u == touched ? K.first >> 16 : FULL;
    const uint32_t head = k == 0u ? FULL & (FULL << ((K.bits >> 5) & 31u)) : FULL;
    keep = bump_nz ? tail : tail & head;
    keep_eo = bump_eo ? tail : tail & head;
    if (pix >= 256u) return 0u;  // outside the tile: every write of this pixel is dropped
    const uint32_t flags = ((K.bits & MS_S_DOWN) != 0u ? MS_F_DOWN : 0u) | (bump_nz ? MS_F_BUMP_NZ : 0u) | (bump_eo ? MS_F_BUMP_EO : 0u) | (carry ? MS_F_CARRY : 0u) | MS_F_LIVE;
    return pix | (flags << 8);
}
template <int SAMPLES>
JD uint32_t ms_lut(const uint32_t* __restrict__ lut, uint32_t lut_n, uint32_t ix) {
    if (SAMPLES == 8) {
        const uint32_t w = ix / 4u;
        return shr32(w < lut_n ? lut[w] : 0u, (ix % 4u) * 8u) & 0xffu;
    }
    const uint32_t w = ix / 2u;
    return shr32(w < lut_n ? lut[w] : 0u, (ix % 2u) * 16u) & 0xffffu;
}
// One entry into the tile's accumulators (fine.wgsl:341-383 / :640-675).
template <int SAMPLES>
JD void ms_apply(MsLds<SAMPLES>& T, uint32_t e, bool even_odd) {
    const uint32_t FULL = SAMPLES == 8 ? 0xffu : 0xffffu;
    const uint32_t flags = e >> (SAMPLES + 8);
    if ((flags & MS_F_LIVE) == 0u) return;
    uint32_t mask = e & FULL;
    const uint32_t pix = (e >> SAMPLES) & 0xffu;
    const bool bump = (flags & (even_odd ? MS_F_BUMP_EO : MS_F_BUMP_NZ)) != 0u;
    if (even_odd) {
        if (bump) mask ^= FULL;
        atomicXor(&T.samples[pix], mask);
        if ((flags & MS_F_CARRY) != 0u) atomicXor(&T.carry_x[pix >> 4], 2u << (pix & 15u));
        return;
    }
    const bool down = (flags & MS_F_DOWN) != 0u;
    const uint32_t whole = down ? 0x1010101u : (uint32_t)-0x1010101;
#pragma unroll
    for (uint32_t h = 0u; h < (SAMPLES == 8 ? 1u : 2u); h++) {
        // eight mask bits -> eight bytes of 0 / 1 in two words (samples 0 2 4 6 | 1 3 5 7 ... in the WGSL's own interleaving)
        const uint32_t m8 = (mask >> (8u * h)) & 0xffu;
        const uint32_t ma = m8 ^ (m8 << 7);
        const uint32_t mb = ma ^ (ma << 14);
        const uint32_t e0 = mb & 0x1010101u, e1 = (mb >> 4) & 0x1010101u;
        uint32_t s0 = down ? (uint32_t)(-(int32_t)e0) : e0, s1 = down ? (uint32_t)(-(int32_t)e1) : e1;
        if (bump) { s0 += whole; s1 += whole; }
        uint32_t* w = &T.samples[pix * (SAMPLES == 8 ? 2u : 4u) + 2u * h];
        atomicAdd(w, s0);
        atomicAdd(w + 1, s1);
    }
    if ((flags & MS_F_CARRY) != 0u) {
        const uint32_t to = pix + 1u;
        atomicAdd(&T.carry_x[to >> 2], (down ? 1u : 0xffffffffu) << ((to & 3u) << 3));
    }
}
// The batch state that lives in registers: uniform values + two per-lane words (lane = segment of the batch).
struct MsState {
    uint32_t base, hi;      // the batch covers the segments [base, hi) of the tile's stream
    uint32_t next;          // the window in flight to T.pre starts here (~0: none)
    uint32_t total;         // touched pixels in the list
    bool direct;            // the batch is ONE segment with more than MS_CAP touched pixels: walked at the fill, no list
    uint32_t first;         // per lane: list index of my segment's first touched pixel (lanes behind the batch: total)
    uint32_t edge;          // per lane: my segment's left-edge term (ms_setup)
    uint32_t clean;         // the sample words hold the cleared state of: 0 the non-zero rule, 1 even-odd, 2 neither
};
template <int SAMPLES>
JD void ms_build(MsLds<SAMPLES>& T, MsState& B, uint32_t lane, uint32_t so, const float* __restrict__ segments, uint32_t segments_n,
                 const uint32_t* __restrict__ lut, uint32_t lut_n) {
    const uint32_t FULL = SAMPLES == 8 ? 0xffu : 0xffffu;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): whatever window was in flight has landed in T.pre (it aliases T.seg)
    wave_sync();
    float x0, y0, x1, y1;
    if (B.next == so) {
        const float4 t = lds_ld_f4(lds_addr(&T.pre[lane]));
        x0 = t.x; y0 = t.y; x1 = t.z; y1 = t.w;
    } else {
        const float2* sp = (const float2*)(segments + (size_t)umin_(so + lane, umax_(segments_n, 1u) - 1u) * 6);
        const float2 p = sp[0], q = sp[1];
        x0 = p.x; y0 = p.y; x1 = q.x; y1 = q.y;
    }
    if (!(so + lane < segments_n && so + lane >= so)) { x0 = 0.0f; y0 = 0.0f; x1 = 0.0f; y1 = 0.0f; }  // robust access: zeros
    MsSeg K;
    uint32_t edge;
    const uint32_t touched = ms_setup<SAMPLES>(x0, y0, x1, y1, K, edge);
#ifdef MS_FORCE_DIRECT_ABOVE  // (soak builds, tools/soak_round6_shapes.sh: segments with more touched pixels than this take the walk at the fill, which no sane scene reaches otherwise)
    const bool force_direct = touched > MS_FORCE_DIRECT_ABOVE;
#else
    const bool force_direct = false;
#endif
    const uint32_t capped = ((K.bits & MS_S_BY_RULE) != 0u || force_direct) ? MS_CAP(SAMPLES) + 1u : umin_(touched, MS_CAP(SAMPLES) + 1u);  // (what does not fit the list ends the batch)
    const uint32_t incl = wave_incl_scan_u32(capped);
    const uint64_t fit = __builtin_amdgcn_ballot_w64(incl <= MS_CAP(SAMPLES));  // a prefix of the lanes (incl is monotone)
    const uint32_t n = (uint32_t)__builtin_popcountll(fit);
    B.base = so;
    B.edge = edge;
    B.direct = n == 0u;
    B.hi = so + (n == 0u ? 1u : n);
    B.total = n == 0u ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(n - 1u));
    const uint32_t first = incl - capped;
    B.first = lane < n ? first : B.total;
    wave_sync();
    if (lane < n) { K.bits |= touched << 10; K.first |= first; T.seg[lane] = K; }
    const bool starts = lane < n && touched != 0u;
    // The touched pixels, 64 per pass; all passes' LUT fetches are issued before the first is consumed.
    constexpr uint32_t PASSES = MS_CAP(SAMPLES) / 64u;
    uint32_t word[PASSES];
#pragma unroll
    for (uint32_t p = 0u; p < PASSES; p++) {
        word[p] = 0u;
        if (p * 64u < B.total && !(FINE_MS_SKIP & 4)) {  // uniform
            T.mark[lane] = 0u;
            wave_sync();
            if (starts && first - p * 64u < 64u) T.mark[first - p * 64u] = lane + 1u;
            wave_sync();
            // the segment the pass's first pixel belongs to when it does not start there
            const uint64_t earlier = __builtin_amdgcn_ballot_w64(starts && first < p * 64u);
            const uint32_t carry = earlier != 0ull ? 64u - (uint32_t)__builtin_clzll(earlier) : 0u;
            const uint32_t owner = umax_(wave_incl_max_u32(T.mark[lane]), carry);
            const uint32_t e = p * 64u + lane;
            if (e < B.total) {
                const MsSeg S = T.seg[(owner - 1u) & 63u];
                uint32_t ix, keep, keep_eo;
                const uint32_t part = ms_pixel<SAMPLES>(S, e - (S.first & 0xffffu), S.bits >> 10, ix, keep, keep_eo);
                T.ent[e] = keep | (part << SAMPLES);  // (the LUT mask is ANDed in below, once it has arrived)
                if (SAMPLES == 8) { const uint32_t w = ix / 4u; word[p] = shr32(w < lut_n ? lut[w] : 0u, (ix % 4u) * 8u); }
                else { const uint32_t w = ix / 2u; word[p] = shr32(w < lut_n ? lut[w] : 0u, (ix % 2u) * 16u); }
            }
            wave_sync();  // (the next pass rewrites the marks)
        }
    }
#pragma unroll
    for (uint32_t p = 0u; p < PASSES; p++) {
        if (p * 64u < B.total) {  // uniform
            const uint32_t e = p * 64u + lane;
            if (e < B.total) atomicAnd(&T.ent[e], (word[p] & FULL) | ~FULL);
        }
    }
    wave_sync();
    // request the next window: 16 bytes per lane straight into T.pre (index clamped: robust access); T.seg is dead from here on
    B.next = B.hi;
    {
        const float* gp = segments + (size_t)umin_(B.next + lane, umax_(segments_n, 1u) - 1u) * 6;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (JK_LDS void*)&T.pre[0], 16, 0, 0);
    }
}
// One FILL (fine.wgsl:148-501 / :503-711): leaves the coverage of the lane's four pixels in area[].
template <int SAMPLES>
JD void ms_fill(MsLds<SAMPLES>& T, MsState& B, uint32_t lane, uint32_t size_and_rule, uint32_t seg_data, int32_t backdrop, const float* __restrict__ segments,
                uint32_t segments_n, const uint32_t* __restrict__ lut, uint32_t lut_n, float (&area)[4]) {
    const bool even_odd = (size_and_rule & 1u) != 0u;  // uniform
    const uint32_t FULL = SAMPLES == 8 ? 0xffu : 0xffffu;
    constexpr uint32_t WORDS = SAMPLES == 8 ? 2u : 4u;
    const uint32_t ly = lane >> 2, lx = lane & 3u;
    // The accumulators start from zero: 0x80 per counter byte (non-zero), parity 0 (even-odd).  The sample words -- 2 to 16 KB-writes
    // per fill if cleared wholesale, and a fill touches ~25 of the 256 pixels -- are left clean by the fill before (its own touched
    // pixels reset after the read-back, below) unless that fill had another rule or took more than one piece of a list.  16 samples
    // only (C3: 604.6 -> 591 us); with 8 the wholesale clear is two stores and the reset pass costs more than it saves (445 -> 462 us).
    const uint32_t cleared = even_odd ? 0u : 0x80808080u;
    {
        const uint4 z4 = make_uint4(cleared, cleared, cleared, cleared);
        uint4* s = (uint4*)&T.samples[0];
        if (B.clean != (even_odd ? 1u : 0u) && !(FINE_MS_SKIP & 8)) {  // uniform
            if (even_odd) s[lane] = z4;
            else {
#pragma unroll
                for (uint32_t i = 0u; i < WORDS; i++) s[lane * WORDS + i] = z4;
            }
        }
        T.carry_x[lane] = cleared;
        if (lane < 4u) T.carry_y[lane] = cleared;
    }
    wave_sync();
    uint32_t pieces = 0u, piece_e0 = 0u, piece_e1 = 0u;  // list pieces this fill has taken; the last one's range
    uint32_t sa = seg_data, remaining = size_and_rule >> 1;
    while (remaining != 0u) {  // uniform
        if (sa - B.base >= B.hi - B.base) ms_build<SAMPLES>(T, B, lane, sa, segments, segments_n, lut, lut_n);
        const uint32_t take = umin_(remaining, B.hi - sa);
        const uint32_t r0 = sa - B.base;  // the batch's segments [r0, r0 + take)
        if (lane - r0 < take && (B.edge & 32u) != 0u) {  // the left-edge terms of my segment
            const uint32_t row = B.edge & 15u;
            if (even_odd) atomicXor(&T.carry_y[0], 1u << row);
            else atomicAdd(&T.carry_y[row >> 2], ((B.edge & 16u) != 0u ? 1u : 0xffffffffu) << ((row & 3u) << 3));
        }
        if (B.direct) {
            // A segment with more touched pixels than the list holds, or one whose first mask depends on the rule (coordinates far
            // outside the tile / infinite: never what path_tiling writes) is walked here, every lane with the segment's constants in
            // its own registers.
            float x0 = 0.0f, y0 = 0.0f, x1 = 0.0f, y1 = 0.0f;
            if (sa < segments_n) {
                const float2* sp = (const float2*)(segments + (size_t)sa * 6);
                const float2 p = sp[0], q = sp[1];
                x0 = p.x; y0 = p.y; x1 = q.x; y1 = q.y;
            }
            MsSeg K;
            uint32_t edge_unused;
            const uint32_t touched = ms_setup<SAMPLES>(x0, y0, x1, y1, K, edge_unused);
            for (uint32_t k0 = 0u; k0 < touched; k0 += 64u) {  // uniform
                const uint32_t k = k0 + lane;
                if (k < touched && k >= k0) {
                    uint32_t ix, keep, keep_eo;
                    const uint32_t part = ms_pixel<SAMPLES>(K, k, touched, ix, keep, keep_eo);
                    ms_apply<SAMPLES>(T, (ms_lut<SAMPLES>(lut, lut_n, ix) & (even_odd ? keep_eo : keep)) | (part << SAMPLES), even_odd);
                }
                if (touched - k0 <= 64u) break;  // (k0 + 64 may wrap)
            }
        } else {
            const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)B.first, (int)(r0 & 63u));
            const uint32_t e1 = r0 + take >= 64u ? B.total : (uint32_t)__builtin_amdgcn_readlane((int)B.first, (int)((r0 + take) & 63u));
            for (uint32_t eb = e0; eb < e1 && !(FINE_MS_SKIP & 1); eb += 64u)  // uniform
                if (eb + lane < e1) ms_apply<SAMPLES>(T, T.ent[eb + lane], even_odd);
            piece_e0 = e0; piece_e1 = e1;
        }
        pieces += B.direct ? 2u : 1u;
        sa += take;
        remaining -= take;
    }
    wave_sync();
    // resolve: the lane's own four pixels (fine.wgsl:386-501 / :677-710)
    if (even_odd) {
        uint32_t px = T.carry_x[ly];
        px ^= px << 1; px ^= px << 2; px ^= px << 4; px ^= px << 8;
        uint32_t py = T.carry_y[0];
        py ^= py << 1; py ^= py << 2; py ^= py << 4; py ^= py << 8;
        const uint32_t row_parity = (py >> ly) ^ (uint32_t)backdrop;
        const uint4 s = ((const uint4*)&T.samples[0])[lane];
        const uint32_t sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (uint32_t i = 0u; i < 4u; i++) {
            const uint32_t parity = row_parity ^ (px >> (lx * 4u + i));
            const uint32_t flip = (uint32_t)(-(int32_t)(parity & 1u));
            area[i] = (float)__builtin_popcount((sv[i] ^ flip) & FULL) * (SAMPLES == 8 ? 0.125f : 0.0625f);
        }
        wave_sync();
        B.clean = 2u;
        if (SAMPLES == 16 && pieces <= 1u) {  // uniform: the pixels this fill touched are all in one range of the list that is still there
            for (uint32_t eb = piece_e0; eb < piece_e1; eb += 64u) {
                if (eb + lane < piece_e1) {
                    const uint32_t e = T.ent[eb + lane];
                    if (((e >> (SAMPLES + 8)) & MS_F_LIVE) != 0u) T.samples[(e >> SAMPLES) & 0xffu] = 0u;
                }
            }
            B.clean = 1u;
            wave_sync();
        }
        return;
    }
    // x: inclusive sum of the carries of my four pixels inside the word, then the totals of the lanes to my left in my row
    uint32_t wx = T.carry_x[lane];
    wx += (wx - 0x808080u) << 8;
    wx += (wx - 0x8080u) << 16;
    {
        const uint32_t tot = ((wx >> 24) - 0x80u) * 0x1010101u;
        const uint32_t t1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tot, JK_DPP_ROW_SHR(1), 0xf, 0xf, false);
        const uint32_t t2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tot, JK_DPP_ROW_SHR(2), 0xf, 0xf, false);
        const uint32_t t3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tot, JK_DPP_ROW_SHR(3), 0xf, 0xf, false);
        wx += (lx >= 1u ? t1 : 0u) + (lx >= 2u ? t2 : 0u) + (lx >= 3u ? t3 : 0u);
    }
    // y: the same on the four uniform words; a row's value = its byte of its word's running sum + the totals of the words above
    uint32_t wy;
    {
        const uint4 yv = *(const uint4*)&T.carry_y[0];
        if (uni((yv.x ^ 0x80808080u) | (yv.y ^ 0x80808080u) | (yv.z ^ 0x80808080u) | (yv.w ^ 0x80808080u)) == 0u) {
            wy = 0u;  // (uniform, and the usual case: no segment of the fill touches the tile's left edge)
        } else {
        uint32_t p[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
        for (int i = 0; i < 4; i++) { p[i] += (p[i] - 0x808080u) << 8; p[i] += (p[i] - 0x8080u) << 16; }
        const uint32_t g = ly >> 2;
        const uint32_t mine = g == 0u ? p[0] : (g == 1u ? p[1] : (g == 2u ? p[2] : p[3]));
        wy = (mine >> ((ly & 3u) << 3)) - 0x80u;
        if (g >= 1u) wy += (p[0] >> 24) - 0x80u;
        if (g >= 2u) wy += (p[1] >> 24) - 0x80u;
        if (g >= 3u) wy += (p[2] >> 24) - 0x80u;
        }
    }
    uint32_t sw[4 * WORDS];
    {
        const uint4* s = (const uint4*)&T.samples[0];
#pragma unroll
        for (uint32_t i = 0u; i < WORDS; i++) { const uint4 v = s[lane * WORDS + i]; sw[4 * i] = v.x; sw[4 * i + 1] = v.y; sw[4 * i + 2] = v.z; sw[4 * i + 3] = v.w; }
    }
#pragma unroll
    for (uint32_t i = 0u; i < 4u; i++) {
        // the winding number every sample of the pixel starts from; a sample is covered when its counter differs from "zero"
        const uint32_t zero = (((wx >> (i * 8u)) + wy) & 0xffu) - (uint32_t)backdrop;
        if (FINE_MS_SKIP & 2) {
            area[i] = u2f((sw[i * WORDS] & 0x7fffffu) | 0x3f000000u);
        } else if (zero >= 256u) {
            area[i] = 1.0f;
        } else if (SAMPLES == 8) {
            const uint32_t d0 = (zero * 0x1010101u) ^ sw[i * 2u], d1 = (zero * 0x1010101u) ^ sw[i * 2u + 1u];
            const uint32_t d0_2 = d0 | (d0 * 2u), d1_2 = d1 | (d1 >> 1);
            const uint32_t d2 = (d0_2 & 0xAAAAAAAAu) | (d1_2 & 0x55555555u);
            const uint32_t d4 = d2 | (d2 * 4u);
            const uint32_t d8 = d4 | (d4 * 16u);
            area[i] = (float)__builtin_popcount(d8 & 0xC0C0C0C0u) * 0.125f;
        } else {
            const uint32_t z4 = zero * 0x1010101u;
            const uint32_t d0 = z4 ^ sw[i * 4u], d1 = z4 ^ sw[i * 4u + 1u], d2 = z4 ^ sw[i * 4u + 2u], d3 = z4 ^ sw[i * 4u + 3u];
            const uint32_t d0_2 = d0 | (d0 * 2u), d1_2 = d1 | (d1 >> 1);
            const uint32_t d01 = (d0_2 & 0xAAAAAAAAu) | (d1_2 & 0x55555555u);
            const uint32_t d01_4 = d01 | (d01 * 4u);
            const uint32_t d2_2 = d2 | (d2 * 2u), d3_2 = d3 | (d3 >> 1);
            const uint32_t d23 = (d2_2 & 0xAAAAAAAAu) | (d3_2 & 0x55555555u);
            const uint32_t d23_4 = d23 | (d23 >> 2);
            const uint32_t d4 = (d01_4 & 0xCCCCCCCCu) | (d23_4 & 0x33333333u);
            const uint32_t d8 = d4 | (d4 * 16u);
            area[i] = (float)__builtin_popcount(d8 & 0xF0F0F0F0u) * 0.0625f;
        }
    }
    wave_sync();
    B.clean = 2u;
    if (SAMPLES == 16 && pieces <= 1u) {  // uniform: as above
        for (uint32_t eb = piece_e0; eb < piece_e1; eb += 64u) {
            if (eb + lane < piece_e1) {
                const uint32_t e = T.ent[eb + lane];
                if (((e >> (SAMPLES + 8)) & MS_F_LIVE) != 0u) {
                    const uint32_t pix = (e >> SAMPLES) & 0xffu;
                    if (SAMPLES == 8) *(uint2*)&T.samples[pix * 2u] = make_uint2(0x80808080u, 0x80808080u);
                    else *(uint4*)&T.samples[pix * 4u] = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
                }
            }
        }
        B.clean = 0u;
        wave_sync();
    }
}

template <int AA, bool CLIPS> struct FineLdsSel { typedef MsLds<AA> type; };
template <bool CLIPS> struct FineLdsSel<0, CLIPS> { typedef FillLdsT<!CLIPS> type; };
// The first JL_BLEND_STACK_SPLIT levels of the clip / blend stack (fine.wgsl:938-973 keeps them in registers; deeper
// levels go to blend_spill), one float4 per pixel, lane-contiguous (conflict-free 16-byte accesses), addressed by the
// (uniform) level: levels 0 and 1 in wave-private LDS (8 KiB), levels 2 and 3 in a per-tile slice of a global scratch
// array (with lazy layers only layers that really draw are saved, and few of those nest three deep; all four levels in
// LDS held the clip instantiations at 7 waves per CU).  In registers the four levels needed a four-way switch with
// sixteen moves per case and 64 VGPRs: 219 VGPRs and ~290 VALU instructions per command.
#ifndef FINE_LDS_LEVELS
#define FINE_LDS_LEVELS 1u  // (two levels in LDS: 11 instead of 12 tile-waves per CU; C4 fine 1.50 instead of 1.37 ms)
#endif
#define FINE_SCR_LEVELS (JL_BLEND_STACK_SPLIT - FINE_LDS_LEVELS)  // levels kept in the global scratch array
template <bool CLIPS> struct FineStackSel { struct type { float4 lvl[FINE_LDS_LEVELS][4][64]; }; };
template <> struct FineStackSel<false> { struct type { float4 lvl[1][1][1]; }; };

// Pixel ownership = the WGSL's: lane = ly*4 + lx (workgroup (4,16)), pixel i = 0..3 at column 4*lx + i.
// AA = 0: analytic area coverage (fine_area); 8 / 16: fine_msaa8 / fine_msaa16.
// Tile-waves per workgroup: two where LDS is small (the CU runs at most 16 workgroups, so single-wave workgroups would cap
// the occupancy at 4 waves per SIMD); one for the clip instantiations (a history note of round 2: their LDS is 10 KB per wave now).
#define FINE_WG_WAVES(CLIPS) ((CLIPS) ? 1 : FINE_WAVES)
#ifndef FINE_LEAN_MS_WAVES_PER_EU
#define FINE_LEAN_MS_WAVES_PER_EU 7  // (C3 msaa8: 580 / 522 / 490 us at 4 / 5 / 6 waves per SIMD; 7 -- 72 registers, four spilled -- once the LDS allows 28 tile-waves: 440 -> 429)
#endif
#define FINE_WAVES_PER_EU(AA, CLIPS, PAINTS) \
    ((CLIPS) ? ((AA) != 0 ? FINE_CLIP_MS_WAVES_PER_EU : FINE_CLIP_WAVES_PER_EU) : ((PAINTS) ? 4 : ((AA) == 16 ? 5 : ((AA) != 0 ? FINE_LEAN_MS_WAVES_PER_EU : FINE_LEAN_WAVES_PER_EU))))  /* (16 samples: 7.7 KB of LDS per tile-wave allow 5 per SIMD anyway) */
template <int AA, bool CLIPS, bool PAINTS>
__global__ __launch_bounds__(64 * FINE_WG_WAVES(CLIPS)) __attribute__((amdgpu_waves_per_eu(FINE_WAVES_PER_EU(AA, CLIPS, PAINTS), FINE_WAVES_PER_EU(AA, CLIPS, PAINTS)))) void k_fine_area(const JlConfig* __restrict__ cfg, FineCfg fc, const float* __restrict__ segments, uint32_t segments_n,
                                                  const uint32_t* __restrict__ ptcl, uint32_t ptcl_n, const uint32_t* __restrict__ info,
                                                  uint32_t info_n, Buf<V4> blend_spill, uint16_t* __restrict__ output, uint32_t out_w,
                                                  uint32_t out_h, const uint16_t* __restrict__ gradients, uint32_t grad_h, FineImages images,
                                                  uint32_t tiles_x, const uint32_t* __restrict__ mask_lut, uint32_t mask_lut_n,
                                                  uint32_t tile_row0,  // first tile row of the launch (band mode)
                                                  float4* __restrict__ clip_scratch,  // CLIPS: stack levels behind the LDS one: [tile][scr_levels][4][64]
                                                  uint32_t scr_levels,                 // levels per tile in clip_scratch (0 ... FINE_SCR_LEVELS, from the scene's clip depth)
                                                  uint32_t* __restrict__ hint_overflow) {  // counts the saves dropped for want of a level (a clip-depth hint that was too small)
    const uint32_t tile_y = blockIdx.y + tile_row0;
    // FINE_WAVES independent waves (= tiles, side by side in x) per workgroup: the CU runs at most 16 workgroups, so
    // single-wave workgroups would cap the occupancy at 4 waves per SIMD.  The waves never synchronise with each other.
    constexpr uint32_t WV = FINE_WG_WAVES(CLIPS);
    __shared__ typename FineLdsSel<AA, CLIPS>::type F_all[WV];
    __shared__ typename FineStackSel<CLIPS>::type S_all[WV];
    const uint32_t wave_in_wg = threadIdx.x >> 6;
    auto& F = F_all[wave_in_wg];
    auto& S = S_all[CLIPS ? wave_in_wg : 0u];  // (one dummy element per wave when !CLIPS: 16 bytes)
    const uint32_t tile_x = blockIdx.x * WV + wave_in_wg;
    if (tile_x >= tiles_x) return;  // tiles_x = the dispatch's x size
    if (ptcl_n == 0u) return;
    const uint32_t ptcl_head = ptcl[0];  // ~0: an earlier stage failed (fine.wgsl:889-893); tested once the tile's first loads are under way
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t ly = lane >> 2, lx = lane & 3u;
    // (the two config fields fine needs travel in the kernel arguments when the host has a shadow of the uniform: a wave's
    // first load is then its PTCL window, not a scalar load it has to wait for first)
    const uint32_t tile_ix = tile_y * (fc.valid ? fc.width_in_tiles : cfg->width_in_tiles) + tile_x;
    const uint32_t scratch_tile = blockIdx.y * tiles_x + tile_x;  // this tile's slice of clip_scratch
    const float xyx = (float)((tile_x * 4u + lx) * 4u);  // WGSL xy.x = f32(global_id.x * 4)
    const float xyy = (float)(tile_y * 16u + ly);         // WGSL xy.y
    V4 rgba[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        rgba[k] = fc.valid ? v4(fc.base_color[0], fc.base_color[1], fc.base_color[2], fc.base_color[3])
                           : v4(cfg->base_color[0], cfg->base_color[1], cfg->base_color[2], cfg->base_color[3]);
    uint32_t clip_depth = 0u;
    // Lazy layers.  BEGIN_CLIP saves the colour so far and starts the layer from zero; content such as the C4 scene opens
    // hundreds of layers over a tile of which only a few draw anything there.  So the save is DEFERRED: levels
    // [pushed_depth, clip_depth) are open but not yet on the stack -- rgba still holds the colour from before the
    // outermost of them.  The first composite inside them performs the pending saves (`materialize`: the outermost
    // pending level gets rgba, the ones inside it the zeros they would have started from).  A layer that is closed while
    // still pending is empty: END_CLIP then blends an all-zero source into rgba, which for the common blend modes leaves
    // it bit for bit as it is (shown at END_CLIP below) -- no stack traffic and no blend arithmetic for empty layers.
    uint32_t pushed_depth = 0u;
    float area[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // The command stream is the same for all 64 lanes.  It is kept in a REGISTER window: lane k of `wcur` holds word
    // wbase + k of the stream (one coalesced 256-byte load per 64 words, bounds-checked per lane: robust access), `wnext`
    // the 64 words after that -- requested one window ahead, so its latency passes under the commands in between.  A
    // command's words are read off the lanes with v_readlane (uniform lane index) into scalar registers: no LDS staging,
    // no per-word address arithmetic, and no memory latency between short commands (a scalar load per command had
    // ~1 us of it: fatal for streams of hundreds of one-word clip commands).
    uint32_t pc = uni(tile_ix * JL_PTCL_INITIAL_ALLOC);  // absolute word index of the next command
    MsState msb;  // (AA != 0) the batch of touched pixels: nothing yet
    msb.base = 0u; msb.hi = 0u; msb.next = 0xffffffffu; msb.total = 0u; msb.direct = false; msb.first = 0u; msb.edge = 0u; msb.clean = 2u;
    auto I = [&](uint32_t i) -> uint32_t { return i < info_n ? info[i] : 0u; };
    const uint32_t blend_offset = pc < ptcl_n ? ptcl[pc] : 0u;
    pc += 1u;
    auto load_grad = [&](int32_t x, uint32_t y) -> V4 {
        if (x < 0 || x >= JL_GRADIENT_WIDTH || y >= grad_h) return v4(0, 0, 0, 0);
        const uint16_t* t = gradients + ((size_t)y * JL_GRADIENT_WIDTH + (size_t)x) * 4;
        uint2 raw = *(const uint2*)t;
        return v4(f16_to_f32((uint16_t)(raw.x & 0xffffu)), f16_to_f32((uint16_t)(raw.x >> 16)), f16_to_f32((uint16_t)(raw.y & 0xffffu)),
                  f16_to_f32((uint16_t)(raw.y >> 16)));
    };
    auto pix_i = [&](int k) -> float { return (float)k; };
    auto pix_spill = [&](int k) -> uint32_t { return ly * JL_TILE_WIDTH + lx * 4u + (uint32_t)k; };
    // Segment window and batch state.  All of it is uniform (scalar registers) or in F (LDS).
    uint32_t cur_base = 0u, nxt_base = 0xffffffffu;  // window base; base of the window in flight to F.pre ("none")
    uint32_t batch_hi = 0u;                          // segments [cur_base, batch_hi) are evaluated
    uint32_t n_pairs = 0u;                           // pairs of the batch
    uint64_t edge_mask = 0ull;                       // window segments of the batch with a y_edge term
    uint32_t next_seg = 0xffffffffu;                 // the segment the rows' consumed counts stand in front of
    const float lyf = (float)ly;
    uint32_t ent_lds = 0u;  // LDS byte address of my quad's plane of entries (stage 4 walks addresses)
    // per-lane walk state of the batch, carried in registers from fill to fill where there are registers (!CLIPS; as LDS words they cost
    // every fill a round trip before its walk could start: C3 fine 362 -> 358 us): my row's mask of pairs, the first pair of the window segment in this lane, my row's entries and how many of
    // them the fills so far have consumed
    uint64_t s4_rowmask = 0ull;
    uint32_t s4_first = 0u, s4_row_addr = 0u, s4_done = 0u;
    if constexpr (AA == 0) ent_lds = lds_addr(&F.ent[lx][0]);

    // Evaluate the batch that starts at segment `so` (uniform).
    auto build_batch = [&](uint32_t so) {
      if constexpr (AA == 0) {
        wave_sync();  // stage 4 of the previous batch is done with F
        // The window always starts at the batch's first segment (so every batch can fill its 64 pair slots); it was
        // requested while the previous batch was evaluated (straight into LDS) unless the fills are not contiguous.
        float c_p0x, c_p0y, c_p1x, c_p1y, c_ye;
        if (nxt_base == so) {
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the window has arrived in F.pre
            wave_sync();
            const float4 t = lds_ld_f4(lds_addr(&F.pre[lane]));  // (explicit LDS loads: merged with the other arm they become flat loads)
            c_p0x = t.x; c_p0y = t.y; c_p1x = t.z; c_p1y = t.w; c_ye = lds_ld_f32(lds_addr(&F.pre_ye[lane]));
        } else {
            load_segraw_clamped(segments, segments_n, so + lane, c_p0x, c_p0y, c_p1x, c_p1y, c_ye);
        }
        {
            const bool ok = so + lane < segments_n;  // robust access: out-of-range segments read as zero
            c_p0x = ok ? c_p0x : 0.0f; c_p0y = ok ? c_p0y : 0.0f; c_p1x = ok ? c_p1x : 0.0f; c_p1y = ok ? c_p1y : 0.0f;
            c_ye = ok ? c_ye : 0.0f;
        }
        cur_base = so;
        const float dlx = c_p1x - c_p0x, dly = c_p1y - c_p0y;
        if constexpr (CLIPS) { F.edge_y[lane] = c_ye; F.edge_s[lane] = (int8_t)(dlx > 0.0f ? 1 : (dlx < 0.0f ? -1 : 0)); }
        else F.edge[lane] = make_float2(c_ye, sign_(dlx));
        // stage 1: conservative superset of the rows with dy != 0.  Coordinates are tile relative (|v| <= 16
        // for what path_tiling writes): with |v| <= 64 every rounding error of the WGSL's row arithmetic is
        // < 1e-4, so widening by 1e-3 is safe; anything else takes all 16 rows.
        const bool sane = abs_(c_p0x) <= 64.0f && abs_(c_p0y) <= 64.0f && abs_(c_p1x) <= 64.0f && abs_(c_p1y) <= 64.0f;
        int32_t ra = 0, rb = 16;
        if (sane) {
            ra = iclamp_((int32_t)floor_(fmin_(c_p0y, c_p1y) - 1.0e-3f), 0, 16);
            rb = iclamp_((int32_t)ceil_(fmax_(c_p0y, c_p1y) + 1.0e-3f), 0, 16);
        }
        const uint32_t my_cnt = (uint32_t)imax_(rb - ra, 0);
        const uint32_t incl = wave_incl_scan_u32(my_cnt);
        // (at most 63 segments and 63 pairs per batch: every "pairs / segments in front of P" mask is (1 << P) - 1 with P < 64)
        const uint64_t fit = __builtin_amdgcn_ballot_w64(incl <= 63u) & 0x7fffffffffffffffull;  // a prefix of the lanes (incl is monotone)
        const uint32_t e_rel = (uint32_t)__builtin_popcountll(fit);       // > 0: one segment has at most 16 pairs
        n_pairs = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(e_rel - 1u));
        const uint32_t first = incl - my_cnt;
        if constexpr (!CLIPS) s4_first = lane < e_rel ? first : n_pairs;  // (CLIPS: packed into F.lanest below)
        // y_edge >= 16 (path_tiling's "no edge" value is 1e9) clamps to 0 for every row of the tile
        edge_mask = __builtin_amdgcn_ballot_w64(c_ye < 16.0f) & fit;
        const uint32_t meta = first | ((uint32_t)ra << 8) | (sane ? (1u << 16) : 0u);
        nxt_base = so + e_rel;
        {   // request the next window: 16 + 4 bytes per lane straight into F.pre / F.pre_ye (index clamped: robust access)
            const float* gp = segments + (size_t)umin_(nxt_base + lane, umax_(segments_n, 1u) - 1u) * 6;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (JK_LDS void*)&F.pre[0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + 4), (JK_LDS void*)&F.pre_ye[0], 4, 0, 0);
        }
        F.pairflag[lane] = 0u;
        if (lane < 16u) F.rowmask[lane] = 0ull;
        batch_hi = so + e_rel;
#if FINE_SKIP == 6
        {   // floor build: every pair lane evaluates the y-part and the classification on the values its own lane holds
            next_seg = so;
            const bool is_pair = lane < n_pairs;
            const float y = c_p0y - (float)(ra & 15);
            const float y0 = clamp_(y, 0.0f, 1.0f);
            const float y1 = clamp_(y + dly, 0.0f, 1.0f);
            const float s2_dy = y0 - y1;
            const float vec_y_recip = 1.0f / dly;
            const float s2_tx0 = ((y0 - y) * vec_y_recip) * dlx, s2_tx1 = ((y1 - y) * vec_y_recip) * dlx;
            const float gx0 = c_p0x + s2_tx0, gx1 = c_p0x + s2_tx1;
            const float xmin0 = fmin_(gx0, gx1), xmax0 = fmax_(gx0, gx1);
            const bool guard = sane && abs_(xmin0) <= 40.0f && abs_(xmax0) <= 40.0f;
            const int32_t c1 = guard ? iclamp_((int32_t)ceil_(xmax0 + 1.0e-3f), 0, 16) : 16;
            const int32_t n0 = guard ? iclamp_((int32_t)floor_(xmin0 - 1.0e-3f), 0, 16) : 0;
            const uint32_t ncross = (is_pair && s2_dy != 0.0f) ? (uint32_t)imax_(c1 - n0, 0) : 0u;
            const uint32_t nspec = wave_reduce_u32(ncross);
            float sink = 0.0f;
            for (uint32_t k0 = 0u; k0 < nspec; k0 += 64u) {  // uniform: one formula per 64 crossing pixels
                const uint32_t X = ((uint32_t)n0 + k0) & 15u;
                const float i_f = (float)(X & 3u);
                const float startx = c_p0x - (float)(4u * (X >> 2));
                const float x0 = startx + s2_tx0, x1 = startx + s2_tx1;
                const float xmn0 = fmin_(x0, x1), xmx0 = fmax_(x0, x1);
                float xmin = fmin_(xmn0 - i_f, 1.0f) - 1.0e-6f;
                float xmax = xmx0 - i_f;
                float b = fmin_(xmax, 1.0f);
                float c = fmax_(b, 0.0f);
                float d = fmax_(xmin, 0.0f);
                float a = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
                sink = a * s2_dy;
                asm volatile("" : "+v"(sink));
            }
            asm volatile("" ::"v"(sink));
            return;
        }
#endif
        wave_sync();
        if (lane < e_rel && my_cnt != 0u) F.pairflag[first & 63u] = lane + 1u;
        wave_sync();
        // pair -> segment: running maximum of the start flags; the segment's values come from its lane (ds_bpermute)
        const uint32_t owner = wave_incl_max_u32(F.pairflag[lane]);
        const uint32_t pseg = (owner - 1u) & 63u;
        const uint32_t pmeta = __shfl(meta, (int)pseg, 64);
        const float p0x = __shfl(c_p0x, (int)pseg, 64), p0y = __shfl(c_p0y, (int)pseg, 64);
        const float sdx = __shfl(dlx, (int)pseg, 64), sdy = __shfl(dly, (int)pseg, 64);
        // stage 2, lane = pair: the WGSL's y-part, then a conservative classification of the row's 16 pixels from the
        // x-span as group 0 (startx = p0x) computes it.  The WGSL evaluates x0 / x1 once per group of four pixels as
        // fl(fl(p0x - 4g) + tx); all of these are within 4 roundings at magnitude <= 128 (< 4e-5) of the group-0
        // value minus 4g, so with a margin of 1e-3
        //   pixels X >= ceil(xmax0 + 1e-3)  have fl(xmax0_g - i) <= 0, i.e. a == 1 exactly: they contribute dy,
        //   pixels X <  floor(xmin0 - 1e-3) have fl(xmin0_g - i) >= 1, i.e. a == +0 exactly: they contribute +-0,
        // and everything in between takes the full formula in stage 3 (which reproduces a == 1 / a == +0 by itself
        // where the margin was not needed).
        const bool is_pair = FINE_SKIP == 3 ? false : lane < n_pairs;
        const uint32_t row = is_pair ? ((((pmeta >> 8) & 31u) + (lane - (pmeta & 0xffu))) & 15u) : 0u;
        const float y = p0y - (float)row;
        const float y0 = clamp_(y, 0.0f, 1.0f);
        const float y1 = clamp_(y + sdy, 0.0f, 1.0f);
        const float s2_dy = y0 - y1;
        const float vec_y_recip = 1.0f / sdy;  // fine.wgsl:845
        const float t0 = (y0 - y) * vec_y_recip;
        const float t1 = (y1 - y) * vec_y_recip;
        const float s2_tx0 = t0 * sdx, s2_tx1 = t1 * sdx;  // (s2_*: read by stage 3 lanes with ds_bpermute)
        const float gx0 = p0x + s2_tx0, gx1 = p0x + s2_tx1;
        const float xmin0 = fmin_(gx0, gx1), xmax0 = fmax_(gx0, gx1);
        // (NaN or far-off spans, and segments stage 1 found insane: every pixel takes the full formula)
        const bool guard = (pmeta & (1u << 16)) != 0u && abs_(xmin0) <= 40.0f && abs_(xmax0) <= 40.0f;
        const int32_t c1 = guard ? iclamp_((int32_t)ceil_(xmax0 + 1.0e-3f), 0, 16) : 16;
        const int32_t n0 = guard ? iclamp_((int32_t)floor_(xmin0 - 1.0e-3f), 0, 16) : 0;
        const uint32_t ncross = (is_pair && s2_dy != 0.0f) ? (uint32_t)imax_(c1 - n0, 0) : 0u;
        if (is_pair) atomicOr((unsigned long long*)&F.rowmask[row], 1ull << lane);
        wave_sync();
        // position of the pair's entry: rows in ascending order, pairs of a row in pair (= segment) order
        const uint64_t rm_pair = F.rowmask[row];
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(rm_pair >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm_pair, 0u));
        const uint32_t c16 = (uint32_t)__builtin_popcountll(F.rowmask[lane & 15u]);
        uint32_t inc16 = c16;  // inclusive prefix inside each 16-lane DPP row (every row of lanes holds the 16 pixel rows)
        inc16 += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc16, JK_DPP_ROW_SHR(1), 0xf, 0xf, false);
        inc16 += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc16, JK_DPP_ROW_SHR(2), 0xf, 0xf, false);
        inc16 += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc16, JK_DPP_ROW_SHR(4), 0xf, 0xf, false);
        inc16 += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc16, JK_DPP_ROW_SHR(8), 0xf, 0xf, false);
        const uint32_t excl16 = inc16 - c16;
        const uint32_t pos = (__shfl(excl16, (int)row, 64) + rank) & 63u;
        if constexpr (CLIPS) {
            F.lanest[lane] = (ent_lds + __shfl(excl16, (int)ly, 64) * 16u) | ((lane < e_rel ? first : n_pairs) << 24);  // nothing consumed yet
        } else {
            s4_row_addr = ent_lds + __shfl(excl16, (int)ly, 64) * 16u;
            s4_done = 0u;  // nothing consumed yet
            s4_rowmask = F.rowmask[ly];
        }
        next_seg = so;
        if (is_pair) {
            // pixel q contributes dy from column c1 on: clamp(q + 1 - c1, 0, 1) is exactly 1 or 0, and dy * 0 = +-0 is what the
            // WGSL's a * dy gives left of the span (no compare + select pairs, which also need wait states)
            const float c1f = (float)c1;
            float cv[16];
#pragma unroll
            for (int q = 0; q < 16; q++) cv[q] = clamp_((float)(q + 1) - c1f, 0.0f, 1.0f) * s2_dy;
#pragma unroll
            for (int q = 0; q < 4; q++) F.ent[q][(FINE_WHATIF & 2) ? lane : pos] = make_float4(cv[4 * q], cv[4 * q + 1], cv[4 * q + 2], cv[4 * q + 3]);
        }
#if FINE_CROSS_INLANE
        // The pair's own lane evaluates its FIRST crossing pixel (round 5): all operands are in its registers, so the pixel
        // costs the formula and one store -- not a mark, a share of the owner scan and five lane shuffles.  Only the
        // pixels behind it (0.8 per pair on C3 instead of 1.8) go through the lane = crossing pixel passes below.
        if (ncross != 0u) {
            const uint32_t X = (uint32_t)n0 & 15u;
            const float i_f = (float)(X & 3u);
            const float startx = p0x - (float)(4u * (X >> 2));
            const float x0 = startx + s2_tx0, x1 = startx + s2_tx1;
            const float xmn0 = fmin_(x0, x1), xmx0 = fmax_(x0, x1);
            float xmin = fmin_(xmn0 - i_f, 1.0f) - 1.0e-6f;
            float xmax = xmx0 - i_f;
            float b = fmin_(xmax, 1.0f);
            float c = fmax_(b, 0.0f);
            float d = fmax_(xmin, 0.0f);
            float a = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
            ((float*)&F.ent[X >> 2][(FINE_WHATIF & 4) ? lane : pos])[X & 3u] = a * s2_dy;
        }
        const uint32_t nrest = ncross != 0u ? ncross - 1u : 0u;
        const int32_t n0r = n0 + 1;
#else
        const uint32_t nrest = ncross;
        const int32_t n0r = n0;
#endif
        const uint32_t sincl = wave_incl_scan_u32(nrest);
        const uint32_t spos = sincl - nrest;
        const uint32_t nspec = (FINE_SKIP == 1 || FINE_SKIP == 3) ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)sincl, 63);
        // stage 3, lane = crossing pixel, in passes of FB_SPEC of them (one pass unless the batch is full of long flat
        // segments).  Crossing pixel k of the batch belongs to the last pair whose first crossing pixel (spos) is <= k:
        // the pairs mark their starts in a byte array, a running maximum over the lanes turns the marks into owners.
        const uint32_t packed = ((uint32_t)n0r & 15u) | (pos << 4) | (spos << 10);
        constexpr uint32_t SPEC = (uint32_t)sizeof(F.specmark);  // crossing pixels per pass: 128 (64 in the clip instantiations)
        for (uint32_t pass = 0u; pass < nspec; pass += SPEC) {
            wave_sync();  // (the entries above / the previous pass's mark reads are done)
            if constexpr (SPEC == 128u) ((uint16_t*)F.specmark)[lane] = 0u; else F.specmark[lane] = 0u;
            wave_sync();
            if (nrest != 0u && spos - pass < SPEC) F.specmark[spos - pass] = (uint8_t)(lane + 1u);
            wave_sync();
            // the pair the first position of the pass belongs to when it does not start there
            const uint64_t before = __builtin_amdgcn_ballot_w64(nrest != 0u && spos < pass);
            uint32_t carry = before != 0ull ? 64u - (uint32_t)__builtin_clzll(before) : 0u;
            const uint32_t n_here = umin_(nspec - pass, SPEC);
            for (uint32_t k0 = 0u; k0 < n_here; k0 += 64u) {
                const uint32_t own = umax_(wave_incl_max_u32(F.specmark[k0 + lane]), carry);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)own, 63);
                const uint32_t j = (own - 1u) & 63u;
                // the pair lane's y-part results
                const float dy = __shfl(s2_dy, (int)j, 64), tx0 = __shfl(s2_tx0, (int)j, 64), tx1 = __shfl(s2_tx1, (int)j, 64);
                const float q0x = __shfl(p0x, (int)j, 64);
                const uint32_t pk = __shfl(packed, (int)j, 64);
                if (k0 + lane < n_here) {
                    const uint32_t X = ((pk & 15u) + (pass + k0 + lane - (pk >> 10))) & 15u;
                    const uint32_t epos = (pk >> 4) & 63u;
                    const uint32_t g = X >> 2;
                    const float i_f = (float)(X & 3u);
                    const float startx = q0x - (float)(4u * g);
                    const float x0 = startx + tx0, x1 = startx + tx1;
                    const float xmn0 = fmin_(x0, x1), xmx0 = fmax_(x0, x1);
                    float xmin = fmin_(xmn0 - i_f, 1.0f) - 1.0e-6f;
                    float xmax = xmx0 - i_f;
                    float b = fmin_(xmax, 1.0f);
                    float c = fmax_(b, 0.0f);
                    float d = fmax_(xmin, 0.0f);
                    float a = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
                    ((float*)&F.ent[(FINE_WHATIF & 4) ? (lane >> 4) : (X >> 2)][(FINE_WHATIF & 4) ? lane : epos])[(FINE_WHATIF & 4) ? (lane & 3u) : (X & 3u)] = a * dy;
                }
            }
        }
        wave_sync();
#if FINE_SKIP == 7
        {
            const uint32_t f = lane >> 4, r = lane & 15u;
            const uint32_t per = (e_rel + 3u) / 4u;
            const uint32_t s_lo = umin_(f * per, e_rel), s_hi = umin_(s_lo + per, e_rel);
            const uint32_t firstv = lane < e_rel ? first : n_pairs;
            const uint32_t p_lo = __shfl(firstv, (int)(s_lo & 63u), 64), p_hi = s_hi < 64u ? __shfl(firstv, (int)(s_hi & 63u), 64) : n_pairs;
            const uint64_t rm = F.rowmask[r];
            auto below = [](uint32_t P) -> uint64_t { return (1ull << (P & 63u)) - 1ull; };
            const uint32_t before = (uint32_t)__builtin_popcountll(rm & below(p_lo));
            const uint32_t cnt = (uint32_t)__builtin_popcountll(rm & below(p_hi)) - before;
            uint32_t addr = lds_addr(&F.ent[0][0]) + (__shfl(excl16, (int)r, 64) + before) * 16u;
            jk_v2f a[8];
#pragma unroll
            for (int q = 0; q < 8; q++) a[q] = {lyf, lyf};
            for (uint32_t trip = 0u; __builtin_amdgcn_ballot_w64(trip < cnt) != 0ull; trip++) {
                if (trip < cnt) {
#pragma unroll
                    for (uint32_t q = 0u; q < 4u; q++) {
                        const float4 v = lds_ld_f4(addr + q * (uint32_t)sizeof(F.ent[0]));
                        a[2 * q] += jk_v2f{v.x, v.y}; a[2 * q + 1] += jk_v2f{v.z, v.w};
                    }
                    addr += 16u;
                }
            }
            {   // one y_edge term per lane stands in for the slot's edge segments
                float2 ed;
                if constexpr (CLIPS) ed = make_float2(F.edge_y[s_lo & 63u], 1.0f); else ed = F.edge[s_lo & 63u];
                const float ye = ed.y * clamp_((float)r - ed.x + 1.0f, 0.0f, 1.0f);
#pragma unroll
                for (int q = 0; q < 8; q++) a[q] += jk_v2f{ye, ye};
            }
            wave_sync();
#pragma unroll
            for (uint32_t q = 0u; q < 4u; q++) F.ent[q][lane] = make_float4(a[2 * q].x, a[2 * q].y, a[2 * q + 1].x, a[2 * q + 1].y);
            wave_sync();
        }
#endif
      } else {
        (void)so;
      }
    };

    auto load_win = [&](uint32_t b) -> uint32_t {
        const uint32_t i = b + lane;
        return (i < ptcl_n && i >= b) ? ptcl[i] : 0u;
    };
    uint32_t wbase = pc;
    uint32_t wcur = load_win(wbase), wnext = load_win(wbase + 64u);
    if (ptcl_head == ~0u) return;  // fine.wgsl:889-893
    auto materialize = [&]() {  // perform the pending saves of BEGIN_CLIP (fine.wgsl:938-950), outermost first
        if constexpr (CLIPS) {
            while (pushed_depth < clip_depth) {  // uniform
                if (pushed_depth < FINE_LDS_LEVELS) {
#pragma unroll
                    for (int k = 0; k < 4; k++) S.lvl[pushed_depth][k][lane] = make_float4(rgba[k].x, rgba[k].y, rgba[k].z, rgba[k].w);
                } else if (pushed_depth < JL_BLEND_STACK_SPLIT) {
                    // (a level the launch has no scratch for can only be asked for when the caller's clip-depth hint was too
                    // small: the save is dropped rather than written over another tile's slice)
                    if (pushed_depth - FINE_LDS_LEVELS < scr_levels) {  // uniform
                        float4* g = clip_scratch + (((size_t)scratch_tile * scr_levels + (pushed_depth - FINE_LDS_LEVELS)) * 4u) * 64u + lane;
#pragma unroll
                        for (int k = 0; k < 4; k++) g[k * 64] = make_float4(rgba[k].x, rgba[k].y, rgba[k].z, rgba[k].w);
                    } else if (hint_overflow != nullptr && lane == 0u) {
                        atomicAdd(hint_overflow, 1u);  // detectable: jh_debug_clip_hint_overflows
                    }
                } else {
                    const uint32_t spill_base = blend_offset + (pushed_depth - JL_BLEND_STACK_SPLIT) * JL_TILE_WIDTH * JL_TILE_HEIGHT;
#pragma unroll
                    for (int k = 0; k < 4; k++) blend_spill.wr(spill_base + pix_spill(k), rgba[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) rgba[k] = v4(0, 0, 0, 0);
                pushed_depth += 1u;
            }
        }
    };
    // END_CLIP of a layer that is still pending (see the END_CLIP command below for why these are exact): true if the layer
    // was closed here -- rgba is then what the full formula gives --, false if the full formula has to be taken.
    // What the caller already knows (uniform), so that a RUN of empty layers over one unchanged backdrop -- C4: 213 per tile -- tests the
    // backdrop once instead of once per layer: area_is_one: the four areas are exactly 1 (a SOLID set them); `known`, bits: RK_NONEG no
    // channel of the wave's pixels is -0 (a `+ 0.0` was applied or the range was verified), RK_RANGE every channel is in [+0, 16] (bit
    // patterns), RK_LUM lum(cb) <= 1 for every pixel.  Bits are raised here when a test passes; the caller clears them whenever rgba may
    // have changed.  Without a -0 among the channels the `+ 0.0` of the plain arm is the identity: skipped.
    auto end_clip_fast = [&](uint32_t blend, float alpha, uint32_t level, bool area_is_one, uint32_t& known) -> bool {
        bool fast = false;
        // (0 <= alpha < inf as an unsigned comparison of the bit pattern -- a scalar compare; an alpha of -0 takes the full formula)
        if (clip_depth != 0u && pushed_depth <= level && f2u(alpha) < 0x7f800000u) {  // uniform
            const bool plain = (blend & 0x7fffu) == 0u;
            const uint32_t mixm = blend >> 8;
            const bool separable = (blend & 0xffu) == 0u && mixm >= 1u && mixm <= 14u;
            const bool needs_lum = mixm >= 12u;
            if (plain || separable) {
                const bool t_area = !area_is_one, t_rgba = !plain && (known & RK_RANGE) == 0u, t_lum = !plain && needs_lum && (known & RK_LUM) == 0u;
                if (!(t_area || t_rgba || t_lum)) {  // uniform: everything this layer needs has been established by an earlier one
                    fast = true;
                } else {
                    // (the range tests are unsigned comparisons of bit patterns: one comparison of the MAXIMUM pattern per
                    // group of values -- v_max3_u32 -- instead of one comparison and one mask operation per value)
                    auto umax3 = [](uint32_t a, uint32_t b, uint32_t c) -> uint32_t { return umax_(umax_(a, b), c); };
                    bool ok = true;
                    if (t_area) ok = umax_(umax3(f2u(area[0]), f2u(area[1]), f2u(area[2])), f2u(area[3])) <= 0x3f800000u;
                    if (t_rgba) {
                        uint32_t m[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) m[k] = umax_(umax3(f2u(rgba[k].x), f2u(rgba[k].y), f2u(rgba[k].z)), f2u(rgba[k].w));
                        ok = ok && umax_(umax3(m[0], m[1], m[2]), m[3]) <= 0x41800000u;
                    }
                    if (t_lum) {  // uniform
                        uint32_t lm = 0u;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float inv_backdrop_a = 1.0f / fmax_(rgba[k].w, 1e-15f);  // blend.wgsl:293-294
                            const float l = lum(v3(rgba[k].x * inv_backdrop_a, rgba[k].y * inv_backdrop_a, rgba[k].z * inv_backdrop_a));
                            lm = umax_(lm, f2u(l));
                        }
                        ok = ok && lm <= 0x3f800000u;
                    }
                    fast = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
                    if (fast && !plain) known |= RK_NONEG | RK_RANGE | (needs_lum ? RK_LUM : 0u);  // (what was not tested now was known before)
                }
                if (fast && plain && (known & RK_NONEG) == 0u) {
#pragma unroll
                    for (int k = 0; k < 4; k++) rgba[k] = v4(rgba[k].x + 0.0f, rgba[k].y + 0.0f, rgba[k].z + 0.0f, rgba[k].w + 0.0f);
                    known |= RK_NONEG;
                }
            }
        }
        return fast;
    };
#ifdef FINE_TIMING  // (variant builds only: where a tile's time goes; read back with jh_debug_clip_hint_overflows' buffer, words 8..)
    uint64_t tm_batch = 0ull, tm_walk = 0ull;
    uint32_t tm_nbatch = 0u, tm_nfill = 0u;
    const uint64_t tm_start = __builtin_readcyclecounter();
#endif
    // One FILL command (fill_path, fine.wgsl:824-878): leaves the finished coverage of the lane's four pixels in area[].
    auto do_fill = [&](uint32_t size_and_rule, uint32_t seg_data, int32_t backdrop) {
        uint32_t n_segs = size_and_rule >> 1;
        // segments behind the end of the buffer read as zero and contribute nothing (robust access), so a corrupt
        // count is cut to the buffer: the loops below are bounded by the buffer size, not by a stream word
        // (area coverage only: in the multisampled fill a zero segment touches the tile corner and does count)
        if constexpr (AA == 0) n_segs = umin_(n_segs, seg_data < segments_n ? segments_n - seg_data : 0u);
        bool even_odd = (size_and_rule & 1u) != 0u;
      if constexpr (AA == 0) {
        float backdrop_f = (float)backdrop;
#pragma unroll
        for (int k = 0; k < 4; k++) area[k] = backdrop_f;
#if FINE_SKIP != 0 && defined(__HIP_DEVICE_COMPILE__)
        // (differential builds: the areas must stay opaque per-lane values.  Without the walk they would be the uniform backdrop, and
        // the compiler would simplify the finalisation and the COMPOSITE behind them as well -- the round-3/4 splits charged that
        // saving to stage 4: 0.10 ms "for the walk" of which a third was the composite's.  Found in round 5.)
#pragma unroll
        for (int k = 0; k < 4; k++) asm volatile("" : "+v"(area[k]));
#endif
        uint32_t sa = seg_data, remaining = n_segs;
        while (remaining != 0u) {  // uniform
#if FINE_SKIP == 4
            batch_hi = sa + remaining;
            cur_base = sa;
#else
#ifdef FINE_TIMING
            const uint64_t tb0 = __builtin_readcyclecounter();
            const bool builds = sa - cur_base >= batch_hi - cur_base;
#endif
            if (sa - cur_base >= batch_hi - cur_base) build_batch(sa);
#ifdef FINE_TIMING
            const uint64_t tb1 = __builtin_readcyclecounter();
            tm_batch += tb1 - tb0; tm_nbatch += builds ? 1u : 0u; tm_nfill += 1u;
#endif
#endif
            const uint32_t take = umin_(remaining, batch_hi - sa);
            const uint32_t r0 = sa - cur_base;  // window-relative segments [r0, r0 + take)
            auto below = [](uint32_t P) -> uint64_t { return (1ull << (P & 63u)) - 1ull; };  // P <= 63
            // stage 4, lane = pixel quad of row ly: the entries of my row are in segment order, so the terms of the WGSL's
            // loop that can change my sum -- a*dy of the segments with a pair in my row -- are added in its order by
            // walking my row's list; a segment with a y_edge term (uniform: a bit of edge_mask) ends a run of such
            // additions for all rows, the term is added, and the walk goes on behind it.
            uint64_t my_rowmask;
            uint32_t my_first, row_addr, done;  // (first pair of the window segment in this lane; my row's entries; how many are consumed)
            if constexpr (CLIPS) {  // (the clip instantiations have no registers to spare: the state stays in LDS)
                my_rowmask = F.rowmask[ly];
                const uint32_t st = F.lanest[lane];
                my_first = st >> 24;
                row_addr = st & 0xffffu;
                done = (st >> 16) & 0xffu;
            } else {
                my_rowmask = s4_rowmask; my_first = s4_first; row_addr = s4_row_addr; done = s4_done;
            }
            auto first_of = [&](uint32_t sl) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)my_first, (int)(sl & 63u)); };  // lane = window segment
            if (sa != next_seg) done = (uint32_t)__builtin_popcountll(my_rowmask & below(first_of(r0)));  // (a fill that does not continue the previous one)
            uint32_t cur = row_addr + done * 16u;
            uint64_t em = edge_mask & (below(r0 + take) & ~below(r0));
#if FINE_SKIP == 6
            {   // floor build: the WGSL's own two packed additions per segment (the value is the lane's y coordinate: anything)
                jk_v2f a01 = {area[0], area[1]}, a23 = {area[2], area[3]};
                for (uint32_t s_ = 0u; s_ < take; s_++) {  // uniform
                    asm volatile("v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %1, %2 op_sel_hi:[1,0]" : "+v"(a01), "+v"(a23) : "v"(make_float2(lyf, lyf)));
                }
                area[0] = a01.x; area[1] = a01.y; area[2] = a23.x; area[3] = a23.y;
            }
#endif
#if FINE_SKIP == 7
            {
                const float4 v = lds_ld_f4(lds_addr(&F.ent[lx][((r0 & 3u) * 16u + ly) & 63u]));
                area[0] += v.x; area[1] += v.y; area[2] += v.z; area[3] += v.w;
            }
#endif
            for (; FINE_SKIP != 2 && FINE_SKIP != 4 && FINE_SKIP != 6 && FINE_SKIP != 7;) {  // uniform
                const uint32_t e_sl = em != 0ull ? (uint32_t)__builtin_ctzll(em) : 0u;
                const uint32_t seg_end = em != 0ull ? e_sl + 1u : r0 + take;  // the run covers window segments < seg_end
                done = (uint32_t)__builtin_popcountll(my_rowmask & below(first_of(seg_end)));
                const uint32_t hi = row_addr + done * 16u;
                // Walk my row's entries [cur, hi), two per trip (both loads in flight before the ordered adds): lanes drop
                // out of EXEC as their rows run out (no lane comes back inside a run), the loop ends when none is left.
                // Written in assembly: as C++ the compiler keeps two copies of the area registers around this loop (four
                // moves per trip) and cannot mask the loads (lanes reading a dummy entry collide with the live ones).
                {
                    jk_v2f a01 = {area[0], area[1]}, a23 = {area[2], area[3]};
                    uint64_t sv, s1;
                    uint32_t t;
                    asm volatile(
                        "s_mov_b64 %[sv], exec\n"
                        "1:\n"
                        "v_cmpx_lt_u32_e32 vcc, %[cur], %[hi]\n"
                        "s_cbranch_execz 3f\n"
                        "ds_read_b128 v[64:67], " FINE_WALK_RD "\n"
                        "v_add_u32_e32 %[t], 16, %[cur]\n"
                        "v_cmp_lt_u32_e32 vcc, %[t], %[hi]\n"
                        "s_mov_b64 %[s1], exec\n"
                        "s_and_b64 exec, exec, vcc\n"
                        "ds_read_b128 v[68:71], " FINE_WALK_RD " offset:16\n"
                        "s_mov_b64 exec, %[s1]\n"
                        "v_add_u32_e32 %[cur], 32, %[cur]\n"
                        "s_waitcnt lgkmcnt(1)\n"
                        "v_pk_add_f32 %[a01], %[a01], v[64:65]\n"
                        "v_pk_add_f32 %[a23], %[a23], v[66:67]\n"
                        "s_and_b64 exec, exec, vcc\n"
                        "s_waitcnt lgkmcnt(0)\n"
                        "v_pk_add_f32 %[a01], %[a01], v[68:69]\n"
                        "v_pk_add_f32 %[a23], %[a23], v[70:71]\n"
                        "s_mov_b64 exec, %[s1]\n"
                        "s_branch 1b\n"
                        "3:\n"
                        "s_mov_b64 exec, %[sv]\n"
                        : [cur] "+v"(cur), [a01] "+v"(a01), [a23] "+v"(a23), [sv] "=&s"(sv), [s1] "=&s"(s1), [t] "=&v"(t)
                        : [hi] "v"(hi) FINE_WALK_RD_OPERAND
                        : "vcc", "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
                    area[0] = a01.x; area[1] = a01.y; area[2] = a23.x; area[3] = a23.y;
                }
                cur = hi;
                if (em == 0ull) break;
                {
                    float2 ed;
                    if constexpr (CLIPS) ed = make_float2(F.edge_y[e_sl], (float)F.edge_s[e_sl]); else ed = F.edge[e_sl];
                    const float y_edge = ed.y * clamp_(lyf - ed.x + 1.0f, 0.0f, 1.0f);
                    area[0] += y_edge; area[1] += y_edge; area[2] += y_edge; area[3] += y_edge;
                }
                em &= em - 1ull;
            }
            if constexpr (CLIPS) lds_st_u8(lds_addr(&F.lanest[lane]) + 2u, (uint8_t)done); else s4_done = done;
#ifdef FINE_TIMING
            tm_walk += __builtin_readcyclecounter() - tb1;
#endif
            next_seg = sa + take;
            sa += take;
            remaining -= take;
        }
        if (even_odd) {
#pragma unroll
            for (int k = 0; k < 4; k++) { float a = area[k]; area[k] = abs_(a - 2.0f * round_(0.5f * a)); }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
#if FINE_FINAL_ASM && defined(__HIP_DEVICE_COMPILE__)
                // min(|a|, 1) as ONE instruction: the compiler puts a canonicalising v_max |a|, |a| in front of its v_min because the sum
                // comes out of inline assembly (it cannot know that an addition's result is never a signalling NaN).
                asm("v_min_f32_e64 %0, |%1|, 1.0" : "=v"(area[k]) : "v"(area[k]));
#else
                area[k] = fmin_(abs_(area[k]), 1.0f);
#endif
            }
        }
      } else {
        (void)n_segs; (void)even_odd;
        ms_fill<AA>(F, msb, lane, size_and_rule, seg_data, backdrop, segments, segments_n, mask_lut, mask_lut_n, area);
      }
    };
    auto ensure_window = [&]() {
    if (pc - wbase > 64u - FINE_TRIP_WORDS) {  // uniform: fewer words than a trip may need are in `wcur`
        // Re-base the window at pc: lane k takes word pc + k from the two windows (ds_bpermute: a lane shuffle through
        // the LDS crossbar, no LDS storage), and the window behind the new one is requested.
        const uint32_t sh = pc - wbase;  // <= 64: a trip advances pc by at most FINE_TRIP_WORDS words (jumps reload)
        const uint32_t src = lane + sh;
        const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src & 63u) << 2), (int)wcur);
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src & 63u) << 2), (int)wnext);
        wcur = src < 64u ? a : b;      // src <= 127
        wbase = pc;
        wnext = load_win(wbase + 64u);
    }
    };
    for (uint32_t guard = 0; guard < (1u << 24); guard++) {
        // (a trip consumes at most FINE_TRIP_WORDS words: up to three BEGIN_CLIPs and a SOLID in front of a command of up to nine)
        ensure_window();
        uint32_t woff = pc - wbase;  // <= 64 - FINE_TRIP_WORDS: the words this trip may consume are all in wcur
        auto W = [&](uint32_t k) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)wcur, (int)(woff + k)); };
        uint32_t tag = W(0);
        {
            // FILL followed by COLOR, the usual pair, in a loop of its own: ONE definition of the colour registers around ONE back edge
            // (as an arm of the general decoder below the pair carried that decoder's flag variables, state copies and branch chain).
            for (uint32_t hot = 0; hot < (1u << 24) && tag == JL_CMD_FILL && W(4) == JL_CMD_COLOR; hot++) {  // uniform
                do_fill(W(1), W(2), (int32_t)W(3));
#if FINE_COLOR_BPERM
                // (the colour through the LDS crossbar into vector registers instead of four v_readlane with a scalar lane select, 8 cycles
                // of the vector pipe each: the kernel is bound by vector issue, the LDS pipe is a third busy)
                auto WB = [&](uint32_t k) -> float { return u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)((woff + k) << 2), (int)wcur)); };
                const V4 fgc = v4(WB(5), WB(6), WB(7), WB(8));
#else
                const V4 fgc = v4(u2f(W(5)), u2f(W(6)), u2f(W(7)), u2f(W(8)));
#endif
                pc += 9u;
                materialize();
#pragma unroll
                for (int k = 0; k < 4; k++)
                    rgba[k] = FINE_SKIP == 5 ? v4(rgba[k].x + area[k], rgba[k].y + fgc.x, rgba[k].z + fgc.y, rgba[k].w + fgc.z * fgc.w) : over(rgba[k], fgc, area[k]);
                ensure_window();
                woff = pc - wbase;
                tag = W(0);
            }
        }
        if constexpr (CLIPS) {
            // Empty layers -- BEGIN_CLIP ... [SOLID] END_CLIP with nothing drawn in between, 213 per tile in the C4 scene -- likewise in a
            // loop of their own: the BEGIN_CLIPs only count, SOLID sets the area, and an END_CLIP that the shortcut can close leaves
            // the loop's state as the general decoder would.  Anything else falls through to the decoder with the words consumed so
            // far accounted for (exactly what its own folding of BEGIN_CLIP / SOLID does).
            // Inside the loop nothing but the plain arm's `+ 0.0` touches rgba and nothing but SOLID touches the area, so what one
            // layer's tests established holds for the layers behind it: once the area is 1 and the backdrop's tests have passed, the
            // commands of an empty layer need NOTHING but counting -- the scalar loop at the head of every trip.  (Round 4.  The C4
            // tile is bound by the CU's one scalar pipe: 118 scalar instructions per empty layer before, the compiler's boolean
            // bookkeeping around the decoder; the counting loop is one compare-and-branch per condition.)
            uint32_t rgba_known = 0u;  // (of this run of empty layers: see end_clip_fast)
            bool area_one = false;
            for (uint32_t hot = 0; hot < (1u << 24); hot++) {  // uniform
                if (area_one) {
                    // The counting loop, by hand: one compare-and-branch per condition, ~47 scalar instructions per BEGIN_CLIP SOLID END_CLIP
                    // (as C++ the structuriser turned the chain of exits into state codes and mask bookkeeping, no better than before).
                    //   BEGIN_CLIP: depth + 1.   SOLID: nothing (the area is 1).   END_CLIP blend alpha: closes an EMPTY layer (one that
                    //   was never materialised: pushed_depth < depth) if 0 <= alpha < inf (bit pattern), the compose operator is src-over
                    //   and what end_clip_fast would test for the mix mode -- plain / clip: RK_NONEG, modes 1..11: + RK_RANGE, hue /
                    //   saturation / color: + RK_LUM -- is already in rgba_known.  Anything else, or fewer than FINE_TRIP_WORDS words left
                    //   in the register window: out (the window is re-based and the loop entered again, or the trip below takes over).
                    for (uint32_t sk = 0; sk < (1u << 24); sk++) {  // uniform
                        ensure_window();
#if FINE_VECTOR_LAYERS
                        // Round 5: the same counting, but by ALL 64 LANES AT ONCE on the register window instead of one scalar
                        // branch chain per command (47 scalar instructions per BEGIN_CLIP SOLID END_CLIP, 10 k of the C4 tile's 19 k).
                        // Lane k looks at word k of the window and at its two predecessors (DPP wave shifts) and decides whether
                        // the stream is still countable THERE, given that it was up to there: a word behind an END_CLIP tag is its
                        // blend (low byte 0 = src-over, mix class covered by rgba_known), the word behind that its alpha
                        // (0 <= alpha < inf by bit pattern, and not one of the bit patterns 3 / 10 / 11, so that inside the accepted
                        // stretch a word of value 3 / 10 / 11 is always a tag); every other word must be a BEGIN_CLIP, SOLID or
                        // END_CLIP tag, an END_CLIP with its nesting depth > 0 and > pushed_depth (prefix counts of the BEGIN /
                        // END tags below the lane: two mbcnt) and its two payload words inside the window.  The first lane that
                        // says no ends the stretch; it is cut back to a command boundary and consumed in one step.
                        {
                            const uint32_t wo = pc - wbase;  // <= 64 - FINE_TRIP_WORDS
                            const uint32_t w0 = wcur;
                            const uint32_t w1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x138, 0xf, 0xf, false);  // wave_shr:1: word k - 1
                            const uint32_t w2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x138, 0xf, 0xf, false);  // word k - 2
                            const bool in = lane >= wo;
                            const bool tag0 = (w0 == JL_CMD_BEGIN_CLIP) | (w0 == JL_CMD_SOLID) | (w0 == JL_CMD_END_CLIP);
                            const bool is_blend = (lane >= wo + 1u) & (w1 == JL_CMD_END_CLIP);
                            const bool is_alpha = (lane >= wo + 2u) & (w2 == JL_CMD_END_CLIP) & (w1 != JL_CMD_END_CLIP);
                            const uint64_t begins = __builtin_amdgcn_ballot_w64(in & (w0 == JL_CMD_BEGIN_CLIP));
                            const uint64_t ends_ = __builtin_amdgcn_ballot_w64(in & (w0 == JL_CMD_END_CLIP));
                            const uint32_t nb_below = __builtin_amdgcn_mbcnt_hi((uint32_t)(begins >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)begins, 0u));
                            const uint32_t ne_below = __builtin_amdgcn_mbcnt_hi((uint32_t)(ends_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ends_, 0u));
                            const uint32_t depth_here = clip_depth + nb_below - ne_below;
                            const uint32_t need = (w0 & 0x7fffu) == 0u ? RK_NONEG : (w0 < 0xc00u ? (RK_NONEG | RK_RANGE) : (w0 < 0xf00u ? (RK_NONEG | RK_RANGE | RK_LUM) : 0xffffffffu));
                            // (no branches: every lane evaluates the three tests, the kind of word selects -- as an if / else chain the
                            // compiler built three divergent regions with their exec-mask bookkeeping, ~100 scalar instructions per step)
                            const bool ok_blend = ((w0 & 0xffu) == 0u) & ((need & ~rgba_known) == 0u);
                            const bool ok_alpha = (w0 <= 0x7f7fffffu) & !tag0;
                            const bool ok_tag = tag0 & ((w0 != JL_CMD_END_CLIP) | ((depth_here != 0u) & (pushed_depth < depth_here) & (lane <= 61u)));
                            const bool ok = (is_blend & ok_blend) | (!is_blend & is_alpha & ok_alpha) | (!is_blend & !is_alpha & ok_tag);
                            const uint64_t bad_mask = __builtin_amdgcn_ballot_w64(in & !ok);
                            const uint32_t bad = bad_mask != 0ull ? (uint32_t)__builtin_ctzll(bad_mask) : 64u;
                            auto below = [](uint32_t P) -> uint64_t { return ~(~0ull << (P & 63u)) | (0ull - (uint64_t)(P >> 6)); };  // P <= 64
                            const uint64_t tags = __builtin_amdgcn_ballot_w64(in & tag0 & !is_blend & !is_alpha) & below(bad);
                            if (tags != 0ull) {  // uniform
                                const uint32_t last = 63u - (uint32_t)__builtin_clzll(tags);
                                const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)w0, (int)last) == JL_CMD_END_CLIP ? 3u : 1u;
                                const uint32_t end = last + len <= bad ? last + len : last;
                                const uint64_t range = below(end) & ~below(wo);
                                clip_depth = clip_depth + (uint32_t)__builtin_popcountll(begins & range) - (uint32_t)__builtin_popcountll(ends_ & range);
                                pc += end - wo;
                            }
                        }
#else
#if defined(__HIP_DEVICE_COMPILE__)
                        uint32_t t_i, t_t, t_b;
                        asm volatile(
                            "1:\n"
                            "  s_sub_u32 %[i], %[pc], %[wb]\n"
                            "  s_cmp_gt_u32 %[i], %[lim]\n"
                            "  s_cbranch_scc1 9f\n"
                            "  v_readlane_b32 %[t], %[w], %[i]\n"
                            "  s_cmp_eq_u32 %[t], 10\n"
                            "  s_cbranch_scc0 2f\n"
                            "  s_add_u32 %[d], %[d], 1\n"
                            "  s_add_u32 %[pc], %[pc], 1\n"
                            "  s_branch 1b\n"
                            "2:\n"
                            "  s_cmp_eq_u32 %[t], 3\n"
                            "  s_cbranch_scc0 3f\n"
                            "  s_add_u32 %[pc], %[pc], 1\n"
                            "  s_branch 1b\n"
                            "3:\n"
                            "  s_cmp_eq_u32 %[t], 11\n"
                            "  s_cbranch_scc0 9f\n"
                            "  s_cmp_eq_u32 %[d], 0\n"
                            "  s_cbranch_scc1 9f\n"
                            "  s_cmp_ge_u32 %[pd], %[d]\n"
                            "  s_cbranch_scc1 9f\n"
                            "  s_add_u32 %[t], %[i], 1\n"
                            "  s_add_u32 %[i], %[i], 2\n"
                            "  v_readlane_b32 %[b], %[w], %[t]\n"
                            "  v_readlane_b32 %[i], %[w], %[i]\n"
                            "  s_cmp_gt_u32 %[i], 0x7f7fffff\n"
                            "  s_cbranch_scc1 9f\n"
                            "  s_and_b32 %[t], %[b], 0xff\n"
                            "  s_cbranch_scc1 9f\n"
                            "  s_mov_b32 %[t], 1\n"
                            "  s_and_b32 %[i], %[b], 0x7fff\n"
                            "  s_cbranch_scc0 4f\n"
                            "  s_mov_b32 %[t], 3\n"
                            "  s_cmp_lt_u32 %[b], 0xc00\n"
                            "  s_cbranch_scc1 4f\n"
                            "  s_mov_b32 %[t], 7\n"
                            "  s_cmp_lt_u32 %[b], 0xf00\n"
                            "  s_cbranch_scc0 9f\n"
                            "4:\n"
                            "  s_andn2_b32 %[t], %[t], %[kn]\n"
                            "  s_cbranch_scc1 9f\n"
                            "  s_sub_u32 %[d], %[d], 1\n"
                            "  s_add_u32 %[pc], %[pc], 3\n"
                            "  s_branch 1b\n"
                            "9:\n"
                            : [pc] "+s"(pc), [d] "+s"(clip_depth), [i] "=&s"(t_i), [t] "=&s"(t_t), [b] "=&s"(t_b)
                            : [wb] "s"(wbase), [pd] "s"(pushed_depth), [kn] "s"(rgba_known), [w] "v"(wcur), [lim] "n"(64 - (int)FINE_TRIP_WORDS)
                            : "scc");
#endif
#endif
                        if (pc - wbase <= 64u - FINE_TRIP_WORDS) break;  // (else: the window ran out, not the commands)
                    }
                    woff = pc - wbase;
                    tag = W(0);  // (what the decoder below sees if this trip leaves the loop)
                }
#ifdef FINE_NO_COUNTING_LOOP  // (A/B builds only: every layer is tested and decoded by the trip below, as before round 4)
                area_one = false; rgba_known = 0u;
#endif
                // (a JUMP followed inside this loop -- so that area_one / rgba_known would survive the chunk boundary -- measured 8 % SLOWER on
                // C4 and 7 % on the nested variant: the window registers redefined inside the loop cost more than the five trips per tile save)
                uint32_t k = 0u, nb = 0u;
                while (nb < 3u && W(k) == JL_CMD_BEGIN_CLIP) { nb++; k++; }
                const bool solid = W(k) == JL_CMD_SOLID;
                if (solid) k++;
#if FINE_LAYER_FILL
                // (round 5) BEGIN_CLIP ... FILL END_CLIP: the layer of a tile that the clip path covers only in part -- 25 per tile in
                // the C4 scene -- stays in this loop as well: its coverage is evaluated, and the END_CLIP that closes a layer with
                // nothing drawn in it takes the shortcut with the area tested (a trip through the general decoder is ~250 scalar
                // instructions).  Up to 3 + 4 + 3 = 10 of the trip's FINE_TRIP_WORDS words.
                const bool filled = !solid && W(k) == JL_CMD_FILL && W(k + 4u) == JL_CMD_END_CLIP;
                if (!filled && W(k) != JL_CMD_END_CLIP) break;
#else
                const bool filled = false;
                if (W(k) != JL_CMD_END_CLIP) break;
#endif
                // (the decoder's order: BEGIN_CLIPs, then SOLID, then the command)
                clip_depth += nb;
                if (solid && !area_one) {
#pragma unroll
                    for (int q = 0; q < 4; q++) area[q] = 1.0f;
                    area_one = true;
                }
                if (filled) {  // uniform
                    do_fill(W(k + 1u), W(k + 2u), (int32_t)W(k + 3u));
                    k += 4u;
                    area_one = false;
                }
                pc += k; woff += k;
                tag = JL_CMD_END_CLIP;
                if (clip_depth == 0u) break;  // (a stray END_CLIP: the decoder's business)
                if (!end_clip_fast(W(1), u2f(W(2)), clip_depth - 1u, area_one, rgba_known)) break;
                clip_depth -= 1u;
                pc += 3u;
                ensure_window();
                woff = pc - wbase;
                tag = W(0);
            }
        }
        if constexpr (CLIPS) {
            // BEGIN_CLIP only counts (the save is deferred, see pushed_depth) and SOLID only sets the area: both are
            // consumed in front of the command that follows instead of in trips of their own -- a trip of this loop
            // carries the sixteen colour registers around its back edge, which costs far more than these commands do.
            // (BEGIN_CLIP SOLID END_CLIP is what a tile in the middle of an empty clip layer sees: C4 has 213 per tile.)
            for (int it = 0; it < 3 && tag == JL_CMD_BEGIN_CLIP; it++) {  // uniform
                clip_depth += 1u;
                pc += 1u; woff += 1u;
                tag = W(0);
            }
            if (tag == JL_CMD_SOLID) {  // uniform
#pragma unroll
                for (int k = 0; k < 4; k++) area[k] = 1.0f;
                pc += 1u; woff += 1u;
                tag = W(0);
            }
        }
        if (tag == JL_CMD_END) break;
        // A solid colour (uniform: it sits in scalar registers) is composited at ONE place below, whichever command brought
        // it -- one definition of rgba per trip keeps the sixteen colour registers where they are (three composite sites
        // cost ~35 register moves per command).
        bool have_fg = false;
        V4 fg = v4(0, 0, 0, 0);
        if (tag == JL_CMD_FILL) {  // fill_path, fine.wgsl:824-878
            do_fill(W(1), W(2), (int32_t)W(3));
            pc += 4u;
            if (W(4) == JL_CMD_COLOR) {  // the usual pair: no second trip through the decoder
                fg = v4(u2f(W(5)), u2f(W(6)), u2f(W(7)), u2f(W(8)));
                have_fg = true;
                pc += 5u;
            }
        } else if (tag == JL_CMD_SOLID) {
#pragma unroll
            for (int k = 0; k < 4; k++) area[k] = 1.0f;
            pc += 1u;
        } else if (tag == JL_CMD_COLOR) {
            fg = v4(u2f(W(1)), u2f(W(2)), u2f(W(3)), u2f(W(4)));
            have_fg = true;
            pc += 5u;
        } else if (CLIPS && tag == JL_CMD_BEGIN_CLIP) {
            clip_depth += 1u;  // (a fourth BEGIN_CLIP in a row; the save is deferred, see pushed_depth)
            pc += 1u;
        } else if (CLIPS && tag == JL_CMD_END_CLIP) {
            pc += 3u;
            const uint32_t blend = W(1);
            const float alpha = u2f(W(2));
            const uint32_t level = clip_depth - 1u;
            // A layer that is still pending is empty: the blend's source rgba_layer * area * alpha is +0 in every lane (given
            // 0 <= area <= 1 and 0 <= alpha < inf), its backdrop is rgba itself, and blend_mix_compose(bg, +0, mode) reduces
            // EXACTLY to
            //   (mode & 0x7fff) == 0 (normal / clip, src-over):  bg * (1 - 0) + 0 = bg + 0 per channel (x * 1 is the identity;
            //                      the addition of +0 is kept: it turns a -0 of the backdrop into +0 like the full formula);
            //   src-over with a separable mix mode (1..11):       (bg.c * 1 + cs'.c * 0, 0 + bg.a * 1), where cs' is a FINITE
            //                      value for 0 <= bg <= 16 (every operation of blend_mix on cb = bg.c / max(bg.a, 1e-15) and
            //                      cs = 0 stays far below overflow), so cs'.c * 0 = +-0 and bg.c + (+-0) = bg.c for bg.c >= +0.
            //   src-over with hue / saturation / color (12..14): cs = 0 makes set_sat return zeros and set_lum start from
            //                      (l, l, l) with l = lum(cb); for 0 <= l <= 1 clip_color changes nothing, so the mixed colour
            //                      is the finite (l, l, l) and the result is bg as for the separable modes.  l is evaluated here
            //                      with the operations of the full formula (cb = bg.c * (1 / max(bg.a, 1e-15)), lum).
            //                      (luminosity, 15, divides by lum(c) - min(c) of c = cb - l, which vanishes on grey
            //                      backdrops: no shortcut.)
            // Anything else (other compose operators, a backdrop outside [+0, 16], NaNs) performs the pending saves and takes
            // the full formula.  The unsigned comparison of the bit patterns tests "+0 <= v <= limit".
            uint32_t no_memo = 0u;
            const bool fast = end_clip_fast(blend, alpha, level, false, no_memo);
            if (!fast) {
                materialize();
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    V4 bg;
                    if (level < FINE_LDS_LEVELS) {
                        const float4 t = S.lvl[level & (FINE_LDS_LEVELS - 1u)][k][lane];  // (written by this lane: no synchronisation)
                        bg = v4(t.x, t.y, t.z, t.w);
                    } else if (level < JL_BLEND_STACK_SPLIT) {
                        float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if (level - FINE_LDS_LEVELS < scr_levels)  // uniform (see materialize)
                            t = clip_scratch[(((size_t)scratch_tile * scr_levels + (level - FINE_LDS_LEVELS)) * 4u + (uint32_t)k) * 64u + lane];
                        bg = v4(t.x, t.y, t.z, t.w);
                    } else {
                        const uint32_t spill_base = blend_offset + (level - JL_BLEND_STACK_SPLIT) * JL_TILE_WIDTH * JL_TILE_HEIGHT;
                        bg = blend_spill.rd(spill_base + pix_spill(k));
                    }
                    V4 src = v4(rgba[k].x * area[k] * alpha, rgba[k].y * area[k] * alpha, rgba[k].z * area[k] * alpha, rgba[k].w * area[k] * alpha);
#if defined(FINE_DIFF_BLEND) && defined(JH_VARIANT_BUILD)  // (differential build: every full END_CLIP blends plain src-over)
                    rgba[k] = blend_mix_compose(bg, src, 0u);
#else
                    // (the plain / clip case inlined here instead of behind the call -- two of three full blends of a nest of clips -- measured
                    // SLOWER, 2928 -> 3095 us on nested C4: the instantiation has no register to spare, ten more spills)
                    rgba[k] = blend_mix_compose(bg, src, blend);
#endif
                }
                pushed_depth = level;
            }
            clip_depth = level;
        } else if (tag == JL_CMD_JUMP) {
            pc = W(1);
            wbase = pc;
            wcur = load_win(wbase);
            wnext = load_win(wbase + 64u);
            continue;
        } else if (!PAINTS && (tag == JL_CMD_LIN_GRAD || tag == JL_CMD_RAD_GRAD || tag == JL_CMD_SWEEP_GRAD || tag == JL_CMD_IMAGE)) {
            // This instantiation is only launched when no ramp and no image is bound: every texel fetch of the
            // WGSL returns 0 then, i.e. the command composites a transparent colour.
            have_fg = true;
            pc += (tag == JL_CMD_IMAGE) ? 2u : 3u;
        } else if (PAINTS && tag == JL_CMD_LIN_GRAD) {
            materialize();
            pc += 3u;
            uint32_t index_mode = W(1);
            uint32_t index = index_mode >> 2, ext = index_mode & 3u;
            uint32_t io = W(2);
            float line_x = u2f(I(io)), line_y = u2f(I(io + 1u)), line_c = u2f(I(io + 2u));
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float d = line_x * xyx + line_y * xyy + line_c;
                float my_d = d + line_x * pix_i(k);
                int32_t x = to_i32(round_(extend_mode(my_d, ext) * 511.0f));
                rgba[k] = over(rgba[k], load_grad(x, index), area[k]);
            }
#if defined(FINE_DIFF_GRAD) && defined(JH_VARIANT_BUILD)  // (differential build: a radial gradient composites like a transparent colour)
        } else if (PAINTS && tag == JL_CMD_RAD_GRAD) {
            have_fg = true;
            pc += 3u;
#endif
        } else if (PAINTS && tag == JL_CMD_RAD_GRAD) {
            materialize();
            pc += 3u;
            uint32_t index_mode = W(1);
            uint32_t index = index_mode >> 2, ext = index_mode & 3u;
            uint32_t io = W(2);
            float m0 = u2f(I(io)), m1 = u2f(I(io + 1u)), m2 = u2f(I(io + 2u)), m3 = u2f(I(io + 3u));
            float xl0 = u2f(I(io + 4u)), xl1 = u2f(I(io + 5u));
            float focal_x = u2f(I(io + 6u));
            float radius = u2f(I(io + 7u));
            uint32_t flags_kind = I(io + 8u);
            uint32_t flags = flags_kind >> 3, kind = flags_kind & 7u;
            bool is_strip = kind == JL_RAD_GRAD_KIND_STRIP, is_circular = kind == JL_RAD_GRAD_KIND_CIRCULAR;
            bool is_focal_on_circle = kind == JL_RAD_GRAD_KIND_FOCAL_ON_CIRCLE;
            bool is_swapped = (flags & JL_RAD_GRAD_SWAPPED) != 0u;
            float r1_recip = is_circular ? 0.0f : (1.0f / radius);
            float less_scale = (is_swapped || (1.0f - focal_x) < 0.0f) ? -1.0f : 1.0f;
            float t_sign = sign_(1.0f - focal_x);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float mx = xyx + pix_i(k), my = xyy;
                float x = m0 * mx + m2 * my + xl0;
                float y = m1 * mx + m3 * my + xl1;
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
                    int32_t gxi = to_i32(round_(t * 511.0f));
                    rgba[k] = over(rgba[k], load_grad(gxi, index), area[k]);
                }
            }
        } else if (PAINTS && tag == JL_CMD_SWEEP_GRAD) {
            materialize();
            pc += 3u;
            uint32_t index_mode = W(1);
            uint32_t index = index_mode >> 2, ext = index_mode & 3u;
            uint32_t io = W(2);
            float m0 = u2f(I(io)), m1 = u2f(I(io + 1u)), m2 = u2f(I(io + 2u)), m3 = u2f(I(io + 3u));
            float xl0 = u2f(I(io + 4u)), xl1 = u2f(I(io + 5u));
            float t0 = u2f(I(io + 6u)), t1 = u2f(I(io + 7u));
            float scale = 1.0f / (t1 - t0);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float mx = xyx + pix_i(k), my = xyy;
                float x = m0 * mx + m2 * my + xl0;
                float y = m1 * mx + m3 * my + xl1;
                float xabs = abs_(x), yabs = abs_(y);
                float slope = fmin_(xabs, yabs) / fmax_(xabs, yabs);
                float s = slope * slope;
                float phi = slope * (0.15912117063999176025390625f +
                                     s * (-5.185396969318389892578125e-2f + s * (2.476101927459239959716796875e-2f + s * (-7.0547382347285747528076171875e-3f))));
                phi = (xabs < yabs) ? (0.25f - phi) : phi;
                phi = (x < 0.0f) ? (0.5f - phi) : phi;
                phi = (y < 0.0f) ? (1.0f - phi) : phi;
                phi = (phi != phi) ? 0.0f : phi;
                phi = (phi - t0) * scale;
                float t = extend_mode(phi, ext);
                int32_t ramp_x = to_i32(round_(t * 511.0f));
                rgba[k] = over(rgba[k], load_grad(ramp_x, index), area[k]);
            }
        } else if (PAINTS && tag == JL_CMD_IMAGE) {
            materialize();
            pc += 2u;
            uint32_t io = W(1);
            float m0 = u2f(I(io)), m1 = u2f(I(io + 1u)), m2 = u2f(I(io + 2u)), m3 = u2f(I(io + 3u));
            float xl0 = u2f(I(io + 4u)), xl1 = u2f(I(io + 5u));
            uint32_t index = I(io + 6u);
            uint32_t width_height = I(io + 7u);
            float ew = (float)(width_height >> 16), eh = (float)(width_height & 0xffffu);
            const uint8_t* ipx = nullptr;
            uint32_t iw = 0, ih = 0;
            bool is_srgb = false;
            if (images.table != nullptr) {
                if (index < (uint32_t)images.n) {  // index comes from a uniform info word: scalar loads
                    const JhImageDesc d = images.table[index];
                    ipx = (const uint8_t*)d.ptr; iw = d.width; ih = d.height; is_srgb = d.srgb != 0u;
                }
            } else {
#pragma unroll
                for (int q = 0; q < FINE_MAX_IMAGES; q++)
                    if ((uint32_t)q == index && q < images.n) { ipx = images.px[q]; iw = images.w[q]; ih = images.h[q]; is_srgb = ((images.srgb_mask >> q) & 1u) != 0u; }
            }
            auto texel = [&](int32_t tx, int32_t ty) -> V4 {
                if (!ipx || tx < 0 || ty < 0 || (uint32_t)tx >= iw || (uint32_t)ty >= ih) return v4(0, 0, 0, 0);
                uint32_t raw = *(const uint32_t*)(ipx + ((size_t)ty * iw + (size_t)tx) * 4);
                float r, g, b, a = (float)(raw >> 24) / 255.0f;
                if (is_srgb) {
                    r = kSrgbToLinear[raw & 0xffu]; g = kSrgbToLinear[(raw >> 8) & 0xffu]; b = kSrgbToLinear[(raw >> 16) & 0xffu];
                } else {
                    r = (float)(raw & 0xffu) / 255.0f; g = (float)((raw >> 8) & 0xffu) / 255.0f; b = (float)((raw >> 16) & 0xffu) / 255.0f;
                }
                return v4(r * a, g * a, b * a, a);  // premul_alpha, fine.wgsl:1105-1107
            };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float mx = xyx + pix_i(k), my = xyy;
                float u = m0 * mx + m2 * my + xl0;
                float v = m1 * mx + m3 * my + xl1;
                if (u < ew && v < eh && area[k] != 0.0f) {
                    float fu = floor_(u), fv = floor_(v), cu = ceil_(u), cv = ceil_(v);
                    float fru = fract_(u), frv = fract_(v);
                    V4 a = texel(to_i32(fu), to_i32(fv));
                    V4 bq = texel(to_i32(fu), to_i32(cv));
                    V4 cq = texel(to_i32(cu), to_i32(fv));
                    V4 dq = texel(to_i32(cu), to_i32(cv));
                    V4 ab = v4(mix_(a.x, bq.x, frv), mix_(a.y, bq.y, frv), mix_(a.z, bq.z, frv), mix_(a.w, bq.w, frv));
                    V4 cd = v4(mix_(cq.x, dq.x, frv), mix_(cq.y, dq.y, frv), mix_(cq.z, dq.z, frv), mix_(cq.w, dq.w, frv));
                    V4 fg = v4(mix_(ab.x, cd.x, fru), mix_(ab.y, cd.y, fru), mix_(ab.z, cd.z, fru), mix_(ab.w, cd.w, fru));
                    rgba[k] = over(rgba[k], fg, area[k]);
                }
            }
        } else {
            break;  // unknown tag: the WGSL would never advance; stop instead of hanging the GPU
        }
        if (have_fg) {  // uniform
            materialize();
#pragma unroll
            for (int k = 0; k < 4; k++) rgba[k] = FINE_SKIP == 5 ? v4(rgba[k].x + area[k], rgba[k].y + fg.x, rgba[k].z + fg.y, rgba[k].w + fg.z * fg.w) : over(rgba[k], fg, area[k]);
        }
    }
#ifdef FINE_TIMING
    if (hint_overflow != nullptr && lane == 0u) {
        unsigned long long* t = (unsigned long long*)(hint_overflow + 8);
        atomicAdd(&t[0], (unsigned long long)(__builtin_readcyclecounter() - tm_start));
        atomicAdd(&t[1], (unsigned long long)tm_batch);
        atomicAdd(&t[2], (unsigned long long)tm_walk);
        atomicAdd(&t[3], (unsigned long long)tm_nbatch);
        atomicAdd(&t[4], (unsigned long long)tm_nfill);
        atomicAdd(&t[5], 1ull);
    }
#endif
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no window may still be in flight to this wave's LDS when it ends
    // fine.wgsl:1092-1102: un-premultiply, store RGBA16F (four adjacent pixels = 32 bytes per lane; 4 lanes = one 128-B row)
    const uint32_t cx0 = tile_x * 16u + lx * 4u;
    const uint32_t cy = tile_y * 16u + ly;
    if (cy < out_h) {
        uint32_t packed[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            V4 fg = rgba[e];
            float a_inv = 1.0f / fmax_(fg.w, 1e-6f);
            uint32_t r = f32_to_f16(fg.x * a_inv), g = f32_to_f16(fg.y * a_inv), b = f32_to_f16(fg.z * a_inv), a = f32_to_f16(fg.w);
            packed[e * 2] = r | (g << 16);
            packed[e * 2 + 1] = b | (a << 16);
        }
        uint16_t* row = output + ((size_t)cy * out_w + cx0) * 4;
        if (cx0 + 3u < out_w && ((out_w & 1u) == 0u)) {
            *(uint4*)row = make_uint4(packed[0], packed[1], packed[2], packed[3]);
            *(uint4*)(row + 8) = make_uint4(packed[4], packed[5], packed[6], packed[7]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (cx0 + (uint32_t)e < out_w) *(uint2*)(row + 4 * e) = make_uint2(packed[e * 2], packed[e * 2 + 1]);
        }
    }
}

}  // namespace

// [config, segments, ptcl, info, blend_spill, output image, gradients image, images[] (, mask_lut for MSAA)]
// aa = 0: fine_area, 8: fine_msaa8, 16: fine_msaa16
static int launch_fine(const JhLaunch& L, int aa) {
    if (L.nb < (aa ? 9 : 7)) return -1;
    if (L.gx == 0 || L.gy == 0) return 0;
    const uint32_t* mask_lut = aa ? (const uint32_t*)L.b[8].ptr : nullptr;
    uint32_t mask_lut_n = aa ? (uint32_t)(L.b[8].size / 4) : 0u;
    if (!mask_lut) mask_lut_n = 0u;
    auto cfg = (const JlConfig*)L.b[0].ptr;
    uint32_t segments_n = (uint32_t)(L.b[1].size / sizeof(JlSegment));
    uint32_t ptcl_n = (uint32_t)(L.b[2].size / 4);
    uint32_t info_n = (uint32_t)(L.b[3].size / 4);
    auto spill = mkbuf<V4>(L.b[4].ptr, L.b[4].size);
    const JhBound& out = L.b[5];
    const JhBound& grad = L.b[6];
    if (out.format != JL_RGBA16_FLOAT || !out.ptr) return -1;
    FineImages imgs;
    imgs.n = 0;
    imgs.srgb_mask = 0u;
    imgs.table = nullptr;
    for (int i = 0; i < FINE_MAX_IMAGES; i++) { imgs.px[i] = nullptr; imgs.w[i] = 0; imgs.h[i] = 0; }
    if (L.n_images > FINE_MAX_IMAGES) {
        if (!L.image_table) return -1;  // the dispatcher builds the table for arrays that do not fit in the arguments
        imgs.table = L.image_table;
        imgs.n = L.n_images;
    } else {
        for (int i = 0; i < L.n_images; i++) {
            imgs.px[i] = (const uint8_t*)L.images[i].ptr;
            imgs.w[i] = L.images[i].width;
            if (L.images[i].format == JL_RGBA8_SRGB) imgs.srgb_mask |= 1u << i;
            imgs.h[i] = L.images[i].height;
            imgs.n = i + 1;
        }
    }
    uint32_t grad_h = (grad.ptr && grad.width == JL_GRADIENT_WIDTH) ? grad.height : 0u;
    // Scenes without clip layers (ConfigUniform.n_clip == 0, read from the host shadow of the uploaded uniform)
    // use the variant without the 64-register blend stack: higher occupancy.
    bool clips = !(L.cfg_host && L.cfg_host->layout.n_clip == 0u);
    // Without ramps and images every gradient/image texel is 0; the instantiation without that code needs 78
    // instead of 109 VGPRs (6 instead of 4 waves per SIMD; the kernel is latency-bound).
    bool paints = grad_h != 0u;
    for (int i = 0; i < L.n_images; i++) paints = paints || L.images[i].ptr != nullptr;
    // band mode: tile rows of the context's bin rows (a bin row = JL_N_TILE_Y tile rows)
    const uint64_t tr0 = (uint64_t)L.band_row0 * JL_N_TILE_Y, tr1 = (uint64_t)L.band_row1 * JL_N_TILE_Y;
    const uint32_t trow0 = tr0 < L.gy ? (uint32_t)tr0 : L.gy, trow1 = tr1 < L.gy ? (uint32_t)tr1 : L.gy;
    if (trow1 <= trow0) return 0;
    const float* seg_ptr = (segments_n != 0u && L.b[1].ptr) ? (const float*)L.b[1].ptr : (const float*)cfg;  // see load_segraw_clamped
    if (seg_ptr == (const float*)cfg) segments_n = 0u;
    FineCfg fc;
    std::memset(&fc, 0, sizeof fc);
    if (L.cfg_host) {
        fc.valid = 1u;
        fc.width_in_tiles = L.cfg_host->width_in_tiles;
        for (int i = 0; i < 4; i++) fc.base_color[i] = L.cfg_host->base_color[i];
    }
    // Blend-stack levels behind the one in LDS and in front of blend_spill live in a per-tile slice of a scratch array, 4 KiB per
    // level and tile.  How many levels a frame can need follows from the nesting depth of its clip layers, which the caller
    // knows (jh_set_clip_depth_hint: the engine shims count it off the draw tags): none for depth <= 1, one for depth 2, ...;
    // without a hint the worst case, 3 levels = 12 KiB per tile (768 MiB at 4096^2 -- what round 3 always reserved).
    float4* clip_scratch = nullptr;
    uint32_t scr_levels = 0u;
    if (clips) {
        scr_levels = FINE_SCR_LEVELS;
        if (L.clip_depth_hint != 0u) scr_levels = L.clip_depth_hint > FINE_LDS_LEVELS ? std::min<uint32_t>(L.clip_depth_hint - FINE_LDS_LEVELS, FINE_SCR_LEVELS) : 0u;
        clip_scratch = (float4*)jh_scratch_get(L.scratch, JH_SCR_D, (uint64_t)L.gx * (trow1 - trow0) * scr_levels * 4096u);
        if (!clip_scratch) return -5;
    }
#define JH_FINE_LAUNCH(A, C, P)                                                                                                          \
    hipLaunchKernelGGL((k_fine_area<A, C, P>), dim3((L.gx + FINE_WG_WAVES(C) - 1) / FINE_WG_WAVES(C), trow1 - trow0), dim3(64 * FINE_WG_WAVES(C)), 0, L.stream, cfg, fc, seg_ptr, \
                       segments_n, (const uint32_t*)L.b[2].ptr, ptcl_n, (const uint32_t*)L.b[3].ptr, info_n, spill, (uint16_t*)out.ptr,         \
                       out.width, out.height, (const uint16_t*)grad.ptr, grad_h, imgs, L.gx, mask_lut, mask_lut_n, trow0, clip_scratch, scr_levels, L.hint_overflow)
#define JH_FINE_PICK(A)                                  \
    do {                                                 \
        if (clips && paints) JH_FINE_LAUNCH(A, true, true);   \
        else if (clips) JH_FINE_LAUNCH(A, true, false);       \
        else if (paints) JH_FINE_LAUNCH(A, false, true);      \
        else JH_FINE_LAUNCH(A, false, false);                 \
    } while (0)
    if (aa == 8) JH_FINE_PICK(8);
    else if (aa == 16) JH_FINE_PICK(16);
    else JH_FINE_PICK(0);
#undef JH_FINE_PICK
#undef JH_FINE_LAUNCH
    return 0;
}

int jh_launch_fine_area(const JhLaunch& L) { return launch_fine(L, 0); }
int jh_launch_fine_msaa(const JhLaunch& L, int samples) { return launch_fine(L, samples); }
