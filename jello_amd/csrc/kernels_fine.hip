// kernels_fine.hip -- K19 fine_area (orig/fine.wgsl:824-878 fill_path, :883-1103 main,
// shared/blend.wgsl): per 16x16 tile, interpret the PTCL, accumulate analytic-area coverage from
// the tile's segments, composite colours / gradients / images / clip-blend groups, and store
// un-premultiplied RGBA16F.
//
// MI355X design: one wave64 per tile (64 lanes x 4 horizontally adjacent pixels, exactly the
// WGSL's (4,16) workgroup).  The PTCL stream and the segment records are the same for all 64 lanes, so
// each datum is fetched ONCE per tile with wide coalesced loads (the algorithmic-bytes model of the
// roofline) and then broadcast on-chip:
//   * PTCL: the 64-word head, then each 256-word chunk (one dwordx4 per lane = 1 KiB per wave
//     instruction) is staged in a wave-private 1 KiB LDS window; the interpreter reads command words
//     with uniform ds_reads, the command index stays in SGPRs (readfirstlane);
//   * segments: coarse allocates a tile's segment slices back to back, so lane i keeps segment
//     base+i of a 64-segment window in registers (plus a prefetched next window) and the per-segment
//     loop broadcasts the 5 floats with v_readlane -- no dependent memory latency per segment.
// Pixels leave as two 16-byte stores per lane (4 px x RGBA16F = 32 B; 4 lanes cover one 128-B row).
// The 4-deep clip/blend stack lives in registers (statically indexed), deeper levels spill to
// blend_spill exactly like the WGSL.
#include "kcommon.h"

using namespace jk;
using namespace jd;

namespace {

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
JD V4 blend_mix_compose(V4 backdrop, V4 src, uint32_t mode) {  // blend.wgsl:288-310
    const float EPSILON = 1e-15f;
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

#define FINE_MAX_IMAGES 8
struct FineImages {
    const uint8_t* px[FINE_MAX_IMAGES];
    uint32_t w[FINE_MAX_IMAGES], h[FINE_MAX_IMAGES];
    int n;
};

JD uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// rgba = rgba * (1 - fg.a*area) + fg*area  (fine.wgsl:923-926 and the gradient/image arms)
JD V4 over(V4 bg, V4 fg, float area) {
    V4 fg_i = v4(fg.x * area, fg.y * area, fg.z * area, fg.w * area);
    float k = 1.0f - fg_i.w;
    return v4(bg.x * k + fg_i.x, bg.y * k + fg_i.y, bg.z * k + fg_i.z, bg.w * k + fg_i.w);
}

struct SegWin {  // lane i holds segment base+i, plus the per-segment (pixel-independent) terms of fill_path
    float p0x, p0y, p1x, p1y, ye;
    float recip, sgn;  // 1 / delta.y and sign(delta.x), computed once per segment instead of once per lane
};
JD SegWin load_segwin(const float* __restrict__ segments, uint32_t segments_n, uint32_t base) {
    SegWin w;
    uint32_t so = base + (threadIdx.x & 63u);
    w.p0x = 0.0f; w.p0y = 0.0f; w.p1x = 0.0f; w.p1y = 0.0f; w.ye = 0.0f;
    if (so < segments_n) {
        const float2* sp = (const float2*)(segments + (size_t)so * 6);
        float2 a = sp[0], b = sp[1], c = sp[2];
        w.p0x = a.x; w.p0y = a.y; w.p1x = b.x; w.p1y = b.y; w.ye = c.x;
    }
    w.recip = 1.0f / (w.p1y - w.p0y);
    w.sgn = sign_(w.p1x - w.p0x);
    return w;
}
JD float bcast(float v, uint32_t lane) { return u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(v), (int)lane)); }

// Pixel ownership inside the wave: lane = (r, c) with r = lane >> 3, c = lane & 7 owns four pixels,
//   k = 0,1: (row r,     columns 2c, 2c+1)      k = 2,3: (row r + 8, columns 2c, 2c+1).
// A segment clipped to a 16x16 tile usually spans only a few rows, so the coverage code runs per
// half-tile (8 rows x 16 px = all 64 lanes x 2 px) and a half none of whose rows the segment crosses
// is skipped with one uniform branch -- the WGSL's (4 px x 16 rows) mapping keeps most lanes idle.
// Per-pixel arithmetic is exactly the WGSL's: a pixel in column X belongs to the WGSL invocation
// lx = X >> 2 with i = X & 3, so x offsets are formed as (p.x - 4*lx) - i, etc.
template <bool CLIPS>
__global__ __launch_bounds__(64) void k_fine_area(const JlConfig* __restrict__ cfg, const float* __restrict__ segments, uint32_t segments_n,
                                                  const uint32_t* __restrict__ ptcl, uint32_t ptcl_n, const uint32_t* __restrict__ info,
                                                  uint32_t info_n, Buf<V4> blend_spill, uint16_t* __restrict__ output, uint32_t out_w,
                                                  uint32_t out_h, const uint16_t* __restrict__ gradients, uint32_t grad_h, FineImages images) {
    __shared__ uint32_t win[JL_PTCL_INCREMENT];  // wave-private PTCL window
    if (ptcl_n == 0u || ptcl[0] == ~0u) return;  // fine.wgsl:889-893
    const uint32_t lane = threadIdx.x;
    const uint32_t pr = lane >> 3, pc = lane & 7u;
    const uint32_t tile_ix = blockIdx.y * cfg->width_in_tiles + blockIdx.x;
    // per-pixel constants (k = 0..3)
    const uint32_t X0 = 2u * pc;                       // column of k = 0,2; k = 1,3 are X0 + 1
    const float lxb = (float)((X0 >> 2) * 4u);         // WGSL local_xy.x of the owning invocation
    const float i0_f = (float)(X0 & 3u);               // WGSL i of the left pixel (0 or 2); right pixel is i0 + 1
    const float xyx = (float)((blockIdx.x * 4u + (X0 >> 2)) * 4u);  // WGSL xy.x = f32(global_id.x * 4)
    const float lrow[2] = {(float)pr, (float)(pr + 8u)};            // WGSL local_xy.y per half
    const float grow[2] = {(float)(blockIdx.y * 16u + pr), (float)(blockIdx.y * 16u + pr + 8u)};  // WGSL xy.y per half
    V4 rgba[4];
#pragma unroll
    for (int k = 0; k < 4; k++) rgba[k] = v4(cfg->base_color[0], cfg->base_color[1], cfg->base_color[2], cfg->base_color[3]);
    V4 bs0[4], bs1[4], bs2[4], bs3[4];  // blend_stack[0..3]
#pragma unroll
    for (int k = 0; k < 4; k++) { bs0[k] = v4(0, 0, 0, 0); bs1[k] = bs0[k]; bs2[k] = bs0[k]; bs3[k] = bs0[k]; }
    uint32_t clip_depth = 0u;
    float area[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // PTCL window: words [win_base, win_base + 256) of the global stream live in `win`
    uint32_t win_base = tile_ix * JL_PTCL_INITIAL_ALLOC;
    {
        uint32_t gi = win_base + lane;
        win[lane] = gi < ptcl_n ? ptcl[gi] : 0u;
    }
    __syncthreads();
    uint32_t cmd_ix = 0u;  // relative to win_base
    auto P = [&](uint32_t rel) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)win[rel & (JL_PTCL_INCREMENT - 1u)]); };
    auto I = [&](uint32_t i) -> uint32_t { return i < info_n ? info[i] : 0u; };
    const uint32_t blend_offset = P(cmd_ix);
    cmd_ix += 1u;
    auto load_grad = [&](int32_t x, uint32_t y) -> V4 {
        if (x < 0 || x >= JL_GRADIENT_WIDTH || y >= grad_h) return v4(0, 0, 0, 0);
        const uint16_t* t = gradients + ((size_t)y * JL_GRADIENT_WIDTH + (size_t)x) * 4;
        uint2 raw = *(const uint2*)t;
        return v4(f16_to_f32((uint16_t)(raw.x & 0xffffu)), f16_to_f32((uint16_t)(raw.x >> 16)), f16_to_f32((uint16_t)(raw.y & 0xffffu)),
                  f16_to_f32((uint16_t)(raw.y >> 16)));
    };
    // pixel k: WGSL i (as float) and spill index inside the tile
    auto pix_i = [&](int k) -> float { return i0_f + (float)(k & 1); };
    auto pix_spill = [&](int k) -> uint32_t { return (pr + 8u * (uint32_t)(k >> 1)) * JL_TILE_WIDTH + X0 + (uint32_t)(k & 1); };
    // segment windows
    SegWin cur, nxt;
    cur.p0x = cur.p0y = cur.p1x = cur.p1y = cur.ye = cur.recip = cur.sgn = 0.0f;
    nxt = cur;
    uint32_t cur_base = 0xffffffffu, nxt_base = 0xffffffffu;  // "no window"
    for (uint32_t guard = 0; guard < (1u << 24); guard++) {
        uint32_t tag = P(cmd_ix);
        if (tag == JL_CMD_END) break;
        if (tag == JL_CMD_FILL) {  // fill_path, fine.wgsl:824-878
            uint32_t size_and_rule = P(cmd_ix + 1u);
            uint32_t seg_data = P(cmd_ix + 2u);
            int32_t backdrop = (int32_t)P(cmd_ix + 3u);
            uint32_t n_segs = size_and_rule >> 1;
            bool even_odd = (size_and_rule & 1u) != 0u;
            float backdrop_f = (float)backdrop;
#pragma unroll
            for (int k = 0; k < 4; k++) area[k] = backdrop_f;
            for (uint32_t s = 0; s < n_segs; s++) {
                uint32_t so = seg_data + s;
                uint32_t rel = so - cur_base;
                if (cur_base == 0xffffffffu || rel >= 64u) {  // uniform: advance / reload the window
                    if (nxt_base != 0xffffffffu && so - nxt_base < 64u) {
                        cur = nxt;
                        cur_base = nxt_base;
                    } else {
                        cur = load_segwin(segments, segments_n, so);
                        cur_base = so;
                    }
                    nxt_base = cur_base + 64u;
                    nxt = load_segwin(segments, segments_n, nxt_base);  // prefetch; consumed much later
                    rel = so - cur_base;
                }
                float p0x = bcast(cur.p0x, rel), p0y = bcast(cur.p0y, rel), p1x = bcast(cur.p1x, rel), p1y = bcast(cur.p1y, rel);
                float y_edge_v = bcast(cur.ye, rel);
                float vec_y_recip = bcast(cur.recip, rel), sgn_dlx = bcast(cur.sgn, rel);
                float dlx = p1x - p0x, dly = p1y - p0y;
                float startx = p0x - lxb;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    float y = p0y - lrow[h];
                    float y0 = clamp_(y, 0.0f, 1.0f);
                    float y1 = clamp_(y + dly, 0.0f, 1.0f);
                    float dy = y0 - y1;
                    if (dy != 0.0f) {
                        float t0 = (y0 - y) * vec_y_recip;
                        float t1 = (y1 - y) * vec_y_recip;
                        float x0 = startx + t0 * dlx;
                        float x1 = startx + t1 * dlx;
                        float xmin0 = fmin_(x0, x1);
                        float xmax0 = fmax_(x0, x1);
#pragma unroll
                        for (int e = 0; e < 2; e++) {
                            float i_f = i0_f + (float)e;
                            float xmin = fmin_(xmin0 - i_f, 1.0f) - 1.0e-6f;
                            float xmax = xmax0 - i_f;
                            float b = fmin_(xmax, 1.0f);
                            float c = fmax_(b, 0.0f);
                            float d = fmax_(xmin, 0.0f);
                            float a = (b + 0.5f * (d * d - c * c) - xmin) / (xmax - xmin);
                            area[2 * h + e] += a * dy;
                        }
                    }
                }
                // y_edge >= 16 (path_tiling's "no edge" value is 1e9) clamps to 0 for every row of the tile: the term
                // would add +-0, which cannot change a sum that started from +0 -- skip it (uniform branch).
                if (y_edge_v < 16.0f) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        float y_edge = sgn_dlx * clamp_(lrow[h] - y_edge_v + 1.0f, 0.0f, 1.0f);
                        area[2 * h] += y_edge;
                        area[2 * h + 1] += y_edge;
                    }
                }
            }
            if (even_odd) {
#pragma unroll
                for (int k = 0; k < 4; k++) { float a = area[k]; area[k] = abs_(a - 2.0f * round_(0.5f * a)); }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) area[k] = fmin_(abs_(area[k]), 1.0f);
            }
            cmd_ix += 4u;
        } else if (tag == JL_CMD_SOLID) {
#pragma unroll
            for (int k = 0; k < 4; k++) area[k] = 1.0f;
            cmd_ix += 1u;
        } else if (tag == JL_CMD_COLOR) {
            V4 fg = v4(u2f(P(cmd_ix + 1u)), u2f(P(cmd_ix + 2u)), u2f(P(cmd_ix + 3u)), u2f(P(cmd_ix + 4u)));
#pragma unroll
            for (int k = 0; k < 4; k++) rgba[k] = over(rgba[k], fg, area[k]);
            cmd_ix += 5u;
        } else if (CLIPS && tag == JL_CMD_BEGIN_CLIP) {
            if (clip_depth < JL_BLEND_STACK_SPLIT) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    switch (clip_depth) {
                        case 0: bs0[k] = rgba[k]; break;
                        case 1: bs1[k] = rgba[k]; break;
                        case 2: bs2[k] = rgba[k]; break;
                        default: bs3[k] = rgba[k]; break;
                    }
                    rgba[k] = v4(0, 0, 0, 0);
                }
            } else {
                uint32_t blend_in_scratch = clip_depth - JL_BLEND_STACK_SPLIT;
                uint32_t spill_base = blend_offset + blend_in_scratch * JL_TILE_WIDTH * JL_TILE_HEIGHT;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    blend_spill.wr(spill_base + pix_spill(k), rgba[k]);
                    rgba[k] = v4(0, 0, 0, 0);
                }
            }
            clip_depth += 1u;
            cmd_ix += 1u;
        } else if (CLIPS && tag == JL_CMD_END_CLIP) {
            uint32_t blend = P(cmd_ix + 1u);
            float alpha = u2f(P(cmd_ix + 2u));
            clip_depth -= 1u;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                V4 bg;
                if (clip_depth < JL_BLEND_STACK_SPLIT) {
                    switch (clip_depth) {
                        case 0: bg = bs0[k]; break;
                        case 1: bg = bs1[k]; break;
                        case 2: bg = bs2[k]; break;
                        default: bg = bs3[k]; break;
                    }
                } else {
                    uint32_t blend_in_scratch = clip_depth - JL_BLEND_STACK_SPLIT;
                    uint32_t spill_base = blend_offset + blend_in_scratch * JL_TILE_WIDTH * JL_TILE_HEIGHT;
                    bg = blend_spill.rd(spill_base + pix_spill(k));
                }
                V4 fg = v4(rgba[k].x * area[k] * alpha, rgba[k].y * area[k] * alpha, rgba[k].z * area[k] * alpha, rgba[k].w * area[k] * alpha);
                rgba[k] = blend_mix_compose(bg, fg, blend);
            }
            cmd_ix += 3u;
        } else if (tag == JL_CMD_JUMP) {
            win_base = P(cmd_ix + 1u);
            cmd_ix = 0u;
            __syncthreads();  // everyone is done reading the old window
            {
                uint32_t gi = win_base + lane * 4u;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (gi + 3u < ptcl_n && (win_base & 3u) == 0u) {
                    v = *(const uint4*)(ptcl + gi);
                } else {
                    if (gi < ptcl_n) v.x = ptcl[gi];
                    if (gi + 1u < ptcl_n) v.y = ptcl[gi + 1u];
                    if (gi + 2u < ptcl_n) v.z = ptcl[gi + 2u];
                    if (gi + 3u < ptcl_n) v.w = ptcl[gi + 3u];
                }
                *(uint4*)(&win[lane * 4u]) = v;
            }
            __syncthreads();
        } else if (tag == JL_CMD_LIN_GRAD) {
            uint32_t index_mode = P(cmd_ix + 1u);
            uint32_t index = index_mode >> 2, ext = index_mode & 3u;
            uint32_t io = P(cmd_ix + 2u);
            float line_x = u2f(I(io)), line_y = u2f(I(io + 1u)), line_c = u2f(I(io + 2u));
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float d = line_x * xyx + line_y * grow[k >> 1] + line_c;
                float my_d = d + line_x * pix_i(k);
                int32_t x = to_i32(round_(extend_mode(my_d, ext) * 511.0f));
                rgba[k] = over(rgba[k], load_grad(x, index), area[k]);
            }
            cmd_ix += 3u;
        } else if (tag == JL_CMD_RAD_GRAD) {
            uint32_t index_mode = P(cmd_ix + 1u);
            uint32_t index = index_mode >> 2, ext = index_mode & 3u;
            uint32_t io = P(cmd_ix + 2u);
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
                float mx = xyx + pix_i(k), my = grow[k >> 1];
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
            cmd_ix += 3u;
        } else if (tag == JL_CMD_SWEEP_GRAD) {
            uint32_t index_mode = P(cmd_ix + 1u);
            uint32_t index = index_mode >> 2, ext = index_mode & 3u;
            uint32_t io = P(cmd_ix + 2u);
            float m0 = u2f(I(io)), m1 = u2f(I(io + 1u)), m2 = u2f(I(io + 2u)), m3 = u2f(I(io + 3u));
            float xl0 = u2f(I(io + 4u)), xl1 = u2f(I(io + 5u));
            float t0 = u2f(I(io + 6u)), t1 = u2f(I(io + 7u));
            float scale = 1.0f / (t1 - t0);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float mx = xyx + pix_i(k), my = grow[k >> 1];
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
            cmd_ix += 3u;
        } else if (tag == JL_CMD_IMAGE) {
            uint32_t io = P(cmd_ix + 1u);
            float m0 = u2f(I(io)), m1 = u2f(I(io + 1u)), m2 = u2f(I(io + 2u)), m3 = u2f(I(io + 3u));
            float xl0 = u2f(I(io + 4u)), xl1 = u2f(I(io + 5u));
            uint32_t index = I(io + 6u);
            uint32_t width_height = I(io + 7u);
            float ew = (float)(width_height >> 16), eh = (float)(width_height & 0xffffu);
            const uint8_t* ipx = nullptr;
            uint32_t iw = 0, ih = 0;
#pragma unroll
            for (int q = 0; q < FINE_MAX_IMAGES; q++)
                if ((uint32_t)q == index && q < images.n) { ipx = images.px[q]; iw = images.w[q]; ih = images.h[q]; }
            auto texel = [&](int32_t tx, int32_t ty) -> V4 {
                if (!ipx || tx < 0 || ty < 0 || (uint32_t)tx >= iw || (uint32_t)ty >= ih) return v4(0, 0, 0, 0);
                uint32_t raw = *(const uint32_t*)(ipx + ((size_t)ty * iw + (size_t)tx) * 4);
                float r = (float)(raw & 0xffu) / 255.0f, g = (float)((raw >> 8) & 0xffu) / 255.0f, b = (float)((raw >> 16) & 0xffu) / 255.0f,
                      a = (float)(raw >> 24) / 255.0f;
                return v4(r * a, g * a, b * a, a);  // premul_alpha, fine.wgsl:1105-1107
            };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float mx = xyx + pix_i(k), my = grow[k >> 1];
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
            cmd_ix += 2u;
        } else {
            break;  // unknown tag: the WGSL would never advance; stop instead of hanging the GPU
        }
    }
    // fine.wgsl:1092-1102: un-premultiply, store RGBA16F (two adjacent pixels = 16 bytes per lane and half)
    const uint32_t cx0 = blockIdx.x * 16u + X0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        uint32_t cy = blockIdx.y * 16u + pr + 8u * (uint32_t)h;
        if (cy >= out_h) continue;
        uint32_t packed[4];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            V4 fg = rgba[2 * h + e];
            float a_inv = 1.0f / fmax_(fg.w, 1e-6f);
            uint32_t r = f32_to_f16(fg.x * a_inv), g = f32_to_f16(fg.y * a_inv), b = f32_to_f16(fg.z * a_inv), a = f32_to_f16(fg.w);
            packed[e * 2] = r | (g << 16);
            packed[e * 2 + 1] = b | (a << 16);
        }
        uint16_t* row = output + ((size_t)cy * out_w + cx0) * 4;
        if (cx0 + 1u < out_w && ((out_w & 1u) == 0u)) {
            *(uint4*)row = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        } else {
            if (cx0 < out_w) *(uint2*)row = make_uint2(packed[0], packed[1]);
            if (cx0 + 1u < out_w) *(uint2*)(row + 4) = make_uint2(packed[2], packed[3]);
        }
    }
}

}  // namespace

// [config, segments, ptcl, info, blend_spill, output image, gradients image, images[]]
int jh_launch_fine_area(const JhLaunch& L) {
    if (L.nb < 7) return -1;
    if (L.gx == 0 || L.gy == 0) return 0;
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
    for (int i = 0; i < FINE_MAX_IMAGES; i++) { imgs.px[i] = nullptr; imgs.w[i] = 0; imgs.h[i] = 0; }
    for (int i = 0; i < L.n_images && i < FINE_MAX_IMAGES; i++) {
        imgs.px[i] = (const uint8_t*)L.images[i].ptr;
        imgs.w[i] = L.images[i].width;
        imgs.h[i] = L.images[i].height;
        imgs.n = i + 1;
    }
    uint32_t grad_h = (grad.ptr && grad.width == JL_GRADIENT_WIDTH) ? grad.height : 0u;
    // Scenes without clip layers (ConfigUniform.n_clip == 0, read from the host shadow of the uploaded uniform)
    // use the variant without the 64-register blend stack: higher occupancy.
    bool clips = !(L.cfg_host && L.cfg_host->layout.n_clip == 0u);
    if (clips)
        hipLaunchKernelGGL(k_fine_area<true>, dim3(L.gx, L.gy), dim3(64), 0, L.stream, cfg, (const float*)L.b[1].ptr, segments_n,
                           (const uint32_t*)L.b[2].ptr, ptcl_n, (const uint32_t*)L.b[3].ptr, info_n, spill, (uint16_t*)out.ptr, out.width,
                           out.height, (const uint16_t*)grad.ptr, grad_h, imgs);
    else
        hipLaunchKernelGGL(k_fine_area<false>, dim3(L.gx, L.gy), dim3(64), 0, L.stream, cfg, (const float*)L.b[1].ptr, segments_n,
                           (const uint32_t*)L.b[2].ptr, ptcl_n, (const uint32_t*)L.b[3].ptr, info_n, spill, (uint16_t*)out.ptr, out.width,
                           out.height, (const uint16_t*)grad.ptr, grad_h, imgs);
    return 0;
}
