// kernels_tile.hip -- binning, tile allocation, path_count / path_tiling ("path_coarse"), backdrop:
//   K11 binning           orig/binning.wgsl:58-184
//   K12 tile_alloc        orig/tile_alloc.wgsl:35-123
//   K13 path_count_setup  orig/path_count_setup.wgsl:17-27
//   K14 path_count        orig/path_count.wgsl:51-202
//   K15 backdrop_dyn      orig/backdrop_dyn.wgsl:28-86
//   K17 path_tiling_setup orig/path_tiling_setup.wgsl:20-32
//   K18 path_tiling       orig/path_tiling.wgsl:39-173
//
// MI355X design: every bump `atomicAdd` of the WGSL becomes count -> exclusive scan -> write, in the
// reference's sequential order, so all outputs are run-to-run identical:
//   binning     per-(workgroup,bin) element counts -> scan (wg-major) -> chunk offsets;
//   tile_alloc  per-draw-object tile counts -> scan -> Path.tiles;
//   path_count  per-line crossing counts -> scan -> SegmentCount slots; per-tile counts by
//               no-return atomics (a sum, order-free); the per-tile arrival rank
//               `seg_within_slice` is computed afterwards as the rank of the crossing's global
//               index inside its tile's list (pc_scatter_part in k_pc_rank_small's launch / k_pc_rank); the value the count atomic
//               happened to return is only used as a unique slot inside that temporary list.
// The WGSL indirect dispatches become grid-stride loops bounded by the IndirectCount the setup
// kernels write, so no host readback is needed.  All of this is HBM/atomic-bound integer work.
#include <algorithm>

#include "kcommon.h"

using namespace jk;
using namespace jd;

namespace {

struct Bb4 { float v[4]; };

JD void bbox_intersect(const float* a, const float* b, float* o) {  // shared/bbox.wgsl:21-23
    float r0 = fmax_(a[0], b[0]), r1 = fmax_(a[1], b[1]), r2 = fmin_(a[2], b[2]), r3 = fmin_(a[3], b[3]);
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}

// ------------------------------------------------------------------------------------------------
// binning.  PASS 0: counts (+ intersected bbox).  PASS 1: chunk offsets known -> headers + bin_data.
// ------------------------------------------------------------------------------------------------
template <int PASS>
__global__ __launch_bounds__(JL_WG) void k_binning(const JlConfig* __restrict__ cfg, Buf<JlDrawMonoid> draw_monoids, Buf<JlPathBbox> path_bbox_buf,
                                                   Buf<Bb4> clip_bbox_buf, Buf<Bb4> intersected_bbox, JlBump* __restrict__ bump,
                                                   Buf<uint32_t> bin_data, Buf<JlBinHeader> bin_header, uint32_t* __restrict__ wg_tot) {
    // PASS 0 leaves the workgroup's total element count in wg_tot[workgroup]; PASS 1 derives its chunk offsets from the
    // totals of the workgroups before it and a prefix over its own bins (the canonical order is workgroup-major, then
    // bin), so no scan launch sits between the two passes.
    __shared__ uint32_t sh_bitmaps[8][JL_N_TILE];
    __shared__ uint32_t sh[8];
    const float SX = 0.00390625f, SY = 0.00390625f;
    uint32_t lid = threadIdx.x;
    uint32_t gid = blockIdx.x * JL_WG + lid;
    for (uint32_t i = 0; i < 8u; i++) sh_bitmaps[i][lid] = 0u;
    if (bump->lines > cfg->lines_size) {  // binning.wgsl:67-78 (uniform: read of a value written by an earlier stage)
        if (PASS == 1 && gid == 0u) atomicOr(&bump->failed, (uint32_t)JL_STAGE_FLATTEN);
        if (PASS == 0 && lid == 0u) wg_tot[blockIdx.x] = 0u;
        return;
    }
    __syncthreads();
    uint32_t element_ix = gid;
    int32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (element_ix < cfg->layout.n_drawobj) {
        JlDrawMonoid dm = draw_monoids.rd(element_ix);
        float clip_bbox[4] = {-1e9f, -1e9f, 1e9f, 1e9f};
        if (dm.clip_ix > 0u) {
            Bb4 cb = clip_bbox_buf.rd(umin_(dm.clip_ix - 1u, cfg->layout.n_clip - 1u));
            clip_bbox[0] = cb.v[0]; clip_bbox[1] = cb.v[1]; clip_bbox[2] = cb.v[2]; clip_bbox[3] = cb.v[3];
        }
        JlPathBbox pb = path_bbox_buf.rd(dm.path_ix);
        float pbf[4] = {(float)pb.x0, (float)pb.y0, (float)pb.x1, (float)pb.y1};
        Bb4 bbox;
        bbox_intersect(clip_bbox, pbf, bbox.v);
        if (PASS == 0) intersected_bbox.wr(element_ix, bbox);
        if (bbox.v[0] < bbox.v[2] && bbox.v[1] < bbox.v[3]) {
            x0 = to_i32(floor_(bbox.v[0] * SX));
            y0 = to_i32(floor_(bbox.v[1] * SY));
            x1 = to_i32(ceil_(bbox.v[2] * SX));
            y1 = to_i32(ceil_(bbox.v[3] * SY));
        }
    }
    int32_t width_in_bins = (int32_t)((cfg->width_in_tiles + 15u) / 16u);
    int32_t height_in_bins = (int32_t)((cfg->height_in_tiles + 15u) / 16u);
    x0 = iclamp_(x0, 0, width_in_bins);
    y0 = iclamp_(y0, 0, height_in_bins);
    x1 = iclamp_(x1, 0, width_in_bins);
    y1 = iclamp_(y1, 0, height_in_bins);
    if (x0 == x1) y1 = y0;
    int32_t x = x0, y = y0;
    uint32_t my_slice = lid / 32u;
    uint32_t my_mask = 1u << (lid & 31u);
    while (y < y1) {
        uint32_t bin = (uint32_t)(y * width_in_bins + x);
        if (bin < JL_N_TILE) atomicOr(&sh_bitmaps[my_slice][bin], my_mask);
        x += 1;
        if (x == x1) { x = x0; y += 1; }
    }
    __syncthreads();
    uint32_t element_count = 0u;
    for (uint32_t i = 0; i < 8u; i++) element_count += __popc(sh_bitmaps[i][lid]);
    if (PASS == 0) {
        MonoidK<1> m;
        m.v[0] = element_count;
        const uint32_t tot = block_reduce_monoid<1>(m, sh).v[0];
        if (lid == 0u) wg_tot[blockIdx.x] = tot;
        return;
    }
    MonoidK<2> part;  // elements of the workgroups before mine / of all workgroups (bump.binning)
    part.v[0] = 0u; part.v[1] = 0u;
    for (uint32_t j = lid; j < gridDim.x; j += JL_WG) {
        const uint32_t v = wg_tot[j];
        part.v[1] += v;
        if (j < blockIdx.x) part.v[0] += v;
    }
    const MonoidK<2> sums = block_reduce_monoid<2>(part, sh);
    __syncthreads();
    if (gid == 0u) bump->binning = sums.v[1];
    uint32_t wg_elements;
    uint32_t chunk_offset = sums.v[0] + block_excl_scan_u32(element_count, sh, &wg_elements);
    if (chunk_offset + element_count > cfg->binning_size) {
        chunk_offset = 0u;
        atomicOr(&bump->failed, (uint32_t)JL_STAGE_BINNING);
    }
    JlBinHeader h;
    h.element_count = element_count;
    h.chunk_offset = chunk_offset;
    bin_header.wr(gid, h);
    // binning.wgsl:144-165 lets every element walk its bins again and write itself at
    // (set bits below it in the bin's bitmaps); thread = bin writing its set bits in ascending order produces the same
    // list without the 256-step walk of an element that covers the whole target (a background rectangle).
    uint32_t out = cfg->layout.bin_data_start + chunk_offset;
    for (uint32_t i = 0; i < 8u; i++) {
        uint32_t bits = sh_bitmaps[i][lid];
        while (bits != 0u) {
            bin_data.wr(out, blockIdx.x * JL_WG + i * 32u + (uint32_t)__builtin_ctz(bits));
            out += 1u;
            bits &= bits - 1u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// tile_alloc.  PASS 0: bbox + tile counts.  PASS 1: offsets known -> Path.tiles, zero the tiles.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(JL_WG) void k_tile_alloc_count(const JlConfig* __restrict__ cfg, Buf<uint32_t> scene, Buf<Bb4> draw_bboxes,
                                                            const JlBump* __restrict__ bump, Buf<JlPath> paths, uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ wg_tot) {
    __shared__ uint32_t sh[8];
    uint32_t drawobj_ix = blockIdx.x * JL_WG + threadIdx.x;
    if ((bump->failed & (JL_STAGE_BINNING | JL_STAGE_FLATTEN)) != 0u) {  // tile_alloc.wgsl:43-50
        counts[drawobj_ix] = 0u;
        if (threadIdx.x == 0u) wg_tot[blockIdx.x] = 0u;
        return;
    }
    const float SX = 1.0f / 16.0f, SY = 1.0f / 16.0f;
    uint32_t drawtag = 0u;
    if (drawobj_ix < cfg->layout.n_drawobj) drawtag = scene.rd(cfg->layout.drawtag_base + drawobj_ix);
    int32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (drawtag != JL_DRAWTAG_NOP && drawtag != JL_DRAWTAG_END_CLIP) {
        Bb4 bbox = draw_bboxes.rd(drawobj_ix);
        if (bbox.v[0] < bbox.v[2] && bbox.v[1] < bbox.v[3]) {
            x0 = to_i32(floor_(bbox.v[0] * SX));
            y0 = to_i32(floor_(bbox.v[1] * SY));
            x1 = to_i32(ceil_(bbox.v[2] * SX));
            y1 = to_i32(ceil_(bbox.v[3] * SY));
        }
    }
    uint32_t ux0 = (uint32_t)iclamp_(x0, 0, (int32_t)cfg->width_in_tiles);
    uint32_t uy0 = (uint32_t)iclamp_(y0, 0, (int32_t)cfg->height_in_tiles);
    uint32_t ux1 = (uint32_t)iclamp_(x1, 0, (int32_t)cfg->width_in_tiles);
    uint32_t uy1 = (uint32_t)iclamp_(y1, 0, (int32_t)cfg->height_in_tiles);
    const uint32_t count = (ux1 - ux0) * (uy1 - uy0);
    counts[drawobj_ix] = count;
    if (drawobj_ix < cfg->layout.n_drawobj && paths.ok(drawobj_ix)) {
        JlPath p;
        p.bbox[0] = ux0; p.bbox[1] = uy0; p.bbox[2] = ux1; p.bbox[3] = uy1;
        p.tiles = 0u; p.pad[0] = 0u; p.pad[1] = 0u; p.pad[2] = 0u;
        paths.p[drawobj_ix] = p;
    }
    // the workgroup's total: the write pass turns the totals of the workgroups before it and a prefix inside its own
    // into the offsets (draw-object order = workgroup-major), so no scan launch sits between the two passes
    MonoidK<1> m;
    m.v[0] = count;
    const uint32_t tot = block_reduce_monoid<1>(m, sh).v[0];
    if (threadIdx.x == 0u) wg_tot[blockIdx.x] = tot;
}
JD void tile_zero_part(const JlConfig* __restrict__ cfg, uint32_t total, Buf<JlTile> tiles, uint32_t block, uint32_t blocks);
// Workgroups [0, write_blocks) write Path.tiles, the rest clear the allocated tiles (two jobs without a dependency on
// each other in one launch: a launch of their own costs each of them ~4.5 us).
__global__ __launch_bounds__(JL_WG) void k_tile_alloc_write(const JlConfig* __restrict__ cfg, JlBump* __restrict__ bump, Buf<JlPath> paths,
                                                            Buf<JlTile> tiles, const uint32_t* __restrict__ counts,
                                                            const uint32_t* __restrict__ wg_tot, uint32_t write_blocks) {
    __shared__ uint32_t sh[8];
    if ((bump->failed & (JL_STAGE_BINNING | JL_STAGE_FLATTEN)) != 0u) return;
    // sum of the workgroup totals before mine (and of all of them: bump.tile, tile_alloc.wgsl:90)
    const uint32_t me = blockIdx.x < write_blocks ? blockIdx.x : write_blocks;
    MonoidK<2> part;
    part.v[0] = 0u; part.v[1] = 0u;
    for (uint32_t j = threadIdx.x; j < write_blocks; j += JL_WG) {
        const uint32_t v = wg_tot[j];
        part.v[1] += v;
        if (j < me) part.v[0] += v;
    }
    const MonoidK<2> sums = block_reduce_monoid<2>(part, sh);
    __syncthreads();
    const uint32_t wg_off = sums.v[0], total = sums.v[1];
    if (blockIdx.x >= write_blocks) {  // uniform
        tile_zero_part(cfg, total, tiles, blockIdx.x - write_blocks, gridDim.x - write_blocks);
        return;
    }
    if (blockIdx.x == 0u && threadIdx.x == 0u) bump->tile = total;
    uint32_t wg_first = blockIdx.x * JL_WG;
    uint32_t drawobj_ix = wg_first + threadIdx.x;
    uint32_t wg_cnt;
    uint32_t my_sub = block_excl_scan_u32(counts[drawobj_ix], sh, &wg_cnt);
    uint32_t offset = wg_off;
    if (offset + wg_cnt > cfg->tiles_size) {  // tile_alloc.wgsl:93-99
        offset = 0u;
        if (threadIdx.x == JL_WG - 1u) atomicOr(&bump->failed, (uint32_t)JL_STAGE_TILE_ALLOC);
    }
    if (drawobj_ix < cfg->layout.n_drawobj && paths.ok(drawobj_ix)) paths.p[drawobj_ix].tiles = offset + my_sub;
}
// The WGSL zeroes a workgroup's tiles with that workgroup's 256 threads (tile_alloc.wgsl:107-111): two workgroups
// clearing the 1.7 M tiles of 300 large circles take 0.3 ms.  All allocated tiles form the range [0, bump.tile), so
// one device-wide pass clears them (in the failure case the contents of the buffer are unspecified anyway).
JD void tile_zero_part(const JlConfig* __restrict__ cfg, uint32_t total, Buf<JlTile> tiles, uint32_t block, uint32_t blocks) {
    const uint32_t n = umin_(umin_(total, cfg->tiles_size), tiles.n);
    uint4* p = (uint4*)tiles.p;  // two tiles per store
    const uint32_t n2 = n >> 1;
    for (uint32_t i = block * JL_WG + threadIdx.x; i < n2; i += blocks * JL_WG) p[i] = make_uint4(0u, 0u, 0u, 0u);
    if ((n & 1u) != 0u && block == 0u && threadIdx.x == 0u) {
        JlTile z;
        z.backdrop = 0;
        z.segment_count_or_ix = 0u;
        tiles.p[n - 1u] = z;
    }
}

// ------------------------------------------------------------------------------------------------
// setup kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_path_count_setup(const JlBump* __restrict__ bump, JlIndirectCount* __restrict__ ind) {
    if (bump->failed != 0u) ind->x = 0u; else ind->x = (bump->lines + (JL_WG - 1u)) / JL_WG;
    ind->y = 1u;
    ind->z = 1u;
}
__global__ void k_path_tiling_setup(const JlBump* __restrict__ bump, JlIndirectCount* __restrict__ ind, Buf<uint32_t> ptcl) {
    if (bump->failed != 0u) {
        ind->x = 0u;
        ptcl.wr(0u, ~0u);
    } else {
        ind->x = (bump->seg_counts + (JL_WG - 1u)) / JL_WG;
    }
    ind->y = 1u;
    ind->z = 1u;
}

// ------------------------------------------------------------------------------------------------
// path_count
// ------------------------------------------------------------------------------------------------
JD uint32_t span(float a, float b) { return to_u32(fmax_(ceil_(fmax_(a, b)) - floor_(fmin_(a, b)), 1.0f)); }
#define ONE_MINUS_ULP 0.99999994f
#define ROBUST_EPSILON 2e-7f
#define TILE_SCALE 0.0625f

struct LineSetup {
    bool valid;
    bool is_down, is_positive_slope;
    float a, b, x0, y0, x_sign, s0y;
    uint32_t imin, imax;
    int32_t ymin, ymax, delta;
    int32_t bbox[4];
    int32_t stride;
    uint32_t tiles;
};

// Everything of path_count.wgsl:61-166 that precedes the side effects.
JD LineSetup line_setup(const JlLineSoup& line, const Buf<JlPath>& paths) {
    LineSetup r;
    r.valid = false;
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
    if (dx + dy == 0.0f) return r;
    if (dy == 0.0f && floor_(s0.y) == s0.y) return r;
    float idxdy = 1.0f / (dx + dy);
    float a = dx * idxdy;
    bool is_positive_slope = s1.x >= s0.x;
    float x_sign = is_positive_slope ? 1.0f : -1.0f;
    float xt0 = floor_(s0.x * x_sign);
    float c = s0.x * x_sign - xt0;
    float y0 = floor_(s0.y);
    float ytop = (s0.y == s1.y) ? ceil_(s0.y) : (y0 + 1.0f);
    float b = fmin_((dy * c + dx * (ytop - s0.y)) * idxdy, ONE_MINUS_ULP);
    float robust_err = floor_(a * ((float)count - 1.0f) + b) - (float)count_x;
    if (robust_err != 0.0f) a -= ROBUST_EPSILON * sign_(robust_err);
    float x0 = xt0 * x_sign + (is_positive_slope ? 0.0f : -1.0f);

    JlPath path = paths.rd(line.path_ix);
    int32_t bbox[4] = {(int32_t)path.bbox[0], (int32_t)path.bbox[1], (int32_t)path.bbox[2], (int32_t)path.bbox[3]};
    float xmin = fmin_(s0.x, s1.x);
    int32_t stride = bbox[2] - bbox[0];
    if (s0.y >= (float)bbox[3] || s1.y <= (float)bbox[1] || xmin >= (float)bbox[2] || stride == 0) return r;
    uint32_t imin = 0u;
    if (s0.y < (float)bbox[1]) {
        float iminf = round_(((float)bbox[1] - y0 + b - a) / (1.0f - a)) - 1.0f;
        if (y0 + iminf - floor_(a * iminf + b) < (float)bbox[1]) iminf += 1.0f;
        imin = to_u32(iminf);
    }
    uint32_t imax = count;
    if (s1.y > (float)bbox[3]) {
        float imaxf = round_(((float)bbox[3] - y0 + b - a) / (1.0f - a)) - 1.0f;
        if (y0 + imaxf - floor_(a * imaxf + b) < (float)bbox[3]) imaxf += 1.0f;
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
            float f = round_((x_sign * ((float)bbox[0] - x0) - b + fudge) / a);
            if ((x0 + x_sign * floor_(a * f + b) < (float)bbox[0]) == is_positive_slope) f += 1.0f;
            int32_t ynext = to_i32(y0 + f - floor_(a * f + b) + 1.0f);
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
            float f = round_((x_sign * ((float)bbox[2] - x0) - b + fudge) / a);
            if ((x0 + x_sign * floor_(a * f + b) < (float)bbox[2]) == is_positive_slope) f += 1.0f;
            if (is_positive_slope) imax = umin_(imax, to_u32(f)); else imin = umax_(imin, to_u32(f));
        }
    }
    imax = umax_(imin, imax);
    ymin = imax_(ymin, bbox[1]);
    ymax = imin_(ymax, bbox[3]);
    r.valid = true;
    r.is_down = is_down; r.is_positive_slope = is_positive_slope;
    r.a = a; r.b = b; r.x0 = x0; r.y0 = y0; r.x_sign = x_sign; r.s0y = s0.y;
    r.imin = imin; r.imax = imax; r.ymin = ymin; r.ymax = ymax; r.delta = delta;
    r.bbox[0] = bbox[0]; r.bbox[1] = bbox[1]; r.bbox[2] = bbox[2]; r.bbox[3] = bbox[3];
    r.stride = stride;
    r.tiles = path.tiles;
    return r;
}

// pass 1: crossings per line
__global__ __launch_bounds__(JL_WG) void k_pc_count(const JlBump* __restrict__ bump, const JlIndirectCount* __restrict__ ind, Buf<JlLineSoup> lines,
                                                    Buf<JlPath> paths, uint32_t* __restrict__ counts, uint32_t counts_n,
                                                    uint32_t* __restrict__ zero, uint32_t zero_n, uint32_t* __restrict__ ptotal, uint32_t ptotal_n,
                                                    unsigned long long* __restrict__ bd_ctr,
                                                    JlIndirectCount* __restrict__ setup_out) {  // != nullptr: path_count_setup was held back (jello_hip.cpp,
                                                                                                // Deferred): this kernel does its work (path_count_setup.wgsl:17-27)
    // (the path ranges, the gate and the dense-tile counter of the later passes start from zero: cleared here, this
    // kernel does not use them, instead of by a separate fill launch; likewise the wide-row counter of the backdrop
    // stage that follows: kcommon.h, JH_CLEAN_*)
    for (uint32_t i = blockIdx.x * JL_WG + threadIdx.x; i < zero_n; i += gridDim.x * JL_WG) zero[i] = 0u;
    if (blockIdx.x == 0u && threadIdx.x == 0u) *bd_ctr = 0ull;
    uint32_t n_lines = umin_(bump->lines, counts_n);
    uint32_t ind_x;
    if (setup_out) {  // (every thread derives the count itself; one of them also stores it for the kernels that follow)
        ind_x = bump->failed != 0u ? 0u : (bump->lines + (JL_WG - 1u)) / JL_WG;
        if (blockIdx.x == 0u && threadIdx.x == 0u) { setup_out->x = ind_x; setup_out->y = 1u; setup_out->z = 1u; }
    } else {
        ind_x = ind->x;
    }
    uint32_t n_threads = umin_(ind_x * JL_WG, counts_n);
    const uint32_t lane = lane_id();
    for (uint32_t g0 = blockIdx.x * JL_WG + (threadIdx.x & ~63u); g0 < n_lines; g0 += gridDim.x * JL_WG) {  // uniform per wave
        const uint32_t gid = g0 + lane;
        uint32_t c = 0u, pix = 0xffffffffu;
        if (gid < n_lines && gid < n_threads && lines.ok(gid)) {
            pix = lines.p[gid].path_ix;
            LineSetup s = line_setup(lines.p[gid], paths);
            if (s.valid) c = s.imax - s.imin;
        }
        if (gid < n_lines) counts[gid] = c;
        // Crossings per PATH (k_pc_emit: a path of more than PC_BIG_PATH crossings takes the list route).  Lines are in path order,
        // so a wave's lines are a few runs of equal path index: the run's sum comes out of one wave prefix sum, its last lane
        // sends ONE atomic.  (Round 5: the sums used to be derived from first / last line of every path by a launch of its own,
        // k_pc_paths, behind the scan; the ranges it also produced are written by k_pc_emit in passing now.)
        const uint32_t incl = wave_incl_scan_u32(c);
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~pix, (int)pix, 0x138, 0xf, 0xf, false);  // wave_shr:1 (lane 0: a head)
        const uint64_t heads = __builtin_amdgcn_ballot_w64(lane == 0u || pix != prev);
        const uint32_t leader = 63u - (uint32_t)__builtin_clzll(heads & ((2ull << lane) - 1ull));  // nearest head at or before the lane
        const uint32_t lead_incl = (uint32_t)__shfl((int)incl, (int)leader, 64), lead_c = (uint32_t)__shfl((int)c, (int)leader, 64);
        const bool last_of_run = lane == 63u || ((heads >> (lane + 1u)) & 1ull) != 0ull;
        const uint32_t run_sum = incl - (lead_incl - lead_c);
        if (last_of_run && run_sum != 0u && pix < ptotal_n) atomicAdd(&ptotal[pix], run_sum);
    }
}
// Crossing ranges of the paths: lines are in path order (canonical LineSoup order), so the crossings of path P are
// the contiguous range [pstart[P], pend[P]) of seg_counts.  Both arrays are zeroed before (paths without lines); k_pc_emit
// writes them at the first and the last line of every path.
// crossings [ps, pe) of path P (empty if it has no line)
JD void path_range(uint32_t P, const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ pend, const uint32_t*, const uint32_t*,
                   uint32_t& ps, uint32_t& pe) {
    ps = pstart[P];
    pe = pend[P];
}

// Paths with more crossings than this take the atomic route (per-tile arrival slots, lists, rank inside the list);
// all others get their slice ranks from k_pc_rank_small without a single atomic.  k_pc_rank_small is one wave per path;
// letting it walk a longer path in blocks of 64 is quadratic and, worse, a chain of memory round trips in ONE wave
// (measured: 45 us for a 256-crossing path, 1.7 ms for 20 circles of 700 crossings), while the atomic route is linear
// and spread over the whole device.  64 was the best threshold on every scene tried (tools/time_shapes.py, C3).
#ifndef PC_BIG_PATH
#define PC_BIG_PATH 64u  // (tools/sweep_pc.sh builds other values)
#endif
JD bool npe_big(uint32_t n) { return n > PC_BIG_PATH; }
JD uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// pass 2: backdrops, SegmentCount records, the tile of every crossing (slice ranks are filled in later).
// path_count.wgsl:168-199 is one thread per line walking the line's crossings; the walk only depends on the crossing
// number i (last_z of the WGSL is floor(a * (i - 1) + b)), so a long line -- an edge of a large rectangle crosses 256
// tiles, 1000 edge-to-edge strokes took 0.5 ms here -- is handed to the whole wave instead: lane = crossing.
#define PC_LONG_LINE 32u
struct EmitCtx {
    const JlConfig* cfg;
    const JlBump* bump;
    Buf<JlTile> tile;
    Buf<JlSegmentCount> seg_counts;
    uint2* tile_of;
    uint32_t* keys;
    uint32_t* kbig;
    uint32_t tile_of_n;
};
JD bool pc_key_is_big(uint32_t key) { return key != 0xffffffffu && (key >> 31) != 0u; }  // keys[]: see emit_crossing
JD void emit_backdrop_row(const EmitCtx& c, const LineSetup& s, int32_t y) {
    uint32_t base = (uint32_t)((int32_t)s.tiles + (y - s.bbox[1]) * s.stride);
    if (c.tile.ok(base)) atomicAdd(&c.tile.p[base].backdrop, s.delta);
}
// DEFER: the (tile, arrival) record of a big path's crossing is left to the caller (who merges the arrival atomics of
// neighbouring lanes); returns the crossing's tile and slot through `want_t` / `want_ix` (want_ix = ~0u: nothing to file).
template <bool DEFER>
JD void emit_crossing(const EmitCtx& c, const LineSetup& s, uint32_t i, uint32_t gid, uint32_t seg_base, bool big, uint32_t& want_t,
                      uint32_t& want_ix) {
    want_t = 0u;
    want_ix = 0xffffffffu;
    float last_z = floor_(s.a * ((float)i - 1.0f) + s.b);
    float zf = s.a * (float)i + s.b;
    float z = floor_(zf);
    int32_t y = to_i32(s.y0 + (float)i - z);
    int32_t x = to_i32(s.x0 + s.x_sign * z);
    int32_t base = (int32_t)s.tiles + (y - s.bbox[1]) * s.stride - s.bbox[0];
    bool top_edge = (i == 0u) ? (s.y0 == s.s0y) : (last_z == z);
    if (top_edge && x + 1 < s.bbox[2]) {
        int32_t x_bump = imax_(x + 1, s.bbox[0]);
        uint32_t t = (uint32_t)(base + x_bump);
        if (c.tile.ok(t)) atomicAdd(&c.tile.p[t].backdrop, s.delta);
    }
    uint32_t t = (uint32_t)(base + x);
    uint32_t seg_ix = seg_base + i - s.imin;
    if (seg_ix < c.cfg->seg_counts_size && c.seg_counts.ok(seg_ix)) {
        JlSegmentCount sc;
        sc.line_ix = gid;
        sc.counts = i;  // low 16 bits; the slice rank is OR-ed in by k_pc_rank_small / k_pc_rank
        c.seg_counts.p[seg_ix] = sc;
        if (seg_ix < c.tile_of_n) {
            // (the tile of the crossing; bit 31: a crossing of a big path, ranked by the list route -- a word of its own per crossing
            // was 18 MB of C3's frame for a flag only that route reads.  A crossing without a valid tile is nobody's business.)
            c.keys[seg_ix] = c.tile.ok(t) ? (t | (big ? 0x80000000u : 0u)) : 0xffffffffu;
            if (big) {
                if (DEFER) {
                    want_t = t;
                    want_ix = seg_ix;
                } else {
                    uint32_t arrival = 0u;  // order-dependent, only a unique slot inside the tile's temporary list
                    if (c.tile.ok(t)) arrival = atomicAdd(&c.tile.p[t].segment_count_or_ix, 1u);
                    c.tile_of[seg_ix] = make_uint2(t, arrival);
                }
            }
        }
    }
}
JD float rl_f(float v, uint32_t src) { return u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(v), (int)src)); }
JD int32_t rl_i(int32_t v, uint32_t src) { return __builtin_amdgcn_readlane(v, (int)src); }
JD uint32_t rl_u(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src); }

__global__ __launch_bounds__(JL_WG) void k_pc_emit(const JlConfig* __restrict__ cfg, const JlBump* __restrict__ bump,
                                                   const JlIndirectCount* __restrict__ ind, Buf<JlLineSoup> lines, Buf<JlPath> paths, Buf<JlTile> tile,
                                                   Buf<JlSegmentCount> seg_counts, const uint32_t* __restrict__ seg_bases, uint32_t bases_n,
                                                   uint2* __restrict__ tile_of, uint32_t* __restrict__ keys, uint32_t* __restrict__ kbig,
                                                   uint32_t tile_of_n, uint32_t* __restrict__ pfirst, uint32_t* __restrict__ plast,
                                                   const uint32_t* __restrict__ ptotal, uint32_t n_paths, uint32_t* __restrict__ gate) {
    EmitCtx c;
    c.cfg = cfg; c.bump = bump; c.tile = tile; c.seg_counts = seg_counts; c.tile_of = tile_of; c.keys = keys; c.kbig = kbig;
    c.tile_of_n = tile_of_n;
    const uint32_t n_lines = umin_(umin_(bump->lines, bases_n), ind->x * JL_WG);
    const uint32_t lane = lane_id();
    for (uint32_t g0 = blockIdx.x * JL_WG + (threadIdx.x & ~63u); g0 < n_lines; g0 += gridDim.x * JL_WG) {  // uniform per wave
        const uint32_t gid = g0 + lane;
        LineSetup s;
        s.valid = false;
        uint32_t P = 0xffffffffu;  // (a line outside the buffer)
        // (the path indices of the lines next door -- lanes 0 and 63 need them for the path ranges below -- are requested with the
        // wave's own lines, not behind line_setup's dependent loads: a round trip less)
        uint32_t edgeP = 0xffffffffu;
        if (lane == 0u && gid > 0u && gid - 1u < n_lines && lines.ok(gid - 1u)) edgeP = lines.p[gid - 1u].path_ix;
        if (lane == 63u && gid + 1u < n_lines && lines.ok(gid + 1u)) edgeP = lines.p[gid + 1u].path_ix;
        if (gid < n_lines && lines.ok(gid)) {
            P = lines.p[gid].path_ix;
            s = line_setup(lines.p[gid], paths);
        }
        // big path (or a path index outside the table): arrival slots by atomics, as the scatter pass / k_pc_rank expect
        bool big = true;
        uint32_t seg_base = 0u;
        if (gid < n_lines) seg_base = seg_bases[gid];
        if (s.valid) {
            if (P < n_paths) big = ptotal[P] > PC_BIG_PATH;  // (= pend - pstart: the crossings of all of the path's lines)
        } else {
            s.imin = 0u; s.imax = 0u; s.ymin = 0; s.ymax = 0;
        }
        {   // the path's crossing range, written at its first and its last line (the neighbours: a lane over, or the line next door)
            uint32_t prevP = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x138, 0xf, 0xf, false);  // wave_shr:1
            uint32_t nextP = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x130, 0xf, 0xf, false);  // wave_shl:1
            if (lane == 0u) prevP = edgeP;
            if (lane == 63u) nextP = edgeP;
            if (gid < n_lines && P < n_paths) {
                if (P != prevP) pfirst[P] = seg_base;
                if (P != nextP) plast[P] = seg_base + (s.valid ? s.imax - s.imin : 0u);
            }
        }
        // open the gate of the list route: the number of tiles its list-base scan has to cover (one lane per wave, and
        // only while the word does not hold the value yet -- 200 k lines of one big path would queue on that word)
        if (__builtin_amdgcn_ballot_w64(s.valid && big && s.imax > s.imin) != 0ull && lane == 0u) {
            const uint32_t want = bump->tile;
            if (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) atomicMax(gate, want);
        }
        const bool is_long = s.valid && (s.imax - s.imin > PC_LONG_LINE || s.ymax - s.ymin > (int32_t)PC_LONG_LINE);
        const bool per_lane = s.valid && !is_long;
        if (per_lane)
            for (int32_t y = s.ymin; y < s.ymax; y++) emit_backdrop_row(c, s, y);
        uint32_t dummy_t, dummy_ix;
        if (__builtin_amdgcn_ballot_w64(per_lane && big) == 0ull) {  // no big path among the wave's lines (all of C3)
            if (per_lane)
                for (uint32_t i = s.imin; i < s.imax; i++) emit_crossing<false>(c, s, i, gid, seg_base, big, dummy_t, dummy_ix);
        } else {
            // Consecutive short lines of a big path (a polyline, both sides of a thin stroke) keep hitting the same tile:
            // 6 M returning atomics on a road map queued per cache line (0.41 ms).  The walk is made wave-uniform and
            // every run of neighbouring lanes with the same tile sends ONE atomic for the run (grouping ALL equal tiles of a
            // step with a ballot loop measured slower: 0.70 vs 0.62 ms of path_count on the road map).
            const uint32_t trip = per_lane ? s.imax - s.imin : 0u;
            const uint32_t max_trip = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_max_u32(trip), 63);
            for (uint32_t r = 0u; r < max_trip; r++) {  // uniform
                uint32_t t = 0u, ix = 0xffffffffu;
                if (r < trip) emit_crossing<true>(c, s, s.imin + r, gid, seg_base, big, t, ix);
                const bool want = ix != 0xffffffffu && c.tile.ok(t);
                const uint32_t key = want ? t : 0xffffffffu - lane;  // (unique: never joins a run)
                const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64);
                const bool head = lane == 0u || key != prev;
                const uint64_t heads = __builtin_amdgcn_ballot_w64(head);
                const uint32_t leader = 63u - (uint32_t)__builtin_clzll(heads & ((2ull << lane) - 1ull));  // nearest head at or before the lane
                const uint64_t later = leader == 63u ? 0ull : heads >> (leader + 1u);
                const uint32_t run = later == 0ull ? 64u - leader : (uint32_t)__builtin_ctzll(later) + 1u;
                uint32_t first = 0u;
                if (want && lane == leader) first = atomicAdd(&c.tile.p[t].segment_count_or_ix, run);
                first = (uint32_t)__shfl((int)first, (int)leader, 64);
                if (ix != 0xffffffffu) c.tile_of[ix] = make_uint2(t, want ? first + (lane - leader) : 0u);
            }
        }
        uint64_t m = __builtin_amdgcn_ballot_w64(is_long);
        while (m != 0ull) {  // uniform
            const uint32_t src = (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
            LineSetup t;
            t.valid = true; t.is_down = false; t.is_positive_slope = false;
            t.a = rl_f(s.a, src); t.b = rl_f(s.b, src); t.x0 = rl_f(s.x0, src); t.y0 = rl_f(s.y0, src);
            t.x_sign = rl_f(s.x_sign, src); t.s0y = rl_f(s.s0y, src);
            t.imin = rl_u(s.imin, src); t.imax = rl_u(s.imax, src);
            t.ymin = rl_i(s.ymin, src); t.ymax = rl_i(s.ymax, src); t.delta = rl_i(s.delta, src);
            t.bbox[0] = rl_i(s.bbox[0], src); t.bbox[1] = rl_i(s.bbox[1], src); t.bbox[2] = rl_i(s.bbox[2], src); t.bbox[3] = rl_i(s.bbox[3], src);
            t.stride = rl_i(s.stride, src); t.tiles = rl_u(s.tiles, src);
            const uint32_t t_gid = g0 + src, t_seg_base = rl_u(seg_base, src);
            const bool t_big = rl_u(big ? 1u : 0u, src) != 0u;
            for (int32_t y = t.ymin + (int32_t)lane; y < t.ymax; y += 64) emit_backdrop_row(c, t, y);
            for (uint32_t i = t.imin + lane; i < t.imax; i += 64u) emit_crossing<false>(c, t, i, t_gid, t_seg_base, t_big, dummy_t, dummy_ix);
        }
    }
}

// Slice ranks without atomics.  One wave takes one path of at most 64 crossings (lane = crossing): its crossings are
// the contiguous range [pstart, pend) and crossings in the same tile are crossings of the same path (tile_alloc gives
// every path its own tile range), hence
//   seg_within_slice(k) = #{ j < k : key[j] == key[k] },   Tile.segment_count = that number + #{ j > k : ... } + 1,
// the canonical (line, crossing) order by construction, from one ballot per DISTINCT tile of the path.
#ifndef PCR_K
#define PCR_K 16u  // consecutive paths per task of k_pc_rank_small (<= 64)
#endif
struct PcScatterArgs {  // the list route's scatter pass rides in the same launch (its last `blocks` workgroups): it and the
    const uint2* tile_of;   // ranking of the small paths touch different tiles and crossings, and both follow the
    const uint32_t* list_base;  // list-base scan; a launch of its own cost 4.7 us even when no path takes the route)
    uint32_t tiles_cap;
    uint32_t* list;
    const uint32_t* kbig;
    uint32_t* gate;
    uint32_t* dense;
    uint32_t dense_cap, blocks;
};
JD void pc_scatter_part(const JlConfig* __restrict__ cfg, const JlBump* __restrict__ bump, Buf<JlTile> tile,
                        const uint2* __restrict__ tile_of, uint32_t n_cap, const uint32_t* __restrict__ list_base,
                        uint32_t tiles_cap, uint32_t* __restrict__ list, const uint32_t* __restrict__ kbig,
                        uint32_t* __restrict__ gate, uint32_t* __restrict__ dense, uint32_t dense_cap, uint32_t block, uint32_t blocks);
__global__ __launch_bounds__(JL_WG) void k_pc_rank_small(const JlConfig* __restrict__ cfg, const JlBump* __restrict__ bump, Buf<JlTile> tile,
                                                         const uint32_t* __restrict__ keys, uint32_t n_cap, const uint32_t* __restrict__ pfirst,
                                                         const uint32_t* __restrict__ plast, const uint32_t* __restrict__ counts,
                                                         const uint32_t* __restrict__ seg_bases, uint32_t n_paths,
                                                         Buf<JlSegmentCount> seg_counts, PcScatterArgs sc, uint32_t* __restrict__ ptotal_zero,
                                                         uint32_t ptotal_words) {
    // (the crossings-per-path sums have served k_pc_emit: back to zero for the next frame's atomics -- kcommon.h, JH_CLEAN_*)
    for (uint32_t i = blockIdx.x * JL_WG + threadIdx.x; i < ptotal_words; i += gridDim.x * JL_WG) ptotal_zero[i] = 0u;
    const uint32_t rank_blocks = gridDim.x - sc.blocks;  // the scatter workgroups come last: they only look at the gate when it is closed
    if (blockIdx.x >= rank_blocks) {  // uniform
        pc_scatter_part(cfg, bump, tile, sc.tile_of, n_cap, sc.list_base, sc.tiles_cap, sc.list, sc.kbig, sc.gate, sc.dense, sc.dense_cap,
                        blockIdx.x - rank_blocks, sc.blocks);
        return;
    }
    const uint32_t n = umin_(umin_(bump->seg_counts, cfg->seg_counts_size), n_cap);
    const uint32_t lane = lane_id();
    const uint32_t waves = (rank_blocks * JL_WG) >> 6;
    const uint32_t wave_ix = (blockIdx.x * JL_WG + threadIdx.x) >> 6;
    // Round 4: a wave ranks a CHUNK of consecutive small paths at a time -- as many as fit its 64 lanes (a C3 path has 23
    // crossings: one path per iteration used 36 % of the lanes, and ~60 % of an iteration's 173 instructions were per-path
    // overhead).  Tiles are unique per path, so "same key" still means "same tile of the same path" inside a chunk, and the
    // ballots per distinct tile rank the crossings of all the chunk's paths at once.  A wave takes tasks of PCR_K consecutive
    // paths (their ranges: one coalesced load per task, requested a task ahead), cuts each task into chunks -- consecutive
    // paths whose crossing ranges adjoin (an empty or a big path, or a range cut off at the buffer's end, ends a chunk) and
    // together hold at most 64 crossings -- and requests the keys / counts words of chunk c + 1 before it ranks chunk c.
    const uint32_t n_tasks = (n_paths + PCR_K - 1u) / PCR_K;
    auto load_ranges = [&](uint32_t task, uint32_t& ps, uint32_t& pe) {
        ps = 0u; pe = 0u;
        const uint32_t P = task * PCR_K + lane;
        if (task < n_tasks && lane < PCR_K && P < n_paths) path_range(P, pfirst, plast, counts, seg_bases, ps, pe);
    };
    uint32_t t_ps = 0u, t_pe = 0u;   // lane i < PCR_K: crossings [t_ps, t_pe) of path task * PCR_K + i (t_pe cut to the buffer)
    uint64_t todo = 0ull;            // paths of the task still to rank (small, non-empty)
    uint32_t task = wave_ix;
    uint32_t r_ps, r_pe;             // the ranges of the NEXT task, in flight
    load_ranges(task, r_ps, r_pe);
    bool task_loaded = false;
    auto advance = [&](uint32_t& cs, uint32_t& ce) -> bool {  // the next chunk [cs, ce) of this wave, false when there is none (uniform)
        for (;;) {
            if (todo != 0ull) {
                const uint32_t a = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                cs = (uint32_t)__builtin_amdgcn_readlane((int)t_ps, (int)a);
                ce = (uint32_t)__builtin_amdgcn_readlane((int)t_pe, (int)a);
                while (todo != 0ull) {
                    const uint32_t b2 = (uint32_t)__builtin_ctzll(todo);
                    const uint32_t nps = (uint32_t)__builtin_amdgcn_readlane((int)t_ps, (int)b2);
                    const uint32_t npe = (uint32_t)__builtin_amdgcn_readlane((int)t_pe, (int)b2);
                    if (nps != ce || npe - cs > 64u) break;
                    ce = npe;
                    todo &= todo - 1ull;
                }
                return true;
            }
            if (task_loaded) task += waves;
            if (task >= n_tasks) return false;
            task_loaded = true;
            const uint32_t ps = r_ps, pe_all = r_pe;
            load_ranges(task + waves, r_ps, r_pe);
            t_ps = ps;
            t_pe = umin_(pe_all, n);
            todo = __builtin_amdgcn_ballot_w64(lane < PCR_K && t_pe > ps && !npe_big(pe_all - ps));  // (the same test as in k_pc_emit)
        }
    };
    auto fetch = [&](uint32_t cs, uint32_t ce, uint32_t& key, uint32_t& cw) {
        key = 0xffffffffu; cw = 0u;
        const uint32_t k = cs + lane;
        if (k < ce) {
            key = keys[k];
            if (seg_counts.ok(k)) cw = seg_counts.p[k].counts;
        }
    };
    uint32_t c_s = 0u, c_e = 0u, key1 = 0xffffffffu, cw1 = 0u;
    bool have = advance(c_s, c_e);
    if (have) fetch(c_s, c_e, key1, cw1);
    while (have) {  // uniform
        const uint32_t cs = c_s, ce = c_e;
        const uint32_t my_key = key1, my_cw = cw1;
        have = advance(c_s, c_e);
        if (have) fetch(c_s, c_e, key1, cw1);
        const uint32_t k = cs + lane;
        const bool valid = k < ce;
        const uint32_t my_t = valid ? my_key : 0xffffffffu;
        const bool mine = valid && my_t != 0xffffffffu;  // 0xffffffff: crossing outside the tile buffer
        uint32_t before = 0u, after = 0u;
        uint64_t rem = __builtin_amdgcn_ballot_w64(mine);
        // One trip per DISTINCT tile: the lanes of that tile keep the trip's ballot (lanes that are not `mine` hold the key
        // 0xffffffff, which no trip asks for); the ranks come out of the kept masks once, after the loop -- the loop itself is
        // a scalar find-first, a lane read, a compare and two selects.
        uint32_t grp_lo = 0u, grp_hi = 0u;
        while (rem != 0ull) {  // at most 64 trips
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)my_t, (int)__builtin_ctzll(rem));
            const bool eq = my_t == t;
            const uint64_t m = __builtin_amdgcn_ballot_w64(eq);
            grp_lo = eq ? (uint32_t)m : grp_lo;
            grp_hi = eq ? (uint32_t)(m >> 32) : grp_hi;
            rem &= ~m;
        }
        if (mine) {
            before = __builtin_amdgcn_mbcnt_hi(grp_hi, __builtin_amdgcn_mbcnt_lo(grp_lo, 0u));
            after = (uint32_t)__builtin_popcount(grp_lo) + (uint32_t)__builtin_popcount(grp_hi) - 1u - before;
        }
        if (mine && seg_counts.ok(k)) {
            seg_counts.p[k].counts = my_cw | (before << 16);
            if (after == 0u && tile.ok(my_t)) tile.p[my_t].segment_count_or_ix = before + 1u;
        }
    }
}

#define PC_DENSE_TILE 16u   // tiles with longer lists are ranked by a wave each (k_pc_rank, phase B)
#define PC_DENSE_LDS 1024u  // entries of a tile's list staged per pass and wave
// pass 3: scatter crossing indices into per-tile lists; the slot inside a list is the (arbitrary but unique)
// arrival number the count atomic returned in pass 2, so no further atomics are needed.
JD void pc_scatter_part(const JlConfig* __restrict__ cfg, const JlBump* __restrict__ bump, Buf<JlTile> tile,
                        const uint2* __restrict__ tile_of, uint32_t n_cap, const uint32_t* __restrict__ list_base,
                        uint32_t tiles_cap, uint32_t* __restrict__ list, const uint32_t* __restrict__ kbig,
                        uint32_t* __restrict__ gate, uint32_t* __restrict__ dense, uint32_t dense_cap, uint32_t block, uint32_t blocks) {
    if (*gate == 0u) return;  // no big path in this frame
    uint32_t n = umin_(umin_(bump->seg_counts, cfg->seg_counts_size), n_cap);
    const uint32_t lane = lane_id();
    for (uint32_t k0 = block * JL_WG + (threadIdx.x & ~63u); k0 < n; k0 += blocks * JL_WG) {  // uniform per wave
        const uint32_t k = k0 + lane;
        bool first_of_dense = false;  // the crossing that arrived first in a tile with a long list announces the tile
        uint32_t t = 0u;
        if (k < n && pc_key_is_big(kbig[k])) {  // (others are ranked by k_pc_rank_small)
            uint2 ta = tile_of[k];
            if (ta.x < tiles_cap && tile.ok(ta.x)) {
                uint32_t pos = list_base[ta.x] + ta.y;
                if (pos < n_cap) list[pos] = k;
                t = ta.x;
                first_of_dense = ta.y == 0u && tile.p[t].segment_count_or_ix > PC_DENSE_TILE;
            }
        }
        const uint64_t m = __builtin_amdgcn_ballot_w64(first_of_dense);
        if (m != 0ull) {  // one atomic per wave
            uint32_t at = 0u;
            if (lane == 0u) at = atomicAdd(&gate[1], (uint32_t)__builtin_popcountll(m));
            at = uni32(at) + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (first_of_dense && at < dense_cap) dense[at] = t;
        }
    }
}
// pass 4: seg_within_slice = rank of k among the crossings of its tile.
// Phase A, thread = crossing: counts the smaller entries of its tile's list -- fine for the usual handful of crossings
// per tile.  A dense polyline (a 200 k-vertex outline: 240 crossings in each tile of its ring) makes that
// 240 uncoalesced loads per crossing (1.1 ms); tiles with more than PC_DENSE_TILE entries are therefore left to
// phase B, wave = tile: the list goes to LDS once, every lane ranks its entries against broadcast reads.
__global__ __launch_bounds__(JL_WG) void k_pc_rank(const JlConfig* __restrict__ cfg, const JlBump* __restrict__ bump, Buf<JlTile> tile,
                                                   const uint2* __restrict__ tile_of, uint32_t n_cap, const uint32_t* __restrict__ list_base,
                                                   uint32_t tiles_cap, const uint32_t* __restrict__ list, Buf<JlSegmentCount> seg_counts,
                                                   const uint32_t* __restrict__ kbig, const uint32_t* __restrict__ gate,
                                                   const uint32_t* __restrict__ dense, uint32_t dense_cap) {
    __shared__ uint32_t sh_list[JL_WG / 64][PC_DENSE_LDS];
    if (*gate == 0u) return;  // no big path in this frame
    uint32_t n = umin_(umin_(bump->seg_counts, cfg->seg_counts_size), n_cap);
    for (uint32_t k = blockIdx.x * JL_WG + threadIdx.x; k < n; k += gridDim.x * JL_WG) {
        if (!pc_key_is_big(kbig[k])) continue;  // ranked by k_pc_rank_small
        uint32_t t = tile_of[k].x;
        if (t >= tiles_cap || !tile.ok(t) || !seg_counts.ok(k)) continue;
        uint32_t cnt = tile.p[t].segment_count_or_ix;
        if (cnt > PC_DENSE_TILE) continue;  // phase B
        uint32_t base = list_base[t];
        uint32_t rank = 0u;
        for (uint32_t j = 0; j < cnt; j++) {
            uint32_t pos = base + j;
            if (pos < n_cap && list[pos] < k) rank++;
        }
        seg_counts.p[k].counts |= rank << 16;
    }
    // phase B: one wave per announced tile
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    uint32_t* my_list = sh_list[wv];
    const uint32_t n_dense = umin_(gate[1], dense_cap);
    const uint32_t waves = (gridDim.x * JL_WG) >> 6;
    for (uint32_t d = uni32((blockIdx.x * JL_WG + threadIdx.x) >> 6); d < n_dense; d += waves) {
        const uint32_t t = uni32(dense[d]);
        if (t >= tiles_cap || !tile.ok(t)) continue;
        const uint32_t base = uni32(list_base[t]);
        uint32_t m = uni32(tile.p[t].segment_count_or_ix);
        if (base >= n_cap) continue;
        m = umin_(m, n_cap - base);
        for (uint32_t o0 = 0u; o0 < m; o0 += PC_DENSE_LDS) {  // the "other" entries, PC_DENSE_LDS at a time
            const uint32_t on = umin_(m - o0, PC_DENSE_LDS);
            wave_sync();
            for (uint32_t i = lane; i < ((on + 3u) & ~3u); i += 64u) my_list[i] = i < on ? list[base + o0 + i] : 0xffffffffu;
            wave_sync();
            for (uint32_t j = lane; j < m; j += 64u) {  // own entries
                const uint32_t e = list[base + j];
                uint32_t r = 0u;
                for (uint32_t i = 0u; i < on; i += 4u) {
                    const uint4 q = *(const uint4*)&my_list[i];  // broadcast read
                    r += (q.x < e ? 1u : 0u) + (q.y < e ? 1u : 0u) + (q.z < e ? 1u : 0u) + (q.w < e ? 1u : 0u);
                }
                if (seg_counts.ok(e)) seg_counts.p[e].counts += r << 16;  // (this lane owns entry e; passes add up)
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backdrop_dyn.wgsl:28-86
// ------------------------------------------------------------------------------------------------
//
// The WGSL gives every row of a path to one thread, which walks the row tile by tile, and the rows of 256 paths to
// one workgroup: fine for small paths, but a row of 100 tiles is a chain of 100 dependent memory round trips and the
// 24 k rows of 300 large circles land on two workgroups (2.8 ms).  Paths wider than `wide_min` tiles (BD_WIDE; less
// when the scene has so few draw objects that this kernel cannot fill the device anyway) are therefore only
// listed here -- (first global row number, first tile, width, rows), appended with one 64-bit atomic per wave that advances the
// entry count and the row total together -- and k_backdrop_wide gives each of their rows to a wave (prefix sums are
// integer sums: any association gives the WGSL's result).
#define BD_WIDE 16u
#define BD_UNIT 4u  // consecutive rows per wave step (one list search per unit)
#define BD_ROWS_MASK ((1ull << 40) - 1ull)
__global__ __launch_bounds__(JL_WG) void k_backdrop_dyn(const JlConfig* __restrict__ cfg, const JlBump* __restrict__ bump, Buf<JlPath> paths,
                                                        Buf<JlTile> tiles, unsigned long long* __restrict__ wide_ctr, uint4* __restrict__ wide_list,
                                                        uint32_t wide_cap, uint32_t wide_min) {
    __shared__ uint32_t sh_row_width[JL_WG];
    __shared__ uint32_t sh_row_count[JL_WG];
    __shared__ uint32_t sh_offset[JL_WG];
    __shared__ uint32_t sh_scan[8];
    if (bump->failed != 0u) return;
    uint32_t lid = threadIdx.x;
    uint32_t drawobj_ix = blockIdx.x * JL_WG + lid;
    uint32_t row_count = 0u;
    if (drawobj_ix < cfg->layout.n_drawobj) {
        JlPath path = paths.rd(drawobj_ix);
        sh_row_width[lid] = path.bbox[2] - path.bbox[0];
        row_count = path.bbox[3] - path.bbox[1];
        sh_offset[lid] = path.tiles;
    } else {
        sh_row_width[lid] = 0u;
        sh_offset[lid] = 0u;
    }
    {
        const bool wide = sh_row_width[lid] > wide_min && row_count > 0u && drawobj_ix < wide_cap;
        const uint64_t m = __builtin_amdgcn_ballot_w64(wide);
        if (m != 0ull) {  // uniform per wave
            const uint32_t rows = wide ? row_count : 0u;
            const uint32_t incl = wave_incl_scan_u32(rows);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t lane = lane_id();
            unsigned long long old = 0ull;
            if (lane == 0u) old = atomicAdd(wide_ctr, ((unsigned long long)__builtin_popcountll(m) << 40) | (unsigned long long)total);
            const uint32_t old_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)old);
            const uint32_t old_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(old >> 32));
            const unsigned long long o = ((unsigned long long)old_hi << 32) | old_lo;
            if (wide) {
                const uint32_t slot = (uint32_t)(o >> 40) + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (slot < wide_cap)  // (first global row, first tile, tiles per row, rows)
                    wide_list[slot] = make_uint4((uint32_t)(o & BD_ROWS_MASK) + incl - rows, sh_offset[lid], sh_row_width[lid], rows);
                row_count = 0u;  // not dealt to this workgroup's threads
            }
        }
    }
    uint32_t total_rows;
    uint32_t excl = block_excl_scan_u32(row_count, sh_scan, &total_rows);
    sh_row_count[lid] = excl + row_count;  // inclusive
    __syncthreads();
    for (uint32_t row = lid; row < total_rows; row += JL_WG) {
        uint32_t el_ix = 0u;
        for (uint32_t i = 0; i < 8u; i++) {
            uint32_t probe = el_ix + (128u >> i);
            if (row >= sh_row_count[probe - 1u]) el_ix = probe;
        }
        uint32_t width = sh_row_width[el_ix];
        if (width > 0u) {
            uint32_t seq_ix = row - (el_ix > 0u ? sh_row_count[el_ix - 1u] : 0u);
            uint32_t tile_ix = sh_offset[el_ix] + seq_ix * width;
            int32_t sum = tiles.rd(tile_ix).backdrop;
            for (uint32_t x = 1u; x < width; x++) {
                tile_ix += 1u;
                if (!tiles.ok(tile_ix)) break;
                sum += tiles.p[tile_ix].backdrop;
                tiles.p[tile_ix].backdrop = sum;
            }
        }
    }
}

// One wave per row of a wide path: 64 tiles per step, inclusive scan by DPP, carry in a scalar.
__global__ __launch_bounds__(JL_WG) void k_backdrop_wide(const JlBump* __restrict__ bump, Buf<JlTile> tiles,
                                                         const unsigned long long* __restrict__ wide_ctr, const uint4* __restrict__ wide_list,
                                                         uint32_t wide_cap) {
    if (bump->failed != 0u) return;
    const unsigned long long c = *wide_ctr;
    const uint32_t n_wide = umin_((uint32_t)(c >> 40), wide_cap);
    const uint32_t total_rows = (uint32_t)(c & BD_ROWS_MASK);
    if (n_wide == 0u) return;
    const uint32_t lane = lane_id();
    const uint32_t waves = (gridDim.x * JL_WG) >> 6;
    for (uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * JL_WG + threadIdx.x) >> 6)) * BD_UNIT; g0 < total_rows;
         g0 += waves * BD_UNIT) {
        uint32_t lo = 0u, hi = n_wide;  // the last entry whose first row is <= g0
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (wide_list[mid].x <= g0) lo = mid; else hi = mid;
        }
        uint4 e = wide_list[lo];
        for (uint32_t g = g0; g < umin_(g0 + BD_UNIT, total_rows); g++) {
            while (g - e.x >= e.w && lo + 1u < n_wide) {  // next entry (entries have at least one row)
                lo += 1u;
                e = wide_list[lo];
            }
            const uint32_t r = g - e.x;
            if (r >= e.w) break;
            const uint32_t width = e.z;
            const uint32_t base = e.y + r * width;
            uint32_t carry = 0u;
            for (uint32_t x0 = 0u; x0 < width; x0 += 256u) {  // four steps of 64 tiles per trip: their loads are in flight together
                uint32_t v[4];
                bool ok[4];
#pragma unroll
                for (uint32_t q = 0u; q < 4u; q++) {
                    const uint32_t x = x0 + q * 64u + lane;
                    ok[q] = x < width && tiles.ok(base + x);
                    v[q] = ok[q] ? (uint32_t)tiles.p[base + x].backdrop : 0u;
                }
#pragma unroll
                for (uint32_t q = 0u; q < 4u; q++) {
                    const uint32_t sum = wave_incl_scan_u32(v[q]) + carry;
                    if (ok[q]) tiles.p[base + x0 + q * 64u + lane].backdrop = (int32_t)sum;
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)sum, 63);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// path_tiling.wgsl:39-173
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(JL_WG) void k_path_tiling(const JlBump* __restrict__ bump, const JlIndirectCount* __restrict__ ind,
                                                       Buf<JlSegmentCount> seg_counts, Buf<JlLineSoup> lines, Buf<JlPath> paths, Buf<JlTile> tiles,
                                                       Buf<JlSegment> segments,
                                                       JlIndirectCount* __restrict__ setup_out, Buf<uint32_t> setup_ptcl) {  // setup_out != nullptr: path_tiling_setup
                                                                                   // was held back: this kernel does its work (path_tiling_setup.wgsl:20-32)
    uint32_t ind_x;
    if (setup_out) {
        const bool failed = bump->failed != 0u;
        ind_x = failed ? 0u : (bump->seg_counts + (JL_WG - 1u)) / JL_WG;
        if (blockIdx.x == 0u && threadIdx.x == 0u) {
            setup_out->x = ind_x; setup_out->y = 1u; setup_out->z = 1u;
            if (failed) setup_ptcl.wr(0u, ~0u);
        }
    } else {
        ind_x = ind->x;
    }
    uint32_t n_segments = umin_(bump->seg_counts, ind_x * JL_WG);
    for (uint32_t gid = blockIdx.x * JL_WG + threadIdx.x; gid < n_segments; gid += gridDim.x * JL_WG) {
        JlSegmentCount sc = seg_counts.rd(gid);
        JlLineSoup line = lines.rd(sc.line_ix);
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
        float b = fmin_((dy * c + dx * (ytop - s0.y)) * idxdy, ONE_MINUS_ULP);
        float robust_err = floor_(a * ((float)count - 1.0f) + b) - (float)count_x;
        if (robust_err != 0.0f) a -= ROBUST_EPSILON * sign_(robust_err);
        int32_t x0i = to_i32(xt0 * x_sign + 0.5f * (x_sign - 1.0f));
        float z = floor_(a * (float)seg_within_line + b);
        int32_t x = x0i + to_i32(x_sign * z);
        int32_t y = to_i32(y0i + (float)seg_within_line - z);
        JlPath path = paths.rd(line.path_ix);
        int32_t bbox[4] = {(int32_t)path.bbox[0], (int32_t)path.bbox[1], (int32_t)path.bbox[2], (int32_t)path.bbox[3]};
        int32_t stride = bbox[2] - bbox[0];
        int32_t tile_ix = (int32_t)path.tiles + (y - bbox[1]) * stride + x - bbox[0];
        JlTile tile = tiles.rd((uint32_t)tile_ix);
        uint32_t seg_start = ~tile.segment_count_or_ix;
        if ((int32_t)seg_start < 0) continue;
        V2 tile_xy = v2((float)x * 16.0f, (float)y * 16.0f);
        V2 tile_xy1 = tile_xy + v2(16.0f, 16.0f);
        if (seg_within_line > 0u) {
            float z_prev = floor_(a * ((float)seg_within_line - 1.0f) + b);
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
            float z_next = floor_(a * ((float)seg_within_line + 1.0f) + b);
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
        JlSegment seg;
        seg.p0[0] = p0.x; seg.p0[1] = p0.y; seg.p1[0] = p1.x; seg.p1[1] = p1.y; seg.y_edge = y_edge; seg.pad = 0u;
        segments.wr(seg_start + seg_within_slice, seg);
    }
}



}  // namespace

#ifndef TILE_GRID_PER_CU
#define TILE_GRID_PER_CU 8  // workgroups per CU of the grid-stride kernels of this file
#endif
#ifndef PC_EMIT_GRID_PER_CU
#define PC_EMIT_GRID_PER_CU 32  // k_pc_emit: the crossings per line vary widely and a wave's share of a small grid is a long chain of dependent
                                // loads: C3 64.9 / 52.2 / 44.0 / 42.2 / 41.0 / 42.6 us with 4 / 8 / 16 / 24 / 32 / 64 workgroups per CU (round 5, same box);
                                // C4 55 -> 48.  The other kernels of the file do not care (k_pc_count 20.5 / 22.1 / 20.8 with 8 / 12 / 16, k_pc_rank_small
                                // 30 / 31 / 31, k_path_tiling 65.5 / 66.8 / 65.2)
#endif
static inline uint32_t stride_grid(const JhLaunch& L, uint64_t n_items, uint32_t per_cu = TILE_GRID_PER_CU) {
    uint64_t blocks = (n_items + JL_WG - 1) / JL_WG;
    uint64_t cap = (uint64_t)(L.num_cus > 0 ? L.num_cus : 256) * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (uint32_t)blocks;
}

// [config, draw_monoids, path_bbox, clip_bbox, intersected_bbox, bump, bin_data, bin_header]
int jh_launch_binning(const JhLaunch& L) {
    if (L.nb < 8) return -1;
    if (L.gx == 0) return 0;
    uint32_t* wg_tot = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_A, (uint64_t)L.gx * 4);
    if (!wg_tot) return -5;
    auto cfg = (const JlConfig*)L.b[0].ptr;
    auto dm = mkbuf<JlDrawMonoid>(L.b[1].ptr, L.b[1].size);
    auto pb = mkbuf<JlPathBbox>(L.b[2].ptr, L.b[2].size);
    auto cb = mkbuf<Bb4>(L.b[3].ptr, L.b[3].size);
    auto ib = mkbuf<Bb4>(L.b[4].ptr, L.b[4].size);
    JlBump* bump = (JlBump*)L.b[5].ptr;
    auto bd = mkbuf<uint32_t>(L.b[6].ptr, L.b[6].size);
    auto bh = mkbuf<JlBinHeader>(L.b[7].ptr, L.b[7].size);
    hipLaunchKernelGGL(k_binning<0>, dim3(L.gx), dim3(JL_WG), 0, L.stream, cfg, dm, pb, cb, ib, bump, bd, bh, wg_tot);
    hipLaunchKernelGGL(k_binning<1>, dim3(L.gx), dim3(JL_WG), 0, L.stream, cfg, dm, pb, cb, ib, bump, bd, bh, wg_tot);
    return 0;
}

// [config, scene, draw_bboxes, bump, paths, tiles]
int jh_launch_tile_alloc(const JhLaunch& L) {
    if (L.nb < 6) return -1;
    if (L.gx == 0) return 0;
    uint32_t n = L.gx * JL_WG;
    uint32_t* counts = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_A, (uint64_t)n * 4);
    uint32_t* wg_tot = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_B, (uint64_t)L.gx * 4);
    if (!counts || !wg_tot) return -5;
    auto cfg = (const JlConfig*)L.b[0].ptr;
    auto scene = mkbuf<uint32_t>(L.b[1].ptr, L.b[1].size);
    auto db = mkbuf<Bb4>(L.b[2].ptr, L.b[2].size);
    JlBump* bump = (JlBump*)L.b[3].ptr;
    auto paths = mkbuf<JlPath>(L.b[4].ptr, L.b[4].size);
    auto tiles = mkbuf<JlTile>(L.b[5].ptr, L.b[5].size);
    hipLaunchKernelGGL(k_tile_alloc_count, dim3(L.gx), dim3(JL_WG), 0, L.stream, cfg, scene, db, (const JlBump*)bump, paths, counts, wg_tot);
    hipLaunchKernelGGL(k_tile_alloc_write, dim3(L.gx + stride_grid(L, (uint64_t)tiles.n / 2u + 1u)), dim3(JL_WG), 0, L.stream, cfg, bump, paths, tiles,
                       (const uint32_t*)counts, (const uint32_t*)wg_tot, L.gx);
    return 0;
}

// [bump, indirect]
int jh_launch_path_count_setup(const JhLaunch& L) {
    if (L.nb < 2) return -1;
    hipLaunchKernelGGL(k_path_count_setup, dim3(1), dim3(1), 0, L.stream, (const JlBump*)L.b[0].ptr, (JlIndirectCount*)L.b[1].ptr);
    return 0;
}
// [bump, indirect, ptcl]
int jh_launch_path_tiling_setup(const JhLaunch& L) {
    if (L.nb < 3) return -1;
    hipLaunchKernelGGL(k_path_tiling_setup, dim3(1), dim3(1), 0, L.stream, (const JlBump*)L.b[0].ptr, (JlIndirectCount*)L.b[1].ptr,
                       mkbuf<uint32_t>(L.b[2].ptr, L.b[2].size));
    return 0;
}

// indirect; [config, bump, lines, paths, tile, seg_counts]
int jh_launch_path_count(const JhLaunch& L) {
    if (L.nb < 6 || !L.indirect) return -1;
    auto cfg = (const JlConfig*)L.b[0].ptr;
    JlBump* bump = (JlBump*)L.b[1].ptr;
    auto lines = mkbuf<JlLineSoup>(L.b[2].ptr, L.b[2].size);
    auto paths = mkbuf<JlPath>(L.b[3].ptr, L.b[3].size);
    auto tile = mkbuf<JlTile>(L.b[4].ptr, L.b[4].size);
    auto segc = mkbuf<JlSegmentCount>(L.b[5].ptr, L.b[5].size);
    auto ind = (const JlIndirectCount*)L.indirect;
    uint32_t lines_cap = lines.n, seg_cap = segc.n, tiles_cap = tile.n;
    uint32_t* counts = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_A, (uint64_t)lines_cap * 4);
    uint32_t* bases = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_B, (uint64_t)lines_cap * 4);
    uint2* tile_of = (uint2*)jh_scratch_get(L.scratch, JH_SCR_C, (uint64_t)seg_cap * 8);
    uint32_t* list = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_D, (uint64_t)seg_cap * 4);
    uint32_t* list_base = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_E, (uint64_t)tiles_cap * 4);
    uint32_t* keys = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_F, (uint64_t)seg_cap * 4);
    uint32_t* kbig = keys;  // (the big-path flag is bit 31 of the crossing's key)
    uint32_t n_paths = paths.n;
    const uint32_t dense_cap = seg_cap / PC_DENSE_TILE + 1u;  // tiles with more than PC_DENSE_TILE crossings of a big path
    uint32_t* dense = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_G, (uint64_t)dense_cap * 4);
    if (!dense) return -5;
    // [pstart | pend | gate, number of dense tiles ... | crossings per path]: zeroed every frame (by the launch in front of k_pc_count's atomics
    // on the last part, see below) (the variables below keep the names of the path_range parameters)
    uint32_t* prange = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_I, ((uint64_t)n_paths * 3 + 64) * 4);
    if (!counts || !bases || !tile_of || !list || !list_base || !keys || !kbig || !prange) return -5;
    uint32_t *pfirst = prange, *plast = prange + n_paths, *gate = prange + 2 * (size_t)n_paths;
    uint32_t gl = stride_grid(L, lines_cap), gs = stride_grid(L, seg_cap);
    unsigned long long* bd_ctr = (unsigned long long*)jh_scratch_get(L.scratch, JH_SCR_BD_CTR, 64);
    if (!bd_ctr) return -5;
    // crossings per path: atomic sums of k_pc_count, so the array is zero BEFORE that kernel starts -- left so by k_pc_rank_small of
    // the frame before, or filled here when the flag is down (first frame, regrown or poisoned scratch)
    uint32_t* ptotal = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_PC_TOT, (uint64_t)n_paths * 4);
    if (!ptotal) return -5;
    const uint32_t ptotal_words = (uint32_t)std::min<uint64_t>(jh_scratch_cap(L.scratch, JH_SCR_PC_TOT) / 4, 0xffffffffull);
    uint32_t* clean = jh_scratch_flags(L.scratch);
    if ((*clean & JH_CLEAN_PC_TOT) == 0u) (void)hipMemsetAsync(ptotal, 0, (size_t)ptotal_words * 4, L.stream);
    *clean &= ~(uint32_t)JH_CLEAN_PC_TOT;
    hipLaunchKernelGGL(k_pc_count, dim3(gl), dim3(JL_WG), 0, L.stream, (const JlBump*)bump, ind, lines, paths, counts, lines_cap, prange,
                       n_paths * 2u + 64u, ptotal, n_paths, bd_ctr, (L.absorb & JH_ABSORB_SETUP) ? (JlIndirectCount*)L.indirect : (JlIndirectCount*)nullptr);
    *jh_scratch_flags(L.scratch) |= JH_CLEAN_BD_CTR;
    int rc = jh_scan_u32(L, counts, 1, bases, lines_cap, &bump->lines, &bump->seg_counts);
    if (rc) return rc;
    const uint32_t *cpf = pfirst, *cpl = plast, *cc = counts, *cb = bases;
    hipLaunchKernelGGL(k_pc_emit, dim3(stride_grid(L, lines_cap, PC_EMIT_GRID_PER_CU)), dim3(JL_WG), 0, L.stream, cfg, (const JlBump*)bump, ind, lines, paths, tile, segc, cb, lines_cap,
                       tile_of, keys, kbig, seg_cap, pfirst, plast, (const uint32_t*)ptotal, n_paths, gate);
    // Big paths only (`gate` = number of tiles if there is one, else 0: the scan then covers 0 elements and the kernels
    // return at once):
    // per-tile list bases = exclusive scan of Tile.segment_count_or_ix, scatter into the lists, rank inside the list.
    rc = jh_scan_u32(L, ((const uint32_t*)tile.p) + 1, 2, list_base, tiles_cap, gate, nullptr);
    if (rc) return rc;
    // the atomics-free ranks of everything else, with the scatter pass of the list route in the same launch
    PcScatterArgs sc;
    sc.tile_of = tile_of; sc.list_base = list_base; sc.tiles_cap = tiles_cap; sc.list = list; sc.kbig = kbig; sc.gate = gate; sc.dense = dense;
    sc.dense_cap = dense_cap; sc.blocks = gs;
    hipLaunchKernelGGL(k_pc_rank_small, dim3(gs + stride_grid(L, (uint64_t)n_paths * 64u)), dim3(JL_WG), 0, L.stream, cfg, (const JlBump*)bump, tile,
                       (const uint32_t*)keys, seg_cap, cpf, cpl, cc, cb, n_paths, segc, sc, ptotal, ptotal_words);
    *clean |= JH_CLEAN_PC_TOT;
    hipLaunchKernelGGL(k_pc_rank, dim3(gs), dim3(JL_WG), 0, L.stream, cfg, (const JlBump*)bump, tile, (const uint2*)tile_of, seg_cap,
                       (const uint32_t*)list_base, tiles_cap, (const uint32_t*)list, segc, (const uint32_t*)kbig, (const uint32_t*)gate,
                       (const uint32_t*)dense, dense_cap);
    return 0;
}

// [config, bump, paths, tiles]
int jh_launch_backdrop_dyn(const JhLaunch& L) {
    if (L.nb < 4) return -1;
    if (L.gx == 0) return 0;
    auto paths = mkbuf<JlPath>(L.b[2].ptr, L.b[2].size);
    auto tiles = mkbuf<JlTile>(L.b[3].ptr, L.b[3].size);
    const uint32_t wide_cap = paths.n;
    // [counter (entries << 40 | rows) | list of (first global row, first tile, width, rows)]
    uint8_t* w = (uint8_t*)jh_scratch_get(L.scratch, JH_SCR_A, 64 + (uint64_t)wide_cap * sizeof(uint4));
    unsigned long long* wide_ctr = (unsigned long long*)jh_scratch_get(L.scratch, JH_SCR_BD_CTR, 64);
    if (!w || !wide_ctr) return -5;
    uint4* wide_list = (uint4*)(w + 64);
    uint32_t* clean = jh_scratch_flags(L.scratch);
    if ((*clean & JH_CLEAN_BD_CTR) == 0u) (void)hipMemsetAsync(wide_ctr, 0, 8, L.stream);  // (path_count did not run in front)
    *clean &= ~(uint32_t)JH_CLEAN_BD_CTR;
    // one thread per row is the better deal only when there are enough paths to fill the device with such threads
    const uint32_t wide_min = L.gx < 64u ? 2u : BD_WIDE;
    hipLaunchKernelGGL(k_backdrop_dyn, dim3(L.gx), dim3(JL_WG), 0, L.stream, (const JlConfig*)L.b[0].ptr, (const JlBump*)L.b[1].ptr, paths, tiles,
                       wide_ctr, wide_list, wide_cap, wide_min);
    // (a wave per BD_UNIT rows; a wide path has tens to hundreds of rows)
    hipLaunchKernelGGL(k_backdrop_wide, dim3(stride_grid(L, (uint64_t)wide_cap * 64u * 64u)), dim3(JL_WG), 0, L.stream, (const JlBump*)L.b[1].ptr, tiles,
                       (const unsigned long long*)wide_ctr, (const uint4*)wide_list, wide_cap);
    return 0;
}

// indirect; [bump, seg_counts, lines, paths, tiles, segments]
int jh_launch_path_tiling(const JhLaunch& L) {
    if (L.nb < 6 || !L.indirect) return -1;
    auto segc = mkbuf<JlSegmentCount>(L.b[1].ptr, L.b[1].size);
    uint32_t g = stride_grid(L, segc.n);
    hipLaunchKernelGGL(k_path_tiling, dim3(g), dim3(JL_WG), 0, L.stream, (const JlBump*)L.b[0].ptr, (const JlIndirectCount*)L.indirect, segc,
                       mkbuf<JlLineSoup>(L.b[2].ptr, L.b[2].size), mkbuf<JlPath>(L.b[3].ptr, L.b[3].size),
                       mkbuf<JlTile>(L.b[4].ptr, L.b[4].size), mkbuf<JlSegment>(L.b[5].ptr, L.b[5].size),
                       (L.absorb & JH_ABSORB_SETUP) ? (JlIndirectCount*)L.indirect : (JlIndirectCount*)nullptr, mkbuf<uint32_t>(L.extra.ptr, L.extra.size));
    return 0;
}
