// kernels_coarse.hip -- K16 coarse (orig/coarse.wgsl:153-462): per 256x256-px bin, merge the binned
// draw objects in draw order, decide per 16x16 tile which of them touch it, and emit the per-tile
// command list (PTCL, shared/ptcl.wgsl) that fine interprets.
//
// MI355X design: the WGSL hands out segment slices, 256-word PTCL chunks and blend-spill space with
// atomicAdd, so seg_data / JUMP targets / blend_ix differ from run to run.  Here coarse runs twice
// over the same templated body:
//   k_coarse<false>  walks every tile's command stream WITHOUT writing it and records, per tile and
//                    summed per workgroup, the segments, PTCL chunk words and blend-spill pixels it
//                    will need;
//   k_coarse<true>   turns them into bases -- exclusive prefixes in (bin, tile-in-bin) order, the
//                    order of the reference's sequential twin (shaders/cpu/cpu.go:1096-1270): the sums
//                    of the workgroups before its own plus a prefix inside its own, no scan launch in
//                    between -- reports the totals in bump.{segments,ptcl,blend}, walks again and
//                    writes PTCL + ~seg_ix.
// => bit-identical PTCL on every run.  A bin is shared by 1 ... 16 workgroups of 256 threads (strips of
// tile rows), one thread per tile in the command walk; bin bitmaps (8 x 256 u32), the batch's element
// records and the Tiles of the current window of elements live in 37 KiB of LDS.
// Algorithmic bytes: 4 B per (draw,bin) bin_data + 32 B Path + 8 B Tile per (draw,tile) + PTCL out.
#include "kcommon.h"

using namespace jk;
using namespace jd;

namespace {

struct Cmd {
    const JlConfig* cfg;
    JlBump* bump;
    Buf<uint32_t> ptcl;
    uint32_t cmd_offset, cmd_limit;
    uint32_t dyn_start;    // first word behind the tiles' initial allocations (coarse.wgsl:75)
    uint32_t chunk_base;   // word offset (relative to dyn_start) of this tile's first chunk
    uint32_t chunk_words;  // PTCL words of dynamic chunks taken so far
    uint32_t seg_base, seg_used;
};

// PTCL words leave as 16-byte stores (dword-aligned addresses: gfx950 runs in unaligned-access mode; a 4-byte store
// per word costs three times the write requests).
struct __attribute__((packed, aligned(4))) PtclQuad { uint32_t a, b, c, d; };
JD void ptcl_wr4(const Buf<uint32_t>& ptcl, uint32_t i, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    if (i + 3u < ptcl.n && i + 3u >= i) {
        PtclQuad q; q.a = a; q.b = b; q.c = c; q.d = d;
        *(PtclQuad*)(ptcl.p + i) = q;
    } else {
        ptcl.wr(i, a); ptcl.wr(i + 1u, b); ptcl.wr(i + 2u, c); ptcl.wr(i + 3u, d);
    }
}

template <bool WRITE>
JD void alloc_cmd(Cmd& c, uint32_t size) {  // coarse.wgsl:70-88
    const bool need = c.cmd_offset + size >= c.cmd_limit;
    uint32_t new_cmd = c.dyn_start + c.chunk_base + c.chunk_words;
    if (WRITE) {
        if (need) {  // (rare: once per 254 words)
            if (new_cmd + JL_PTCL_INCREMENT > c.cfg->ptcl_size) {
                new_cmd = 0u;
                atomicOr(&c.bump->failed, (uint32_t)JL_STAGE_COARSE);
            }
            c.ptcl.wr(c.cmd_offset, JL_CMD_JUMP);
            c.ptcl.wr(c.cmd_offset + 1u, new_cmd);
        }
    }
    c.chunk_words += need ? JL_PTCL_INCREMENT : 0u;
    c.cmd_offset = need ? new_cmd : c.cmd_offset;
    c.cmd_limit = need ? new_cmd + (JL_PTCL_INCREMENT - JL_PTCL_HEADROOM) : c.cmd_limit;
}

#define COARSE_UNROLL 4u
#ifndef COARSE_TILE_CACHE
#define COARSE_TILE_CACHE 1536u  // with the rest of the LDS 37.5 KiB: four workgroups per CU (>= 256: the pairs of one element)
#endif
// Workgroups per CU the bins are split for.  Every workgroup of a bin repeats the merge of the bin's element lists, so more of
// them buy latency with redundant work.  With ONE frame on the device 4 is the optimum (C3 coarse 0.110 ms, C4 0.68; with 2: 0.114
// / 0.76); with TWO frames in flight (bench.py's default) the other frame's kernels fill the idle CUs anyway and the
// redundant work is what counts: 2 gives C3 0.947 -> 0.919 ms per frame and C4 1.90 -> 1.63 (1: 0.913 / 1.59, but +2 % / +9 % for a
// frame on its own).  profiles/r04_variants_in_flight.txt
#ifndef COARSE_WG_PER_CU
#define COARSE_WG_PER_CU 2u
#endif
#ifndef COARSE_MAX_SPLIT
#define COARSE_MAX_SPLIT 16u
#endif

// Element record, word 0 (see stage2 in k_coarse)
#define CM_CLIP 1u             // BEGIN_CLIP or END_CLIP (draw tag bit 0)
#define CM_BLEND 2u            // ... whose blend word is not the plain clip
#define CM_EVENODD_INCLUDE 4u  // even-odd rule in the (draw, tile) include test (draw flags bit 0)
#define CM_EVENODD_FILL 8u     // even-odd rule in the FILL command
#define CM_PATH 16u            // a path command (FILL or SOLID) precedes the brush command
#define CM_BEGIN 32u
#define CM_END 64u
#define CM_NBRUSH_SHIFT 8u     // words of the brush command (0 ... 5)

// The command walk of one tile: its PTCL write position, clip state and the element it looks at next.  A lane can walk
// COARSE_TPL tiles side by side; measured with 2: exactly twice the time per trip (C3 78+60 -> 124+83 us, C4 552+316 ->
// 1121+608 us).  With one walking wave per SIMD the walk is bound by the instructions it issues, not by the latency
// of its LDS chain, so independent chains have nothing to hide in -- what pays is fewer instructions per trip.
#ifndef COARSE_TPL
#define COARSE_TPL 1u
#endif
#if COARSE_TPL != 1
#error "the write pass derives a tile's bases from its thread index: one tile per lane"
#endif
struct Walk {
    Cmd c;
    uint32_t blend_offset, clip_zero_depth, clip_depth, render_blend_depth, max_blend_depth;
    uint32_t tile_x, tile_y, my_xy, slot;
    bool has_tile;
    uint32_t slice_ix, bitmap, nz, el_next;  // nz: slices behind slice_ix that hold elements of this tile
    uint4 q0, q2;
    JlTile tile;
};

// CLIPS = false: instantiation for scenes without clip layers (ConfigUniform.n_clip == 0): no BEGIN/END_CLIP draw
// objects can occur, which removes the clip-depth state and half of the divergent control flow of the command walk.
template <bool WRITE, bool CLIPS>
__global__ __launch_bounds__(JL_WG) void k_coarse(const JlConfig* __restrict__ cfg, Buf<uint32_t> scene, Buf<JlDrawMonoid> draw_monoids,
                                                  Buf<JlBinHeader> bin_headers, Buf<uint32_t> info_bin_data, Buf<JlPath> paths, Buf<JlTile> tiles,
                                                  JlBump* __restrict__ bump, Buf<uint32_t> ptcl, uint32_t* __restrict__ cnt_seg,
                                                  uint32_t* __restrict__ cnt_chunk, uint32_t* __restrict__ cnt_blend,
                                                  uint32_t* __restrict__ wg_tot, uint32_t n_wg, uint32_t bin_row0, uint32_t split) {
    // cnt_*[slot]: what the counting pass found per tile; wg_tot[c * n_wg + wg]: their sums per workgroup, wg = bin * split
    // + strip -- the canonical (bin, tile) order is workgroup-major, so the write pass gets a tile's bases as (sum over
    // the workgroups before its own) + (exclusive prefix inside its own): it scans for itself, no scan launches between.
    // bin_row0: first bin row of the launch (band mode writes the PTCL of its band only; the counting pass always
    // covers the whole target, so that every allocation base is the one of the unsharded run)
    // split (1 ... 16): a bin is shared by `split` workgroups (blockIdx.z), each owning 16 / split of its tile rows.
    // The merge of the bin's element lists is repeated by each of them (cheap); the (draw, tile) include test and the
    // per-tile command walk -- the expensive parts -- cover the workgroup's rows only.  With one workgroup per bin a
    // 2048^2 target keeps 64 of 256 CUs busy and a 4096^2 target one wave per SIMD.
    const uint32_t bin_y = blockIdx.y + bin_row0;
    const uint32_t part_rows = JL_N_TILE_Y / split, part_y0 = blockIdx.z * part_rows, part_y1 = part_y0 + part_rows;
    __shared__ uint32_t sh_bitmaps[8][JL_N_TILE];
    __shared__ uint32_t sh_part_count[JL_WG];
    __shared__ uint32_t sh_part_offsets[JL_WG];
    __shared__ uint32_t sh_drawobj_ix[JL_WG];
    __shared__ uint32_t sh_tile_count[JL_WG];
    // per-batch draw object data staged once by the draw's own thread, so that the per-(draw,tile) include test and the
    // serial per-tile command walk read LDS instead of chasing scene / draw_monoid / info pointers through HBM; packed
    // so that one walk step is three 16-byte LDS reads issued together:
    __shared__ uint4 sh_r0[JL_WG];  // tag, draw flags, tile base (of bin-relative tile (0,0)), tile stride
    __shared__ uint4 sh_r1[JL_WG];  // x0 | y0 << 16, width, first (draw, tile) pair of the draw in the batch, info offset
    __shared__ uint4 sh_r2[JL_WG];  // scene[dd .. dd+3]: colour / ramp index / blend+alpha
    __shared__ uint32_t sh_scan[8];
    __shared__ uint32_t sh_red[12];
    // (backdrop, segment count) of the (draw, tile) pairs of the current window of elements.  The command walk reads Tiles
    // from here ONLY: a global load inside its loop makes every trip wait for the PTCL stores of the trip before
    // (vmcnt counts loads and stores in one order) -- 1 us per trip in the write pass.
    __shared__ uint2 sh_tile_cache[COARSE_TILE_CACHE];

    const uint32_t lid = threadIdx.x;
    const uint32_t width_in_bins = (cfg->width_in_tiles + JL_N_TILE_X - 1u) / JL_N_TILE_X;
    const uint32_t bin_ix = width_in_bins * bin_y + blockIdx.x;
    if (bin_ix * split + blockIdx.z >= n_wg) return;  // (uniform) a ConfigUniform that contradicts the dispatch: nothing to index the scratch with
    // the first part_tiles / COARSE_TPL threads walk COARSE_TPL tiles each: tiles t = lid + k * walkers of the workgroup's part
    const uint32_t part_tiles = part_rows * JL_N_TILE_X, walkers = part_tiles / COARSE_TPL;
    Walk W[COARSE_TPL];
#pragma unroll
    for (uint32_t k = 0; k < COARSE_TPL; k++) {
        W[k].has_tile = lid < walkers;
        const uint32_t t = W[k].has_tile ? lid + k * walkers : 0u;
        W[k].tile_x = t % JL_N_TILE_X;
        W[k].tile_y = part_y0 + t / JL_N_TILE_X;
        W[k].my_xy = W[k].tile_y * JL_N_TILE_X + W[k].tile_x;  // the tile's index inside the bin
        W[k].slot = bin_ix * JL_N_TILE + W[k].my_xy;           // position in the canonical (bin, tile) order
    }

    {  // coarse.wgsl:161-176
        uint32_t failed = bump->failed & (JL_STAGE_BINNING | JL_STAGE_TILE_ALLOC | JL_STAGE_FLATTEN);
        if (bump->seg_counts > cfg->seg_counts_size) failed |= JL_STAGE_PATH_COUNT;
        if (failed != 0u) {
            if (WRITE) {
                if (blockIdx.x == 0u && blockIdx.y == 0u && lid == 0u) atomicOr(&bump->failed, failed);
            } else {
#pragma unroll
                for (uint32_t k = 0; k < COARSE_TPL; k++)
                    if (W[k].has_tile) { cnt_seg[W[k].slot] = 0u; cnt_chunk[W[k].slot] = 0u; cnt_blend[W[k].slot] = 0u; }
                if (lid < 3u) wg_tot[lid * n_wg + bin_ix * split + blockIdx.z] = 0u;
            }
            return;
        }
    }
    const uint32_t n_partitions = (cfg->layout.n_drawobj + JL_N_TILE - 1u) / JL_N_TILE;
    const uint32_t bin_tile_x = JL_N_TILE_X * blockIdx.x;
    const uint32_t bin_tile_y = JL_N_TILE_Y * bin_y;
    const uint32_t BLEND_CLIP = (128u << 8) | 0u;  // MIX_CLIP << 8 | COMPOSE_SRC_OVER (Jello numbering, blend.wgsl:199-202)
    uint32_t my_base_seg = 0u, my_base_chunk = 0u, my_base_blend = 0u;
    if (WRITE) {
        const uint32_t my_wg = bin_ix * split + blockIdx.z;
        MonoidK<3> before, all;
#pragma unroll
        for (int c = 0; c < 3; c++) { before.v[c] = 0u; all.v[c] = 0u; }
        const bool totals = blockIdx.x == 0u && blockIdx.y == 0u && blockIdx.z == 0u;  // (uniform) this workgroup also reports the frame's totals
        for (uint32_t j = lid; j < (totals ? n_wg : my_wg); j += JL_WG) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const uint32_t v = wg_tot[(uint32_t)c * n_wg + j];
                all.v[c] += v;
                if (j < my_wg) before.v[c] += v;
            }
        }
        const MonoidK<3> carry = block_reduce_monoid<3>(before, sh_red);
        __syncthreads();
        if (totals) {
            const MonoidK<3> t = block_reduce_monoid<3>(all, sh_red);
            if (lid == 0u) { bump->segments = t.v[0]; bump->ptcl = t.v[1]; bump->blend = t.v[2]; }
            __syncthreads();
        }
        const bool mine = W[0].has_tile;
        uint32_t tot;
        my_base_seg = carry.v[0] + block_excl_scan_u32(mine ? cnt_seg[W[0].slot] : 0u, sh_scan, &tot);
        __syncthreads();
        my_base_chunk = carry.v[1] + block_excl_scan_u32(mine ? cnt_chunk[W[0].slot] : 0u, sh_scan, &tot);
        __syncthreads();
        my_base_blend = carry.v[2] + block_excl_scan_u32(mine ? cnt_blend[W[0].slot] : 0u, sh_scan, &tot);
        __syncthreads();
    }
#pragma unroll
    for (uint32_t k = 0; k < COARSE_TPL; k++) {
        Walk& w = W[k];
        const uint32_t this_tile_ix = (bin_tile_y + w.tile_y) * cfg->width_in_tiles + bin_tile_x + w.tile_x;
        w.c.cfg = cfg; w.c.bump = bump; w.c.ptcl = ptcl;
        w.c.cmd_offset = this_tile_ix * JL_PTCL_INITIAL_ALLOC;
        w.c.cmd_limit = w.c.cmd_offset + (JL_PTCL_INITIAL_ALLOC - JL_PTCL_HEADROOM);
        w.c.dyn_start = cfg->width_in_tiles * cfg->height_in_tiles * JL_PTCL_INITIAL_ALLOC;
        w.c.chunk_base = my_base_chunk;
        w.c.chunk_words = 0u;
        w.c.seg_base = my_base_seg;
        w.c.seg_used = 0u;
        w.clip_zero_depth = 0u; w.clip_depth = 0u; w.render_blend_depth = 0u; w.max_blend_depth = 0u;
        w.blend_offset = w.c.cmd_offset;
        w.c.cmd_offset += 1u;
        w.slice_ix = 7u; w.bitmap = 0u; w.nz = 0u; w.el_next = 0xffffffffu;
        w.q0 = make_uint4(0u, 0u, 0u, 0u); w.q2 = w.q0;
        w.tile.backdrop = 0; w.tile.segment_count_or_ix = 0u;
    }
    uint32_t partition_ix = 0u, rd_ix = 0u, wr_ix = 0u, part_start_ix = 0u, ready_ix = 0u;

    // The batch loop is software-pipelined: an element's record needs three dependent memory round trips (bin_data ->
    // tag / draw monoid -> info, draw data, path), which used to sit in front of every batch.  Now the next batch is
    // gathered and its first-level loads are issued before this batch's include test, its second-level loads before
    // this batch's command walk, and the record is complete when the walk is.
    auto gather = [&]() -> uint32_t {  // coarse.wgsl:201-241: this thread's element of the next batch (~0u: none)
        for (;;) {
            if (ready_ix == wr_ix && partition_ix < n_partitions) {
                part_start_ix = ready_ix;
                uint32_t count = 0u;
                if (partition_ix + lid < n_partitions) {
                    uint32_t in_ix = (partition_ix + lid) * JL_N_TILE + bin_ix;
                    JlBinHeader bh = bin_headers.rd(in_ix);
                    count = bh.element_count;
                    sh_part_offsets[lid] = bh.chunk_offset;
                }
                uint32_t tot;
                uint32_t excl = block_excl_scan_u32(count, sh_scan, &tot);
                sh_part_count[lid] = part_start_ix + excl + count;
                __syncthreads();
                ready_ix = sh_part_count[JL_WG - 1u];
                partition_ix += JL_WG;
            }
            uint32_t ix = rd_ix + lid;
            if (ix >= wr_ix && ix < ready_ix) {
                uint32_t part_ix = 0u;
                for (uint32_t i = 0; i < 8u; i++) {
                    uint32_t probe = part_ix + (128u >> i);
                    if (ix >= sh_part_count[probe - 1u]) part_ix = probe;
                }
                ix -= (part_ix > 0u) ? sh_part_count[part_ix - 1u] : part_start_ix;
                uint32_t offset = cfg->layout.bin_data_start + sh_part_offsets[part_ix];
                sh_drawobj_ix[lid] = info_bin_data.rd(offset + ix);
            }
            wr_ix = umin_(rd_ix + JL_N_TILE, ready_ix);
            if (wr_ix - rd_ix >= JL_N_TILE || (wr_ix >= ready_ix && partition_ix >= n_partitions)) break;
            __syncthreads();
        }
        // sh_drawobj_ix[0 .. wr_ix - rd_ix) holds the merged binning results of the batch (every thread its own slot).
        return (lid + rd_ix < wr_ix) ? sh_drawobj_ix[lid] : 0xffffffffu;
    };
    // first-level loads of an element: tag and draw monoid
    uint32_t n_tag = JL_DRAWTAG_NOP;
    JlDrawMonoid n_dm;
    n_dm.path_ix = 0u; n_dm.clip_ix = 0u; n_dm.scene_offset = 0u; n_dm.info_offset = 0u;
    auto stage1 = [&](uint32_t obj) {
        n_tag = JL_DRAWTAG_NOP;
        if (obj != 0xffffffffu) {
            n_tag = scene.rd(cfg->layout.drawtag_base + obj);
            n_dm = draw_monoids.rd(obj);
        }
    };
    // second-level loads and the record (r1.z is filled in after the batch's tile-count scan)
    uint4 n_r0 = make_uint4(0u, 0u, 0u, 0u), n_r1 = make_uint4(0u, 0u, 0u, 0u), n_r2 = make_uint4(0u, 0u, 0u, 0u);
    uint32_t n_tile_count = 0u;
    // The record is laid out for the command walk, which runs once per (element, tile) and must not branch on the draw
    // tag: the tag is decoded HERE, once per element, into a few flag bits and the words of the brush command.
    //   r0 = (meta, brush word 0, tile base of bin-relative tile (0,0), tile stride)
    //   r1 = (x0 | y0 << 16, width | ceil(2^16 / width) << 5, first (draw, tile) pair of the draw in the batch, -)
    //   r2 = brush words 1..4
    auto stage2 = [&]() {
        const uint32_t tag = n_tag;
        n_tile_count = 0u;
        n_r0 = make_uint4(0u, 0u, 0u, 0u); n_r1 = make_uint4(0u, 0u, 0u, 0u); n_r2 = make_uint4(0u, 0u, 0u, 0u);
        if (tag != JL_DRAWTAG_NOP) {
            const JlDrawMonoid dm0 = n_dm;
            uint32_t path_ix = dm0.path_ix;
            uint32_t dd0 = cfg->layout.drawdata_base + dm0.scene_offset;
            const uint32_t di = dm0.info_offset;
            const uint32_t draw_flags = info_bin_data.rd(di);
            const uint4 sc = make_uint4(scene.rd(dd0), scene.rd(dd0 + 1u), scene.rd(dd0 + 2u), scene.rd(dd0 + 3u));
            uint32_t meta = (draw_flags & 1u) != 0u ? CM_EVENODD_INCLUDE : 0u;
            if ((tag & 1u) != 0u) {
                meta |= CM_CLIP;
                if (sc.x != BLEND_CLIP) meta |= CM_BLEND;
            }
            if (tag == JL_DRAWTAG_FILL_COLOR) {
                meta |= CM_PATH | (meta & CM_EVENODD_INCLUDE ? CM_EVENODD_FILL : 0u) | (5u << CM_NBRUSH_SHIFT);
                n_r0.y = JL_CMD_COLOR; n_r2 = sc;
            } else if (tag == JL_DRAWTAG_FILL_LIN_GRADIENT || tag == JL_DRAWTAG_FILL_RAD_GRADIENT || tag == JL_DRAWTAG_FILL_SWEEP_GRADIENT) {
                meta |= CM_PATH | (meta & CM_EVENODD_INCLUDE ? CM_EVENODD_FILL : 0u) | (3u << CM_NBRUSH_SHIFT);
                n_r0.y = tag == JL_DRAWTAG_FILL_LIN_GRADIENT ? JL_CMD_LIN_GRAD : (tag == JL_DRAWTAG_FILL_RAD_GRADIENT ? JL_CMD_RAD_GRAD : JL_CMD_SWEEP_GRAD);
                n_r2.x = sc.x; n_r2.y = di + 1u;
            } else if (tag == JL_DRAWTAG_FILL_IMAGE) {
                meta |= CM_PATH | (meta & CM_EVENODD_INCLUDE ? CM_EVENODD_FILL : 0u) | (2u << CM_NBRUSH_SHIFT);
                n_r0.y = JL_CMD_IMAGE; n_r2.x = di + 1u;
            } else if (CLIPS && tag == JL_DRAWTAG_BEGIN_CLIP) {
                meta |= CM_BEGIN | (1u << CM_NBRUSH_SHIFT);
                n_r0.y = JL_CMD_BEGIN_CLIP;
            } else if (CLIPS && tag == JL_DRAWTAG_END_CLIP) {
                meta |= CM_END | CM_PATH | (3u << CM_NBRUSH_SHIFT);  // (its path command is never even-odd: coarse.wgsl:423)
                n_r0.y = JL_CMD_END_CLIP; n_r2.x = sc.x; n_r2.y = sc.y;
            }
            n_r0.x = meta;
            JlPath path = paths.rd(path_ix);
            uint32_t stride = path.bbox[2] - path.bbox[0];
            n_r0.w = stride;
            int32_t dx = (int32_t)path.bbox[0] - (int32_t)bin_tile_x;
            int32_t dy = (int32_t)path.bbox[1] - (int32_t)bin_tile_y;
            int32_t x0 = iclamp_(dx, 0, JL_N_TILE_X);
            int32_t y0 = iclamp_(dy, (int32_t)part_y0, (int32_t)part_y1);
            int32_t x1 = iclamp_((int32_t)path.bbox[2] - (int32_t)bin_tile_x, 0, JL_N_TILE_X);
            int32_t y1 = iclamp_((int32_t)path.bbox[3] - (int32_t)bin_tile_y, (int32_t)part_y0, (int32_t)part_y1);
            {   // width (<= 16) and ceil(2^16 / width): the include test divides pair indices (< 4096) by the width
                const uint32_t wdt = (uint32_t)(x1 - x0);
                n_r1.y = wdt | ((wdt ? (65536u + wdt - 1u) / wdt : 0u) << 5);
            }
            n_r1.x = (uint32_t)x0 | ((uint32_t)y0 << 16);
            n_tile_count = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
            n_r0.z = path.tiles - (uint32_t)(dy * (int32_t)stride + dx);
        }
    };
    stage1(gather());
    stage2();
    for (;;) {
        for (uint32_t i = 0; i < 8u; i++) sh_bitmaps[i][lid] = 0u;
        const uint32_t tile_count = n_tile_count;
        uint4 r0 = n_r0, r1 = n_r1, r2 = n_r2;
        uint32_t total_tile_count;
        uint32_t excl_tc = block_excl_scan_u32(tile_count, sh_scan, &total_tile_count);
        sh_tile_count[lid] = excl_tc + tile_count;
        r1.z = excl_tc;
        sh_r0[lid] = r0; sh_r1[lid] = r1; sh_r2[lid] = r2;
        __syncthreads();
        rd_ix += JL_N_TILE;
        const bool has_next = !(rd_ix >= ready_ix && partition_ix >= n_partitions);  // uniform
        if (has_next) stage1(gather());
        // A batch is worked on in windows of elements whose (draw, tile) pairs fit the Tile cache -- one window unless
        // the batch holds large paths (C4's clip rectangles); include test and command walk alternate per window.
        uint32_t win_e0 = 0u, win_p0 = 0u;
        bool did_stage2 = false;
        for (;;) {
        uint32_t win_e1 = win_e0;  // largest e1 with sh_tile_count[e1 - 1] - win_p0 <= COARSE_TILE_CACHE (uniform)
        for (uint32_t step = 256u; step > 0u; step >>= 1)
            if (win_e1 + step <= JL_N_TILE && sh_tile_count[win_e1 + step - 1u] - win_p0 <= COARSE_TILE_CACHE) win_e1 += step;
        const uint32_t win_p1 = sh_tile_count[win_e1 - 1u];
        // (draw, tile) include test, coarse.wgsl:318-341.  The Tile loads are issued four at a time per thread instead
        // of one dependent load per iteration, and what they return is kept in LDS for the command walk below.
        for (uint32_t base = win_p0; base < win_p1; base += COARSE_UNROLL * JL_N_TILE) {
            uint32_t p_el[COARSE_UNROLL], p_xy[COARSE_UNROLL], p_tile[COARSE_UNROLL];
            JlTile p_t[COARSE_UNROLL];
#pragma unroll
            for (uint32_t u = 0; u < COARSE_UNROLL; u++) {
                const uint32_t ix = base + u * JL_N_TILE + lid;
                p_el[u] = 0xffffffffu; p_xy[u] = 0u; p_tile[u] = 0u;
                p_t[u].backdrop = 0; p_t[u].segment_count_or_ix = 0u;
                if (ix < win_p1) {
                    uint32_t el_ix = 0u;
#pragma unroll
                    for (uint32_t i = 0; i < 8u; i++) {
                        uint32_t probe = el_ix + (128u >> i);
                        if (ix >= sh_tile_count[probe - 1u]) el_ix = probe;
                    }
                    const uint4 q0 = sh_r0[el_ix], q1 = sh_r1[el_ix];
                    uint32_t seq_ix = ix - q1.z;
                    uint32_t width = q1.y & 31u;
                    uint32_t row = (seq_ix * (q1.y >> 5)) >> 16;  // seq_ix / width, exact for seq_ix < 4096 and width <= 16
                    uint32_t x = (q1.x & 0xffffu) + (seq_ix - row * width);
                    uint32_t y = (q1.x >> 16) + row;
                    p_el[u] = el_ix;
                    p_xy[u] = y * JL_N_TILE_X + x;
                    p_tile[u] = q0.z + q0.w * y + x;
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < COARSE_UNROLL; u++)
                if (p_el[u] != 0xffffffffu) p_t[u] = tiles.rd(p_tile[u]);
#pragma unroll
            for (uint32_t u = 0; u < COARSE_UNROLL; u++) {
                if (p_el[u] == 0xffffffffu) continue;
                const uint32_t ix = base + u * JL_N_TILE + lid;
                const uint32_t el_ix = p_el[u];
                const JlTile tile = p_t[u];
                sh_tile_cache[ix - win_p0] = make_uint2((uint32_t)tile.backdrop, tile.segment_count_or_ix);
                const uint32_t meta = sh_r0[el_ix].x;
                const bool is_clip = (meta & CM_CLIP) != 0u, is_blend = (meta & CM_BLEND) != 0u;
                const bool even_odd = (meta & CM_EVENODD_INCLUDE) != 0u;
                uint32_t n_segs = tile.segment_count_or_ix;
                int32_t bd = tile.backdrop;
                int32_t absbd = bd < 0 ? (int32_t)(0u - (uint32_t)bd) : bd;
                bool backdrop_clear = (even_odd ? (absbd & 1) : bd) == 0;
                bool include_tile = n_segs != 0u || (backdrop_clear == is_clip) || is_blend;
                if (include_tile) atomicOr(&sh_bitmaps[el_ix / 32u][p_xy[u]], 1u << (el_ix & 31u));
            }
        }
        if (has_next && !did_stage2) { stage2(); did_stage2 = true; }
        __syncthreads();
        // Write the per-tile command lists (coarse.wgsl:344-444).  The reads of a tile's NEXT element (bitmap -> record ->
        // cached Tile) are issued before the commands of the current one are written.
        auto next_el = [&](Walk& w) -> uint32_t {  // next set bit of the tile's bitmaps, ~0u at the end
            if (w.bitmap == 0u) {  // on to the next slice that holds something (a tile of C3 sees 10 of a batch's 256 elements)
                if (w.nz == 0u) return 0xffffffffu;
                w.slice_ix = (uint32_t)__builtin_ctz(w.nz);
                w.nz &= w.nz - 1u;
                w.bitmap = sh_bitmaps[w.slice_ix][w.my_xy];
            }
            const uint32_t e = w.slice_ix * 32u + (uint32_t)__builtin_ctz(w.bitmap);
            w.bitmap &= w.bitmap - 1u;
            return e;
        };
        auto fetch = [&](Walk& w) {  // record and Tile of element el_next for this tile (nothing useful if there is none)
            const uint32_t e = w.el_next & (JL_N_TILE - 1u);
            const uint4 q1 = sh_r1[e];
            w.q0 = sh_r0[e]; w.q2 = sh_r2[e];
            // the pair's slot in the include-test order: what that pass loaded is in LDS
            const uint32_t pair = q1.z + (w.tile_y - (q1.x >> 16)) * (q1.y & 31u) + (w.tile_x - (q1.x & 0xffffu));
            const uint2 tc = sh_tile_cache[umin_(pair - win_p0, COARSE_TILE_CACHE - 1u)];
            w.tile.backdrop = (int32_t)tc.x;
            w.tile.segment_count_or_ix = tc.y;
        };
        bool any = false;
#pragma unroll
        for (uint32_t k = 0; k < COARSE_TPL; k++) {
            Walk& w = W[k];
            // (bits of later windows are not set yet: the walk of a window ends by itself at win_e1)
            const bool walks = w.has_tile && win_e0 < JL_N_TILE;
            w.slice_ix = walks ? win_e0 / 32u : 7u;
            uint32_t sl[8];
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) sl[i] = sh_bitmaps[i][w.my_xy];
            w.bitmap = 0u; w.nz = 0u;
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) {
                if (walks && i == w.slice_ix) w.bitmap = sl[i] & (0xffffffffu << (win_e0 & 31u));
                if (walks && i > w.slice_ix && sl[i] != 0u) w.nz |= 1u << i;
            }
            w.el_next = next_el(w);
            fetch(w);
            any = any || w.el_next != 0xffffffffu;
        }
        // One trip per element and tile.  The body is written with selects, not branches: a wave's tiles walk different
        // elements (fills with and without segments, clips that are open, empty or skipped), and as a tree of divergent
        // branches a trip cost ~1500 cycles of exec-mask bookkeeping for ~30 useful instructions.  A tile that has run
        // out of elements goes through the motions with an empty record (meta 0: no command, no state change).
        while (any) {
            any = false;
#pragma unroll
            for (uint32_t k = 0; k < COARSE_TPL; k++) {
                Walk& w = W[k];
                Cmd& c = w.c;
                const bool live = w.el_next != 0xffffffffu;
                const uint4 q0 = w.q0, q2 = w.q2;
                const JlTile tile = w.tile;
                if (live) w.el_next = next_el(w);
                fetch(w);
                any = any || w.el_next != 0xffffffffu;
                // Everything below is 0/1 integer arithmetic on vector registers: as bools the compiler keeps the
                // conditions in scalar mask registers, and the scalar <-> vector hand-overs cost more than the ops.
                const uint32_t meta = live ? q0.x : 0u;
                const uint32_t n_segs = tile.segment_count_or_ix;
                const uint32_t is_begin = CLIPS ? (meta / CM_BEGIN) & 1u : 0u, is_end = CLIPS ? (meta / CM_END) & 1u : 0u;
                const uint32_t has_path = (meta / CM_PATH) & 1u;
                const uint32_t active = CLIPS ? 1u - umin_(w.clip_zero_depth, 1u) : 1u;  // coarse.wgsl:352
                const uint32_t has_segs = umin_(n_segs, 1u);
                const uint32_t zero_tile = 1u - umin_(n_segs | (uint32_t)tile.backdrop, 1u);
                if (CLIPS) {  // clip state, coarse.wgsl:398-441
                    const uint32_t open = is_begin & active;  // BEGIN_CLIP seen by a live tile: skipped from here if the clip is empty, ...
                    const uint32_t opened = open & (1u - zero_tile);  // ... else one more blend level
                    // (clip_zero_depth is 0 whenever `active`; END_CLIP of a skipped stretch ends it at its own depth)
                    const uint32_t ends_skip = is_end & (1u - active) & (1u - umin_(w.clip_depth ^ w.clip_zero_depth, 1u));
                    w.clip_zero_depth = (w.clip_zero_depth + (open & zero_tile) * (w.clip_depth + 1u)) * (1u - ends_skip);
                    w.render_blend_depth += opened;
                    w.max_blend_depth = umax_(w.max_blend_depth, w.render_blend_depth);
                    w.render_blend_depth -= is_end & active;
                    w.clip_depth += is_begin - is_end;
                }
                const uint32_t emit_path = active & has_path;
                const uint32_t emit_brush = active & (1u - (is_begin & zero_tile));
                // path command: FILL (4 words) if the tile has segments, else SOLID (1 word); coarse.wgsl:90-112
                const uint32_t s1 = emit_path * (1u + 3u * has_segs);
                const uint32_t seg_ix = c.seg_base + c.seg_used;
                c.seg_used += emit_path * n_segs;
                alloc_cmd<WRITE>(c, s1);  // (no-op for s1 == 0: cmd_offset < cmd_limit between commands)
                // Stores: a command of fewer than four words is written as four -- the words behind it belong to this
                // tile's chunk (cmd_offset + 1 < cmd_limit and the two words of headroom) and are either overwritten
                // by the next command or never reached -- so that a trip has three predicated stores instead of eight.
                if (WRITE) {
                    const bool room = c.cmd_offset + 8u <= c.ptcl.n && c.cmd_offset + 8u > c.cmd_offset;
                    const uint32_t rule = (n_segs << 1) | ((meta / CM_EVENODD_FILL) & 1u);
                    if (s1 == 4u) {
                        const uint32_t tile_ix = q0.z + q0.w * w.tile_y + w.tile_x;
                        if (tiles.ok(tile_ix)) tiles.p[tile_ix].segment_count_or_ix = ~seg_ix;
                    }
                    if (s1 != 0u) {
                        PtclQuad q; q.a = has_segs ? JL_CMD_FILL : JL_CMD_SOLID; q.b = rule; q.c = seg_ix; q.d = (uint32_t)tile.backdrop;
                        if (room) *(PtclQuad*)(c.ptcl.p + c.cmd_offset) = q;
                        else if (s1 == 4u) ptcl_wr4(c.ptcl, c.cmd_offset, q.a, q.b, q.c, q.d);
                        else c.ptcl.wr(c.cmd_offset, q.a);
                    }
                }
                c.cmd_offset += s1;
                // brush command: the words stage2 prepared
                const uint32_t s2 = emit_brush * ((meta >> CM_NBRUSH_SHIFT) & 7u);
                alloc_cmd<WRITE>(c, s2);
                if (WRITE) {
                    const bool room = c.cmd_offset + 8u <= c.ptcl.n && c.cmd_offset + 8u > c.cmd_offset;
                    if (s2 != 0u) {
                        if (room) {
                            PtclQuad q; q.a = q0.y; q.b = q2.x; q.c = q2.y; q.d = q2.z;
                            *(PtclQuad*)(c.ptcl.p + c.cmd_offset) = q;
                        } else {
                            c.ptcl.wr(c.cmd_offset, q0.y);
                            if (s2 >= 2u) c.ptcl.wr(c.cmd_offset + 1u, q2.x);
                            if (s2 >= 3u) c.ptcl.wr(c.cmd_offset + 2u, q2.y);
                            if (s2 >= 4u) c.ptcl.wr(c.cmd_offset + 3u, q2.z);
                        }
                    }
                    if (s2 == 5u) c.ptcl.wr(c.cmd_offset + 4u, q2.w);
                }
                c.cmd_offset += s2;
            }
        }
        if (win_p1 >= total_tile_count) break;
        __syncthreads();  // the next window's include test overwrites the Tile cache
        win_e0 = win_e1; win_p0 = win_p1;
        }
        if (!has_next) break;
        __syncthreads();
    }
    MonoidK<3> my_tot;
    my_tot.v[0] = 0u; my_tot.v[1] = 0u; my_tot.v[2] = 0u;
#pragma unroll
    for (uint32_t k = 0; k < COARSE_TPL; k++) {
        Walk& w = W[k];
        if (!w.has_tile) continue;
        uint32_t scratch_size = 0u;
        const bool in_target = bin_tile_x + w.tile_x < cfg->width_in_tiles && bin_tile_y + w.tile_y < cfg->height_in_tiles;
        if (in_target && w.max_blend_depth > JL_BLEND_STACK_SPLIT) scratch_size = (w.max_blend_depth - JL_BLEND_STACK_SPLIT) * JL_TILE_WIDTH * JL_TILE_HEIGHT;
        if (WRITE) {
            if (in_target) {
                w.c.ptcl.wr(w.c.cmd_offset, JL_CMD_END);
                uint32_t blend_ix = 0u;
                if (scratch_size != 0u) {
                    blend_ix = my_base_blend;
                    if (blend_ix + scratch_size > cfg->blend_size) atomicOr(&bump->failed, (uint32_t)JL_STAGE_COARSE);
                }
                w.c.ptcl.wr(w.blend_offset, blend_ix);
            }
        } else {
            cnt_seg[w.slot] = w.c.seg_used;
            cnt_chunk[w.slot] = w.c.chunk_words;
            cnt_blend[w.slot] = scratch_size;
            my_tot.v[0] = w.c.seg_used; my_tot.v[1] = w.c.chunk_words; my_tot.v[2] = scratch_size;
        }
    }
    if (!WRITE) {
        const MonoidK<3> t = block_reduce_monoid<3>(my_tot, sh_red);
        if (lid < 3u) wg_tot[lid * n_wg + bin_ix * split + blockIdx.z] = lid == 0u ? t.v[0] : (lid == 1u ? t.v[1] : t.v[2]);
    }
}

// The frame's totals when the band of the write pass is empty (otherwise its first workgroup reports them).
__global__ __launch_bounds__(JL_WG) void k_coarse_totals(const uint32_t* __restrict__ wg_tot, uint32_t n_wg, JlBump* __restrict__ bump) {
    __shared__ uint32_t sh_red[12];
    MonoidK<3> all;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        all.v[c] = 0u;
        for (uint32_t j = threadIdx.x; j < n_wg; j += JL_WG) all.v[c] += wg_tot[(uint32_t)c * n_wg + j];
    }
    const MonoidK<3> t = block_reduce_monoid<3>(all, sh_red);
    if (threadIdx.x == 0u) { bump->segments = t.v[0]; bump->ptcl = t.v[1]; bump->blend = t.v[2]; }
}

}  // namespace

// [config, scene, draw_monoids, bin_headers, info_bin_data, paths, tiles, bump, ptcl]
int jh_launch_coarse(const JhLaunch& L) {
    if (L.nb < 9) return -1;
    if (L.gx == 0 || L.gy == 0) return 0;
    uint32_t n = L.gx * L.gy * JL_N_TILE;
    auto cfg = (const JlConfig*)L.b[0].ptr;
    auto scene = mkbuf<uint32_t>(L.b[1].ptr, L.b[1].size);
    auto dm = mkbuf<JlDrawMonoid>(L.b[2].ptr, L.b[2].size);
    auto bh = mkbuf<JlBinHeader>(L.b[3].ptr, L.b[3].size);
    auto ibd = mkbuf<uint32_t>(L.b[4].ptr, L.b[4].size);
    auto paths = mkbuf<JlPath>(L.b[5].ptr, L.b[5].size);
    auto tiles = mkbuf<JlTile>(L.b[6].ptr, L.b[6].size);
    JlBump* bump = (JlBump*)L.b[7].ptr;
    auto ptcl = mkbuf<uint32_t>(L.b[8].ptr, L.b[8].size);
    // workgroups per bin: enough to give every CU COARSE_WG_PER_CU workgroups (the LDS of one allows four per CU)
    const uint32_t want = COARSE_WG_PER_CU * (uint32_t)(L.num_cus > 0 ? L.num_cus : 256);
    uint32_t split = 1u;
    while (split < COARSE_MAX_SPLIT && L.gx * L.gy * split < want) split *= 2u;
    const uint32_t n_wg = L.gx * L.gy * split;
    uint32_t* scr = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_A, ((uint64_t)n + n_wg) * 4 * 3);
    if (!scr) return -5;
    uint32_t *cnt_seg = scr, *cnt_chunk = scr + n, *cnt_blend = scr + 2 * (size_t)n, *wg_tot = scr + 3 * (size_t)n;
    dim3 grid(L.gx, L.gy, split), blk(JL_WG);
    const uint32_t row0 = L.band_row0 < L.gy ? L.band_row0 : L.gy, row1 = L.band_row1 < L.gy ? L.band_row1 : L.gy;
    dim3 grid_w(L.gx, row1 > row0 ? row1 - row0 : 0u, split);
    const bool clips = !(L.cfg_host && L.cfg_host->layout.n_clip == 0u);  // host shadow of the uploaded ConfigUniform
#define JH_COARSE(W, C, G, ROW0) hipLaunchKernelGGL((k_coarse<W, C>), G, blk, 0, L.stream, cfg, scene, dm, bh, ibd, paths, tiles, bump, ptcl, cnt_seg, cnt_chunk, cnt_blend, wg_tot, n_wg, ROW0, split)
    if (clips) JH_COARSE(false, true, grid, 0u); else JH_COARSE(false, false, grid, 0u);
    if (grid_w.y == 0u) {  // an empty band: nobody to report the totals (otherwise the write pass's first workgroup does)
        hipLaunchKernelGGL(k_coarse_totals, dim3(1), blk, 0, L.stream, (const uint32_t*)wg_tot, n_wg, bump);
        return 0;
    }
    if (clips) JH_COARSE(true, true, grid_w, row0); else JH_COARSE(true, false, grid_w, row0);
#undef JH_COARSE
    return 0;
}
