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
#include <cstring>

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
    // MODE 2 (one walk + relocation): `ptcl` is the scratch copy of the PTCL, chunks come from an arena in walk order
    uint32_t* arena_ctr;            // words of the arena handed out so far
    uint2* owner;                   // per arena chunk: (tile slot, ordinal of the chunk in its tile's stream)
    unsigned long long* masks;      // per 64 words of the scratch PTCL: [0] seg_ix words of FILL commands, [1] JUMP target words
    uint32_t* aux;                  // per 4 words (FILL commands are at least four words apart): the Tile of the FILL whose seg_ix lies there
    uint32_t slot;
    uint32_t* pool;                 // LDS, two words per wave: [next, end) of the wave's share of the arena.  (Not registers: the walk
                                    // loop runs once per window of elements, a lane that left one early would come back with a stale copy --
                                    // and, as the first active lane, hand out chunks a second time.)
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

#ifndef COARSE_POOL_CHUNKS
#define COARSE_POOL_CHUNKS 16u  // chunks a wave takes from the arena at a time (one-walk route)
#endif
// MODE 0: count only.  1: write at the canonical addresses (bases known).  2: write into the scratch PTCL, chunks from the
// arena; what depends on the canonical allocation (chunk addresses, JUMP targets, seg_ix) is left tile-relative and marked.
template <int MODE>
JD void alloc_cmd(Cmd& c, uint32_t size) {  // coarse.wgsl:70-88
    const bool need = c.cmd_offset + size >= c.cmd_limit;
    uint32_t new_cmd = c.dyn_start + c.chunk_base + c.chunk_words;
    if (MODE == 1) {
        if (need) {  // (rare: once per 254 words)
            if (new_cmd + JL_PTCL_INCREMENT > c.cfg->ptcl_size) {
                new_cmd = 0u;
                atomicOr(&c.bump->failed, (uint32_t)JL_STAGE_COARSE);
            }
            c.ptcl.wr(c.cmd_offset, JL_CMD_JUMP);
            c.ptcl.wr(c.cmd_offset + 1u, new_cmd);
        }
    }
    if (MODE == 2) {
        // Chunks come out of the WAVE's share of the arena, COARSE_POOL_CHUNKS at a time: a returning atomic in this loop waits for
        // every PTCL store the wave has in flight (loads, stores and atomics share one in-order counter) -- with one atomic per
        // chunk the walk took 803 us instead of 478 on C4.  The lanes that need a chunk in this trip take consecutive ones.
        const uint64_t nm = __builtin_amdgcn_ballot_w64(need);
        if (nm != 0ull) {  // uniform
            const uint32_t cnt = (uint32_t)__builtin_popcountll(nm) * JL_PTCL_INCREMENT;
            uint32_t pn = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.pool[0]), pe = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.pool[1]);
            if (pe - pn < cnt) {  // (what is left of the old share is given up: the arena has room for that, see jh_launch_coarse)
                const uint32_t grab = umax_(cnt, COARSE_POOL_CHUNKS * JL_PTCL_INCREMENT);
                uint32_t base = 0u;
                if (lane_id() == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {  // the first lane still walking
                    base = atomicAdd(c.arena_ctr, grab);
                    // the chunks of the share that are not handed out right now have no owner yet -- and may never get one (the
                    // tail of the wave's last share, a share given up early): the relocation must not take what an earlier frame
                    // left in their records for one of its chunks
                    for (uint32_t i = cnt; i < grab; i += JL_PTCL_INCREMENT)
                        if (c.dyn_start + base + i + JL_PTCL_INCREMENT <= c.ptcl.n) c.owner[(base + i) / JL_PTCL_INCREMENT] = make_uint2(0xffffffffu, 0u);
                }
                pn = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                pe = pn + grab;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u));
            const uint32_t off = pn + rank * JL_PTCL_INCREMENT;
            wave_sync();  // (every lane has read the pool)
            if (lane_id() == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) { c.pool[0] = pn + cnt; c.pool[1] = pe; }
            wave_sync();
            if (need) {
                new_cmd = c.dyn_start + off;
                // (an arena that overflows: k_coarse_bases raises the flag -- the canonical PTCL may or may not have overflowed too)
                if (new_cmd + JL_PTCL_INCREMENT > c.ptcl.n || new_cmd + JL_PTCL_INCREMENT < new_cmd) {
                    new_cmd = 0u;
                } else {
                    c.owner[off / JL_PTCL_INCREMENT] = make_uint2(c.slot, c.chunk_words / JL_PTCL_INCREMENT);
                    ulonglong2* m = (ulonglong2*)(c.masks + (size_t)(new_cmd >> 6) * 2u);
#pragma unroll
                    for (int i = 0; i < 4; i++) m[i] = make_ulonglong2(0ull, 0ull);  // the chunk's four 64-word blocks: nothing marked yet
                }
                c.ptcl.wr(c.cmd_offset, JL_CMD_JUMP);
                c.ptcl.wr(c.cmd_offset + 1u, new_cmd);  // (provisional: the relocation writes the canonical address over it)
                if (c.cmd_offset + 1u < c.ptcl.n) atomicOr(c.masks + (size_t)((c.cmd_offset + 1u) >> 6) * 2u + 1u, 1ull << ((c.cmd_offset + 1u) & 63u));
            }
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
#ifndef COARSE_RELOCATE
#define COARSE_RELOCATE 1  // scenes with clip layers walk once and relocate (0: the two-pass route for every scene)
#endif
#ifndef COARSE_MAX_SPLIT
#define COARSE_MAX_SPLIT 16u
#endif
// The one-walk route of scenes with clip layers walks a tile's elements with the LANES of a wave (round 6, see the walk below);
// 0: one lane per tile, as the scenes without clips and rounds 1-5.
#ifndef COARSE_PAR_WALK
#define COARSE_PAR_WALK 1
#endif
// ... and splits the bins for this many workgroups per CU: every wave of a workgroup walks, so more workgroups are more walkers
// (C4 k_coarse<2>: 554 / 328 / 245 us with 1 / 2 / 4; nested C4 1061 / 616 / 423; the LDS of one allows four)
#ifndef COARSE_PAR_WG_PER_CU
#define COARSE_PAR_WG_PER_CU 4u
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
struct CoarseReloc {  // MODE 2 only (see Cmd)
    uint32_t* arena_ctr;
    uint2* owner;
    unsigned long long* masks;
    uint32_t* aux;
    uint32_t* end_pos;  // per tile slot: index of the stream's END word in the scratch PTCL
};

template <int MODE, bool CLIPS>
__global__ __launch_bounds__(JL_WG) void k_coarse(const JlConfig* __restrict__ cfg, Buf<uint32_t> scene, Buf<JlDrawMonoid> draw_monoids,
                                                  Buf<JlBinHeader> bin_headers, Buf<uint32_t> info_bin_data, Buf<JlPath> paths, Buf<JlTile> tiles,
                                                  JlBump* __restrict__ bump, Buf<uint32_t> ptcl, uint32_t* __restrict__ cnt_seg,
                                                  uint32_t* __restrict__ cnt_chunk, uint32_t* __restrict__ cnt_blend,
                                                  uint32_t* __restrict__ wg_tot, uint32_t n_wg, uint32_t bin_row0, uint32_t split, CoarseReloc R) {
    constexpr bool WRITE = MODE != 0;
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
    constexpr bool PAR = MODE == 2 && CLIPS && COARSE_PAR_WALK != 0;  // lanes = the elements of ONE tile (see the walk)
    __shared__ uint2 sh_tile_cache[PAR ? 1u : COARSE_TILE_CACHE];
    // PAR: the walk state of the workgroup's tiles (write position and limit, chunk words, segments, clip state: Walk / Cmd below) and,
    // per wave, two lists of the elements that include the tile it works on / the one it prepares
    __shared__ uint32_t sh_st[8][PAR ? JL_N_TILE : 1u];
    __shared__ uint8_t sh_list[JL_WG / 64][2][PAR ? JL_N_TILE : 1u];
    __shared__ uint32_t sh_pool[JL_WG / 64][2];  // MODE 2: the waves' shares of the chunk arena (alloc_cmd)
    if (MODE == 2 && threadIdx.x < JL_WG / 64) { sh_pool[threadIdx.x][0] = 0u; sh_pool[threadIdx.x][1] = 0u; }  // (barriers follow before any walk)

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
            }
            if (MODE != 1) {
#pragma unroll
                for (uint32_t k = 0; k < COARSE_TPL; k++)
                    if (W[k].has_tile) { cnt_seg[W[k].slot] = 0u; cnt_chunk[W[k].slot] = 0u; cnt_blend[W[k].slot] = 0u; }
                if (lid < 3u) wg_tot[lid * n_wg + bin_ix * split + blockIdx.z] = 0u;
            }
            if (MODE == 2) {
#pragma unroll
                for (uint32_t k = 0; k < COARSE_TPL; k++)
                    if (W[k].has_tile) R.end_pos[W[k].slot] = 0xffffffffu;  // nothing to relocate
            }
            return;
        }
    }
    const uint32_t n_partitions = (cfg->layout.n_drawobj + JL_N_TILE - 1u) / JL_N_TILE;
    const uint32_t bin_tile_x = JL_N_TILE_X * blockIdx.x;
    const uint32_t bin_tile_y = JL_N_TILE_Y * bin_y;
    const uint32_t BLEND_CLIP = (128u << 8) | 0u;  // MIX_CLIP << 8 | COMPOSE_SRC_OVER (Jello numbering, blend.wgsl:199-202)
    uint32_t my_base_seg = 0u, my_base_chunk = 0u, my_base_blend = 0u;
    if (MODE == 1) {
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
        w.c.arena_ctr = R.arena_ctr; w.c.owner = R.owner; w.c.masks = R.masks; w.c.aux = R.aux; w.c.slot = w.slot;
        w.c.pool = &sh_pool[lid >> 6][0];
        // (only a tile of the target owns its head: the index of one beyond the right edge is another tile's)
        if (MODE == 2 && w.has_tile && bin_tile_x + w.tile_x < cfg->width_in_tiles && bin_tile_y + w.tile_y < cfg->height_in_tiles &&
            (size_t)this_tile_ix * JL_PTCL_INITIAL_ALLOC + JL_PTCL_INITIAL_ALLOC <= ptcl.n) {  // the head's 64-word block: nothing marked yet
            R.masks[(size_t)this_tile_ix * 2u] = 0ull;
            R.masks[(size_t)this_tile_ix * 2u + 1u] = 0ull;
        }
        w.clip_zero_depth = 0u; w.clip_depth = 0u; w.render_blend_depth = 0u; w.max_blend_depth = 0u;
        w.blend_offset = w.c.cmd_offset;
        w.c.cmd_offset += 1u;
        w.slice_ix = 7u; w.bitmap = 0u; w.nz = 0u; w.el_next = 0xffffffffu;
        w.q0 = make_uint4(0u, 0u, 0u, 0u); w.q2 = w.q0;
        w.tile.backdrop = 0; w.tile.segment_count_or_ix = 0u;
    }
    if constexpr (PAR) {  // (tile t of the workgroup's part = thread t; barriers follow before the first walk)
        if (W[0].has_tile) {
            sh_st[0][lid] = W[0].c.cmd_offset; sh_st[1][lid] = W[0].c.cmd_limit; sh_st[2][lid] = 0u; sh_st[3][lid] = 0u;
            sh_st[4][lid] = 0u; sh_st[5][lid] = 0u; sh_st[6][lid] = 0u; sh_st[7][lid] = 0u;
        }
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
        uint32_t win_e1 = PAR ? JL_N_TILE : win_e0;  // largest e1 with sh_tile_count[e1 - 1] - win_p0 <= COARSE_TILE_CACHE (uniform); PAR: no cache, the whole batch
        for (uint32_t step = 256u; step > 0u && !PAR; step >>= 1)
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
                if (!PAR) sh_tile_cache[ix - win_p0] = make_uint2((uint32_t)tile.backdrop, tile.segment_count_or_ix);
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
        if constexpr (PAR) {
            // ---- The walk, lanes = the ELEMENTS of one tile (round 6) ------------------------------------------------------------------
            // With one lane per tile a wave issues ~130 instructions per (tile, element) trip for 64 tiles at best, one walking wave
            // per workgroup, and a C4 tile sees 640 elements: 0.51 ms of one instruction after the other on a quarter of the SIMDs.
            // Here every wave of the workgroup takes tiles of its own (tile = wave, wave + 4, ...), and the elements of the batch that
            // include the tile -- the set bits of its bitmaps, compacted -- are its lanes.  What the sequential walk carries from
            // element to element becomes wave arithmetic:
            //   * write positions: prefix sums of the command sizes; a chunk boundary is the first command that does not fit
            //     (ballot + count-trailing-zeros), the commands behind it are re-based, and so on (a boundary per ~50 commands);
            //   * segment indices: a prefix sum;
            //   * the clip state machine (coarse.wgsl:398-441).  A BEGIN_CLIP whose tile is empty starts a stretch that emits nothing
            //     up to its END_CLIP.  The stretches nest properly, so the elements that emit nothing are the UNION of the
            //     (begin, end] intervals of ALL empty BEGIN_CLIPs, reached or not: an element is skipped iff more empty BEGIN_CLIPs
            //     than their END_CLIPs lie in front of it (two mbcnt), plus the stretch the tile entered the chunk in.  An END_CLIP
            //     finds its BEGIN_CLIP as the clip kernels do: per nesting level one ballot of the level's BEGINs, masked to the lanes
            //     below, count-leading-zeros.  Blend depth and its maximum are a prefix sum and a wave maximum.
            // The Tile of a (tile, element) pair is read from memory again (the include test has just had it: L2), requested for the
            // wave's NEXT tile before the current one is worked on; no Tile cache, no windows.  Commands are stored with their exact
            // sizes: the lanes of one store instruction must not overlap (the one-lane walk pads to 16 bytes and lets the next command
            // overwrite the padding).
            const uint32_t wv = lid >> 6, lane = lid & 63u;
            const uint64_t below = (1ull << lane) - 1ull, above = ~below & ~(1ull << lane);
            auto mbcnt64 = [&](uint64_t m) -> uint32_t { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
            // the elements of tile t in buffer `buf` of the wave's lists; returns their number and requests the first 64 Tiles
            auto prepare = [&](uint32_t t, uint32_t buf, uint32_t& e_out, JlTile& tl) -> uint32_t {
                e_out = 0u; tl.backdrop = 0; tl.segment_count_or_ix = 0u;
                if (t >= part_tiles) return 0u;  // uniform
                const uint32_t ty = part_y0 + t / JL_N_TILE_X, tx = t % JL_N_TILE_X, xy = ty * JL_N_TILE_X + tx;
                uint32_t n = 0u;
#pragma unroll
                for (uint32_t c = 0u; c < 4u; c++) {
                    const uint64_t m = (uint64_t)uni(sh_bitmaps[2u * c][xy]) | ((uint64_t)uni(sh_bitmaps[2u * c + 1u][xy]) << 32);
                    if (m != 0ull) {  // uniform
                        if ((m >> lane) & 1ull) sh_list[wv][buf][(n + mbcnt64(m)) & 255u] = (uint8_t)(64u * c + lane);
                        n += (uint32_t)__builtin_popcountll(m);
                    }
                }
                wave_sync();
                if (lane < n) {
                    e_out = sh_list[wv][buf][lane];
                    const uint4 q0 = sh_r0[e_out];
                    tl = tiles.rd(q0.z + q0.w * ty + tx);
                }
                return n;
            };
            uint32_t nx_e; JlTile nx_tile;
            uint32_t nx_n = prepare(wv, 0u, nx_e, nx_tile);
            uint32_t buf = 0u;
            for (uint32_t t = wv; t < part_tiles; t += JL_WG / 64u, buf ^= 1u) {  // uniform per wave
                const uint32_t n = nx_n, e_first = nx_e;
                const JlTile tile_first = nx_tile;
                nx_n = prepare(t + JL_WG / 64u, buf ^ 1u, nx_e, nx_tile);
                if (n == 0u) continue;  // uniform: nothing of this batch includes the tile
                const uint32_t ty = part_y0 + t / JL_N_TILE_X, tx = t % JL_N_TILE_X;
                const uint32_t slot = bin_ix * JL_N_TILE + ty * JL_N_TILE_X + tx;
                uint32_t off = uni(sh_st[0][t]), lim = uni(sh_st[1][t]), chunkw = uni(sh_st[2][t]), segused = uni(sh_st[3][t]);
                uint32_t czd = uni(sh_st[4][t]), depth = uni(sh_st[5][t]), rbd = uni(sh_st[6][t]), maxbd = uni(sh_st[7][t]);
                const uint32_t dyn_start = cfg->width_in_tiles * cfg->height_in_tiles * JL_PTCL_INITIAL_ALLOC;
                for (uint32_t k0 = 0u; k0 < n; k0 += 64u) {  // uniform
                    const bool live = k0 + lane < n;
                    uint32_t e = e_first;
                    JlTile tile = tile_first;
                    if (k0 != 0u) {  // (more than 64 elements of one batch on one tile: rare)
                        e = live ? sh_list[wv][buf][k0 + lane] : 0u;
                        tile.backdrop = 0; tile.segment_count_or_ix = 0u;
                        if (live) { const uint4 t0 = sh_r0[e]; tile = tiles.rd(t0.z + t0.w * ty + tx); }
                    }
                    const uint4 q0 = sh_r0[e], q2 = sh_r2[e];
                    const uint32_t meta = live ? q0.x : 0u;
                    const uint32_t n_segs = live ? tile.segment_count_or_ix : 0u;
                    const uint32_t is_begin = (meta / CM_BEGIN) & 1u, is_end = (meta / CM_END) & 1u, has_path = (meta / CM_PATH) & 1u;
                    const uint32_t has_segs = umin_(n_segs, 1u);
                    const uint32_t zero_tile = live ? 1u - umin_(n_segs | (uint32_t)tile.backdrop, 1u) : 0u;
                    // nesting depth in front of the element
                    const uint32_t dlt = is_begin - is_end;
                    const uint32_t d_incl = wave_incl_scan_u32(dlt);
                    const uint32_t D = depth + d_incl - dlt;
                    const uint64_t begins = __builtin_amdgcn_ballot_w64(is_begin != 0u);
                    const uint64_t zb = __builtin_amdgcn_ballot_w64((is_begin & zero_tile) != 0u);
                    // partners inside the chunk, level by level
                    uint32_t has_partner = 0u, partner_empty = 0u, matched = 0u;
                    for (uint64_t todo = begins; todo != 0ull;) {  // uniform: one trip per nesting level with a BEGIN_CLIP in the chunk
                        const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)D, (int)__builtin_ctzll(todo));
                        const bool bl_me = is_begin != 0u && D == L, el_me = is_end != 0u && D == L + 1u;
                        const uint64_t bl = __builtin_amdgcn_ballot_w64(bl_me), el = __builtin_amdgcn_ballot_w64(el_me);
                        if (el_me) {
                            const uint64_t c = bl & below;
                            if (c != 0ull) { has_partner = 1u; partner_empty = (uint32_t)(zb >> (63u - (uint32_t)__builtin_clzll(c))) & 1u; }
                        }
                        if (bl_me) {
                            const uint64_t ea = el & above, ba = bl & above;
                            matched = (ea != 0ull && (ba == 0ull || __builtin_ctzll(ea) < __builtin_ctzll(ba))) ? 1u : 0u;
                        }
                        todo &= ~bl;
                    }
                    const uint64_t ze = __builtin_amdgcn_ballot_w64((is_end & partner_empty) != 0u);
                    uint32_t skipped = mbcnt64(zb) > mbcnt64(ze) ? 1u : 0u;  // (the intervals nest: the counts below a lane never cross)
                    uint32_t czd_out;
                    bool entered_persists = false;
                    if (czd != 0u) {  // uniform: the tile entered the chunk inside a skipped stretch; it ends at the END_CLIP of that depth
                        const uint64_t fin = __builtin_amdgcn_ballot_w64(is_end != 0u && has_partner == 0u && D == czd);
                        const uint32_t endl = fin != 0ull ? (uint32_t)__builtin_ctzll(fin) : 64u;
                        if (lane <= endl) skipped = 1u;
                        entered_persists = fin == 0ull;
                    }
                    if (entered_persists) {
                        czd_out = czd;
                    } else {  // the outermost empty BEGIN_CLIP still open behind the chunk
                        const uint64_t open_z = __builtin_amdgcn_ballot_w64((is_begin & zero_tile) != 0u && matched == 0u);
                        czd_out = open_z != 0ull ? (uint32_t)__builtin_amdgcn_readlane((int)D, (int)__builtin_ctzll(open_z)) + 1u : 0u;
                    }
                    const uint32_t active = 1u - skipped;
                    // blend depth
                    const uint32_t opened = is_begin & active & (1u - zero_tile), closes = is_end & active;
                    const uint32_t rd = opened - closes;
                    const uint32_t r_incl = wave_incl_scan_u32(rd);
                    const uint32_t peak = rbd + r_incl - rd + opened;
                    maxbd = umax_(maxbd, (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_max_u32(peak), 63));
                    rbd += (uint32_t)__builtin_amdgcn_readlane((int)r_incl, 63);
                    depth += (uint32_t)__builtin_amdgcn_readlane((int)d_incl, 63);
                    czd = czd_out;
                    // commands: sizes, segment index, positions
                    const uint32_t emit_path = active & has_path;
                    const uint32_t emit_brush = active & (1u - (is_begin & zero_tile));
                    const uint32_t s1 = emit_path * (1u + 3u * has_segs);
                    const uint32_t s2 = emit_brush * ((meta >> CM_NBRUSH_SHIFT) & 7u);
                    const uint32_t segs_here = emit_path * n_segs;
                    const uint32_t sg_incl = wave_incl_scan_u32(segs_here);
                    const uint32_t seg_ix = segused + sg_incl - segs_here;  // (tile-relative: the relocation adds the tile's base)
                    segused += (uint32_t)__builtin_amdgcn_readlane((int)sg_incl, 63);
                    const uint32_t sz_incl = wave_incl_scan_u32(s1 + s2);
                    uint32_t o1 = off + sz_incl - (s1 + s2), o2 = o1 + s1;  // as if everything fitted the current chunk
                    uint32_t fl = 0u, fslot = 0u;  // commands in front of (fl, fslot) are placed
                    for (uint32_t guard = 0u; guard < 130u; guard++) {  // uniform; one trip per chunk boundary
                        const bool at1 = lane > fl || (lane == fl && fslot == 0u), at2 = lane >= fl;
                        const uint64_t m1 = __builtin_amdgcn_ballot_w64(s1 != 0u && at1 && o1 + s1 >= lim);
                        const uint64_t m2 = __builtin_amdgcn_ballot_w64(s2 != 0u && at2 && o2 + s2 >= lim);
                        if ((m1 | m2) == 0ull) break;
                        const uint32_t l1 = m1 != 0ull ? (uint32_t)__builtin_ctzll(m1) : 64u, l2 = m2 != 0ull ? (uint32_t)__builtin_ctzll(m2) : 64u;
                        const uint32_t bl = l1 <= l2 ? l1 : l2, bslot = l1 <= l2 ? 0u : 1u;
                        const uint32_t bo = (uint32_t)__builtin_amdgcn_readlane((int)(bslot == 0u ? o1 : o2), (int)bl);
                        // a chunk from the wave's share of the arena (alloc_cmd<2>, one chunk at a time)
                        uint32_t pn = uni(sh_pool[wv][0]), pe = uni(sh_pool[wv][1]);
                        if (pe - pn < JL_PTCL_INCREMENT) {
                            const uint32_t grab = COARSE_POOL_CHUNKS * JL_PTCL_INCREMENT;
                            uint32_t base = 0u;
                            if (lane == 0u) base = atomicAdd(R.arena_ctr, grab);
                            base = uni(base);
                            // (the chunks of the share that are not handed out now have no owner yet and may never get one)
                            if (lane >= 1u && lane < COARSE_POOL_CHUNKS && dyn_start + base + lane * JL_PTCL_INCREMENT + JL_PTCL_INCREMENT <= ptcl.n)
                                R.owner[base / JL_PTCL_INCREMENT + lane] = make_uint2(0xffffffffu, 0u);
                            pn = base; pe = base + grab;
                        }
                        wave_sync();
                        if (lane == 0u) { sh_pool[wv][0] = pn + JL_PTCL_INCREMENT; sh_pool[wv][1] = pe; }
                        wave_sync();
                        uint32_t new_cmd = dyn_start + pn;
                        if (new_cmd + JL_PTCL_INCREMENT > ptcl.n || new_cmd + JL_PTCL_INCREMENT < new_cmd) {
                            new_cmd = 0u;  // (an arena that overflows: k_coarse_bases raises the flag)
                        } else {
                            if (lane == 0u) R.owner[pn / JL_PTCL_INCREMENT] = make_uint2(slot, chunkw / JL_PTCL_INCREMENT);
                            if (lane < 4u) ((ulonglong2*)(R.masks + (size_t)(new_cmd >> 6) * 2u))[lane] = make_ulonglong2(0ull, 0ull);  // nothing marked yet
                        }
                        if (lane == 0u) {
                            ptcl.wr(bo, JL_CMD_JUMP);
                            ptcl.wr(bo + 1u, new_cmd);  // (provisional: the relocation writes the canonical address over it)
                            if (bo + 1u < ptcl.n) atomicOr(R.masks + (size_t)((bo + 1u) >> 6) * 2u + 1u, 1ull << ((bo + 1u) & 63u));
                        }
                        if (lane > bl || (lane == bl && bslot == 0u)) o1 = o1 - bo + new_cmd;
                        if (lane >= bl) o2 = o2 - bo + new_cmd;
                        lim = new_cmd + (JL_PTCL_INCREMENT - JL_PTCL_HEADROOM);
                        chunkw += JL_PTCL_INCREMENT;
                        fl = bl; fslot = bslot;
                    }
                    off = (uint32_t)__builtin_amdgcn_readlane((int)(o2 + s2), 63);
                    // stores, exact sizes
                    if (s1 == 4u) {
                        const uint32_t at = o1 + 2u;
                        if (at < ptcl.n) {  // seg_ix is tile-relative: marked, the relocation adds the tile's base and writes the Tile
                            atomicOr(R.masks + (size_t)(at >> 6) * 2u, 1ull << (at & 63u));
                            R.aux[at >> 2] = q0.z + q0.w * ty + tx;
                        }
                        ptcl_wr4(ptcl, o1, JL_CMD_FILL, (n_segs << 1) | ((meta / CM_EVENODD_FILL) & 1u), seg_ix, (uint32_t)tile.backdrop);
                    } else if (s1 == 1u) {
                        ptcl.wr(o1, JL_CMD_SOLID);
                    }
                    if (s2 >= 4u) {
                        ptcl_wr4(ptcl, o2, q0.y, q2.x, q2.y, q2.z);
                        if (s2 == 5u) ptcl.wr(o2 + 4u, q2.w);
                    } else if (s2 != 0u) {
                        ptcl.wr(o2, q0.y);
                        if (s2 >= 2u) ptcl.wr(o2 + 1u, q2.x);
                        if (s2 >= 3u) ptcl.wr(o2 + 2u, q2.y);
                    }
                }
                if (lane == 0u) {
                    sh_st[0][t] = off; sh_st[1][t] = lim; sh_st[2][t] = chunkw; sh_st[3][t] = segused;
                    sh_st[4][t] = czd; sh_st[5][t] = depth; sh_st[6][t] = rbd; sh_st[7][t] = maxbd;
                }
            }
        } else {
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
                alloc_cmd<MODE>(c, s1);  // (no-op for s1 == 0: cmd_offset < cmd_limit between commands)
                // Stores: a command of fewer than four words is written as four -- the words behind it belong to this
                // tile's chunk (cmd_offset + 1 < cmd_limit and the two words of headroom) and are either overwritten
                // by the next command or never reached -- so that a trip has three predicated stores instead of eight.
                if (WRITE) {
                    const bool room = c.cmd_offset + 8u <= c.ptcl.n && c.cmd_offset + 8u > c.cmd_offset;
                    const uint32_t rule = (n_segs << 1) | ((meta / CM_EVENODD_FILL) & 1u);
                    if (s1 == 4u) {
                        const uint32_t tile_ix = q0.z + q0.w * w.tile_y + w.tile_x;
                        if (MODE == 1) {
                            if (tiles.ok(tile_ix)) tiles.p[tile_ix].segment_count_or_ix = ~seg_ix;
                        } else if (c.cmd_offset + 2u < c.ptcl.n) {  // seg_ix is tile-relative here: marked, the relocation adds the tile's base and writes the Tile
                            const uint32_t at = c.cmd_offset + 2u;
                            atomicOr(c.masks + (size_t)(at >> 6) * 2u, 1ull << (at & 63u));
                            c.aux[at >> 2] = tile_ix;
                        }
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
                alloc_cmd<MODE>(c, s2);
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
        }  // (!PAR)
        if (PAR || win_p1 >= total_tile_count) break;
        __syncthreads();  // the next window's include test overwrites the Tile cache
        win_e0 = win_e1; win_p0 = win_p1;
        }
        if (!has_next) break;
        __syncthreads();
    }
    if constexpr (PAR) {  // (thread = tile again)
        __syncthreads();
        if (W[0].has_tile) {
            W[0].c.cmd_offset = sh_st[0][lid]; W[0].c.chunk_words = sh_st[2][lid]; W[0].c.seg_used = sh_st[3][lid];
            W[0].max_blend_depth = sh_st[7][lid];
        }
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
        if (MODE == 1) {
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
            if (MODE == 2) {  // (blend_ix is the relocation's business: it needs the prefix over the tiles)
                if (in_target) w.c.ptcl.wr(w.c.cmd_offset, JL_CMD_END);
                R.end_pos[w.slot] = in_target ? w.c.cmd_offset : 0xffffffffu;
            }
            cnt_seg[w.slot] = w.c.seg_used;
            cnt_chunk[w.slot] = w.c.chunk_words;
            cnt_blend[w.slot] = scratch_size;
            my_tot.v[0] = w.c.seg_used; my_tot.v[1] = w.c.chunk_words; my_tot.v[2] = scratch_size;
        }
    }
    if (MODE != 1) {
        const MonoidK<3> t = block_reduce_monoid<3>(my_tot, sh_red);
        if (lid < 3u) wg_tot[lid * n_wg + bin_ix * split + blockIdx.z] = lid == 0u ? t.v[0] : (lid == 1u ? t.v[1] : t.v[2]);
    }
}

// ---- one walk + relocation (round 5; scenes with clip layers) ------------------------------------------------------------
// k_coarse<2> has written every tile's stream into a scratch copy of the PTCL -- heads in place, chunks wherever the arena had
// room -- and left per tile what it needs (segments, chunk words, blend space).  k_coarse_bases turns the needs into the
// canonical bases (exclusive prefixes in (bin, tile) order: sum of the workgroups in front + prefix inside the workgroup, as
// the write pass of the two-pass route does for itself), reports the totals and the PTCL overflow; k_coarse_relocate then
// moves every live 64-word block to its canonical address and completes the marked words on the way: seg_ix + the tile's
// segment base (and ~seg_ix into the FILL's Tile), JUMP targets, blend_ix.
__global__ __launch_bounds__(JL_WG) void k_coarse_bases(const JlConfig* __restrict__ cfg, JlBump* __restrict__ bump, const uint32_t* __restrict__ cnt_seg,
                                                        const uint32_t* __restrict__ cnt_chunk, const uint32_t* __restrict__ cnt_blend,
                                                        const uint32_t* __restrict__ wg_tot, uint32_t n_wg, uint32_t split, uint32_t* __restrict__ base_seg,
                                                        uint32_t* __restrict__ base_chunk, uint32_t* __restrict__ base_blend,
                                                        uint32_t* __restrict__ arena_ctr, uint32_t* __restrict__ arena_used, uint32_t arena_cap) {
    __shared__ uint32_t sh_scan[8];
    __shared__ uint32_t sh_red[12];
    const uint32_t lid = threadIdx.x, my_wg = blockIdx.x;
    const uint32_t bin_ix = my_wg / split, strip = my_wg % split;
    const uint32_t part_rows = JL_N_TILE_Y / split, part_tiles = part_rows * JL_N_TILE_X;
    MonoidK<3> before, all;
#pragma unroll
    for (int c = 0; c < 3; c++) { before.v[c] = 0u; all.v[c] = 0u; }
    const bool totals = my_wg == 0u;
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
        if (lid == 0u) {
            bump->segments = t.v[0]; bump->ptcl = t.v[1]; bump->blend = t.v[2];
            // coarse.wgsl:76-79: the chunk that does not fit raises the flag -- the last one handed out ends at dyn_start + total
            const uint64_t dyn_start = (uint64_t)cfg->width_in_tiles * cfg->height_in_tiles * JL_PTCL_INITIAL_ALLOC;
            if (t.v[1] != 0u && dyn_start + t.v[1] > cfg->ptcl_size) atomicOr(&bump->failed, (uint32_t)JL_STAGE_COARSE);
            *arena_used = *arena_ctr;  // (the walk is over: what the relocation has to look at)
            // the arena gives up what is left of a wave's share when the wave takes a new one: it can run out although the
            // canonical PTCL would have fitted -- reported like any other overflow (the regrow loop then grows both)
            if (*arena_ctr > arena_cap) {
                atomicOr(&bump->failed, (uint32_t)JL_STAGE_COARSE);
                bump->ptcl = umax_(bump->ptcl, *arena_ctr);  // (what the regrow loop sizes the next attempt from: the arena's real use)
            }
        }
        __syncthreads();
    }
    const bool mine = lid < part_tiles;
    const uint32_t slot = bin_ix * JL_N_TILE + strip * part_tiles + lid;
    uint32_t tot;
    const uint32_t a = carry.v[0] + block_excl_scan_u32(mine ? cnt_seg[slot] : 0u, sh_scan, &tot);
    __syncthreads();
    const uint32_t b = carry.v[1] + block_excl_scan_u32(mine ? cnt_chunk[slot] : 0u, sh_scan, &tot);
    __syncthreads();
    const uint32_t d = carry.v[2] + block_excl_scan_u32(mine ? cnt_blend[slot] : 0u, sh_scan, &tot);
    if (mine) { base_seg[slot] = a; base_chunk[slot] = b; base_blend[slot] = d; }
}

// A wave per 256 words of the scratch PTCL, a lane per four words (16-byte loads and stores): the heads of the band's tiles, four to
// a wave (units [0, n_head_units)), then the arena's chunks.  (One 64-word block per wave took 102 us on C4: four dependent
// look-ups in front of 256 bytes of traffic.)
__global__ __launch_bounds__(JL_WG) void k_coarse_relocate(const JlConfig* __restrict__ cfg, JlBump* __restrict__ bump, Buf<uint32_t> tmp, Buf<uint32_t> ptcl,
                                                           Buf<JlTile> tiles, const uint32_t* __restrict__ cnt_chunk, const uint32_t* __restrict__ cnt_blend,
                                                           const uint32_t* __restrict__ base_seg, const uint32_t* __restrict__ base_chunk,
                                                           const uint32_t* __restrict__ base_blend, const uint32_t* __restrict__ end_pos,
                                                           const uint2* __restrict__ owner, const unsigned long long* __restrict__ masks,
                                                           const uint32_t* __restrict__ aux, const uint32_t* __restrict__ arena_used,
                                                           uint32_t bin_row0, uint32_t bin_row1, uint32_t n_slots) {
    const uint32_t lane = lane_id();
    const uint32_t width_in_bins = (cfg->width_in_tiles + JL_N_TILE_X - 1u) / JL_N_TILE_X;
    const uint32_t dyn_start = cfg->width_in_tiles * cfg->height_in_tiles * JL_PTCL_INITIAL_ALLOC;
    const uint32_t slot0 = bin_row0 * width_in_bins * JL_N_TILE, slot1 = umin_(bin_row1 * width_in_bins * JL_N_TILE, n_slots);
    const uint32_t n_head_units = slot1 > slot0 ? (slot1 - slot0 + 3u) / 4u : 0u;
    const uint32_t n_chunks = umin_(*arena_used, tmp.n > dyn_start ? tmp.n - dyn_start : 0u) / JL_PTCL_INCREMENT;
    const uint32_t units = n_head_units + n_chunks;
    const uint32_t n_waves = (gridDim.x * JL_WG) >> 6;
    const uint32_t part = lane >> 4, wl = (lane & 15u) * 4u;  // which 64-word block of the unit, which four words of it
    for (uint32_t u = (blockIdx.x * JL_WG + threadIdx.x) >> 6; u < units; u += n_waves) {  // uniform per wave
        uint32_t slot, src, dst, next_chunk;  // next_chunk: ordinal of the chunk a JUMP in this block leads to
        bool head = u < n_head_units, live = true;
        if (head) {
            slot = slot0 + u * 4u + part;
            live = slot < slot1;
            const uint32_t bin_ix = slot / JL_N_TILE, xy = slot % JL_N_TILE;
            const uint32_t tx = (bin_ix % width_in_bins) * JL_N_TILE_X + xy % JL_N_TILE_X, ty = (bin_ix / width_in_bins) * JL_N_TILE_Y + xy / JL_N_TILE_X;
            live = live && tx < cfg->width_in_tiles && ty < cfg->height_in_tiles;
            if (live) live = end_pos[slot] != 0xffffffffu;
            src = dst = (ty * cfg->width_in_tiles + tx) * JL_PTCL_INITIAL_ALLOC + wl;
            next_chunk = 0u;
        } else {
            const uint32_t c = u - n_head_units;
            const uint2 ow = owner[c];
            slot = ow.x;
            src = dyn_start + c * JL_PTCL_INCREMENT + part * 64u + wl;
            dst = 0u; next_chunk = 0u;
            live = slot >= slot0 && slot < slot1;  // (else: another band's tile)
            if (live) {
                const uint32_t k = ow.y, n_k = cnt_chunk[slot] / JL_PTCL_INCREMENT;
                // the stream's last chunk is live up to its END word; a chunk in front of it up to its JUMP (copied whole)
                live = k < n_k && !(k + 1u == n_k && (src & ~63u) > end_pos[slot]);
                dst = dyn_start + base_chunk[slot] + k * JL_PTCL_INCREMENT + part * 64u + wl;
                next_chunk = k + 1u;
            }
        }
        // Words behind a unit's JUMP target or END are nobody's: they are not written (the two-pass route leaves them alone as well,
        // and what the arena holds there depends on the order in which the waves took their chunks: tools/determinism.py hashes
        // the whole buffer).  The JUMP mask of a chunk's earlier blocks comes from the first lane of their quarter of the wave.
        // (all of a lane's loads are requested before any of them is looked at: one round trip, not two)
        const bool in_range = src + 3u < tmp.n, fetch = live && in_range;
        const unsigned long long raw_jm = fetch ? masks[(size_t)(src >> 6) * 2u + 1u] : 0ull;
        const unsigned long long raw_fm = fetch ? masks[(size_t)(src >> 6) * 2u] : 0ull;
        const uint4 in = fetch ? *(const uint4*)(tmp.p + src) : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t end_at = fetch ? end_pos[slot] : 0u;
        bool dead_block = false;
        if (!head) {  // uniform
            const uint32_t lo = (uint32_t)raw_jm, hi = (uint32_t)(raw_jm >> 32);
            const bool j0 = ((uint32_t)__builtin_amdgcn_readlane((int)lo, 0) | (uint32_t)__builtin_amdgcn_readlane((int)hi, 0)) != 0u;
            const bool j1 = ((uint32_t)__builtin_amdgcn_readlane((int)lo, 16) | (uint32_t)__builtin_amdgcn_readlane((int)hi, 16)) != 0u;
            const bool j2 = ((uint32_t)__builtin_amdgcn_readlane((int)lo, 32) | (uint32_t)__builtin_amdgcn_readlane((int)hi, 32)) != 0u;
            dead_block = (part > 0u && j0) || (part > 1u && j1) || (part > 2u && j2);
        }
        if (!fetch || dead_block) continue;
        const bool ends_here = head ? (end_at >> 6) == (src >> 6) : (end_at >= (src & ~63u) - part * 64u && end_at - ((src & ~63u) - part * 64u) < JL_PTCL_INCREMENT);
        bool keep[4];
#pragma unroll
        for (uint32_t e = 0; e < 4u; e++) {
            const uint32_t ix = (src & 63u) + e;
            keep[e] = (raw_jm & ((1ull << ix) - 1ull)) == 0ull && !(ends_here && src + e > end_at);
        }
        if (!(keep[0] || keep[1] || keep[2] || keep[3])) continue;
        const unsigned long long fm = raw_fm >> (src & 63u), jm = raw_jm >> (src & 63u);
        uint32_t w[4] = {in.x, in.y, in.z, in.w};
        if (((fm | jm) & 15ull) != 0ull || (head && wl == 0u)) {
            const uint32_t jump_to = dyn_start + base_chunk[slot] + next_chunk * JL_PTCL_INCREMENT;
#pragma unroll
            for (uint32_t e = 0; e < 4u; e++) {
                if ((fm >> e) & 1ull) {
                    w[e] += base_seg[slot];
                    const uint32_t tile_ix = aux[(src + e) >> 2];
                    if (tiles.ok(tile_ix)) tiles.p[tile_ix].segment_count_or_ix = ~w[e];
                }
                if ((jm >> e) & 1ull) w[e] = jump_to;
            }
            if (head && wl == 0u) {  // blend_ix, coarse.wgsl:452-460
                w[0] = 0u;
                const uint32_t need = cnt_blend[slot];
                if (need != 0u) {
                    w[0] = base_blend[slot];
                    if (w[0] + need > cfg->blend_size) atomicOr(&bump->failed, (uint32_t)JL_STAGE_COARSE);
                }
            }
        }
        if (dst + 3u < ptcl.n && keep[0] && keep[1] && keep[2] && keep[3]) {
            *(uint4*)(ptcl.p + dst) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
#pragma unroll
            for (uint32_t e = 0; e < 4u; e++)
                if (keep[e]) ptcl.wr(dst + e, w[e]);
        }
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
    const bool clips_early = !(L.cfg_host && L.cfg_host->layout.n_clip == 0u);
    const uint32_t want = (clips_early && COARSE_RELOCATE && COARSE_PAR_WALK ? COARSE_PAR_WG_PER_CU : COARSE_WG_PER_CU) * (uint32_t)(L.num_cus > 0 ? L.num_cus : 256);
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
    CoarseReloc R;
    std::memset(&R, 0, sizeof R);
#define JH_COARSE(W, C, G, ROW0, P) hipLaunchKernelGGL((k_coarse<W, C>), G, blk, 0, L.stream, cfg, scene, dm, bh, ibd, paths, tiles, bump, P, cnt_seg, cnt_chunk, cnt_blend, wg_tot, n_wg, ROW0, split, R)
#if COARSE_RELOCATE
    // Scenes with clip layers: ONE walk into a scratch copy of the PTCL, then the relocation (the walk of such a scene is long --
    // C4: ~640 trips per tile -- and the counting pass repeated all of it; for a scene without clips the relocation's traffic
    // eats what it saves).
    if ((clips || COARSE_RELOCATE == 2) && ptcl.n != 0u) {
        // the scratch PTCL: the real one's size + what the waves' arena shares can leave unused (a share given up early, the tail of
        // the last one): 2 x COARSE_POOL_CHUNKS chunks per walking wave
        const uint64_t slack = (uint64_t)n_wg * 4u * 2u * COARSE_POOL_CHUNKS * JL_PTCL_INCREMENT;
        const uint64_t words = (uint64_t)ptcl.n + slack > 0xfffffff0ull ? 0xfffffff0ull : (uint64_t)ptcl.n + slack;
        uint32_t* tmp = (uint32_t*)jh_scratch_get(L.scratch, JH_SCR_B, words * 4);
        // [masks: 2 x u64 per 64 words | aux: u32 per 4 words | owner: uint2 per 256 words | end_pos, base_seg, base_chunk, base_blend: n each | arena counter, arena used]
        const uint64_t n_blocks = (words + 63u) / 64u, n_aux = (words + 3u) / 4u, n_own = words / JL_PTCL_INCREMENT + 1u;
        const uint64_t bytes = n_blocks * 16u + n_aux * 4u + n_own * 8u + (uint64_t)n * 16u + 256u;
        uint8_t* side = (uint8_t*)jh_scratch_get(L.scratch, JH_SCR_C, bytes);
        if (!tmp || !side) return -5;
        R.masks = (unsigned long long*)side;
        R.owner = (uint2*)(side + n_blocks * 16u);
        R.aux = (uint32_t*)(side + n_blocks * 16u + n_own * 8u);
        uint32_t* per_slot = R.aux + n_aux;
        R.end_pos = per_slot;
        uint32_t *base_seg = per_slot + n, *base_chunk = per_slot + 2 * (size_t)n, *base_blend = per_slot + 3 * (size_t)n;
        R.arena_ctr = per_slot + 4 * (size_t)n;
        uint32_t* arena_used = R.arena_ctr + 1;
        (void)hipMemsetAsync(R.arena_ctr, 0, 8, L.stream);
        const uint64_t cfg_dyn = L.cfg_host ? (uint64_t)L.cfg_host->width_in_tiles * L.cfg_host->height_in_tiles * JL_PTCL_INITIAL_ALLOC : 0u;
        // (owner records of chunks no tile took this frame must not look like one of the band's: slot ~0)
        auto tmpbuf = mkbuf<uint32_t>(tmp, words * 4);
        if (clips) JH_COARSE(2, true, grid, 0u, tmpbuf); else JH_COARSE(2, false, grid, 0u, tmpbuf);
        hipLaunchKernelGGL(k_coarse_bases, dim3(n_wg), blk, 0, L.stream, cfg, bump, (const uint32_t*)cnt_seg, (const uint32_t*)cnt_chunk, (const uint32_t*)cnt_blend,
                           (const uint32_t*)wg_tot, n_wg, split, base_seg, base_chunk, base_blend, R.arena_ctr, arena_used,
                           (uint32_t)(words > (uint64_t)cfg_dyn ? words - cfg_dyn : 0u));
        if (grid_w.y != 0u) {
            const uint32_t rg = (uint32_t)(L.num_cus > 0 ? L.num_cus : 256) * 8u;
            hipLaunchKernelGGL(k_coarse_relocate, dim3(rg), blk, 0, L.stream, cfg, bump, tmpbuf, ptcl, tiles, (const uint32_t*)cnt_chunk, (const uint32_t*)cnt_blend,
                               (const uint32_t*)base_seg, (const uint32_t*)base_chunk, (const uint32_t*)base_blend, (const uint32_t*)R.end_pos,
                               (const uint2*)R.owner, (const unsigned long long*)R.masks, (const uint32_t*)R.aux, (const uint32_t*)arena_used, row0, row1, n);
        }
        return 0;
    }
#endif
    if (clips) JH_COARSE(0, true, grid, 0u, ptcl); else JH_COARSE(0, false, grid, 0u, ptcl);
    if (grid_w.y == 0u) {  // an empty band: nobody to report the totals (otherwise the write pass's first workgroup does)
        hipLaunchKernelGGL(k_coarse_totals, dim3(1), blk, 0, L.stream, (const uint32_t*)wg_tot, n_wg, bump);
        return 0;
    }
    if (clips) JH_COARSE(1, true, grid_w, row0, ptcl); else JH_COARSE(1, false, grid_w, row0, ptcl);
#undef JH_COARSE
    return 0;
}
