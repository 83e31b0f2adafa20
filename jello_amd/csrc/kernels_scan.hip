// kernels_scan.hip -- monoid scans of the element pipeline on gfx950:
//   K1-K4 pathtag_reduce / reduce2 / scan1 / scan_{small,large}  (orig/pathtag_*.wgsl)
//   K5    bbox_clear                                              (orig/bbox_clear.wgsl:13-24)
//   K7-K8 draw_reduce / draw_leaf                                 (orig/draw_reduce.wgsl, draw_leaf.wgsl)
//   plus the generic u32 exclusive scan every deterministic allocator is built on.
// All of these are HBM-bound integer work: 16 B / tag word in, 20 B out (K4); wave64 DPP
// prefix (row_shr + row_bcast) + a 4-entry LDS exchange per 256-thread block replaces the WGSL's 8-round LDS ladder.
#include "kcommon.h"
#include <cstring>

using namespace jk;
using namespace jd;

// ------------------------------------------------------------------------------------------------
// generic exclusive scan (u32)
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Single-pass scan with decoupled look-back (Merrill & Garland 2016), the form the north star asks for: ONE launch, every
// workgroup scans a tile of LB_TILE elements and gets the sum of everything before it from the tiles in front of it.
//   * tile ids are handed out by an atomic counter in start order, so a tile only ever waits for tiles that already run
//     (no deadlock whatever order the hardware starts workgroups in);
//   * a tile publishes ONE 64-bit descriptor -- value << 2 | status (1 = the tile's own sum, 2 = the inclusive prefix up
//     to and including it) -- with a relaxed device-scope store: value and validity travel in one word, no fence;
//   * wave 0 of the workgroup looks back 64 descriptors at a time (one per lane): the nearest INCLUSIVE one ends the
//     look-back, every descriptor between it and the tile must be valid (else re-read), their values are summed by the wave;
//     then the tile publishes its inclusive prefix, which lets its successors stop early;
//   * only the tiles that hold elements take part (the element count may live on the device); the last of them to finish
//     puts the descriptors and both counters back to zero, so the next launch (or hipGraph replay) needs no reset.
// Round 3, C3 on MI355X (profiles/r03_scan.md): 12.8 us (2.4 M slots) and 16.6 us (3.3 M lines) per scan against 19.6 us
// for the two-launch scan of round 2 (512 ranges reduced, then carry-in + scan; deleted in round 4 together with its
// JH_SCAN_LOOKBACK=0 switch -- an alternate nothing tested); with 4 K-element tiles it LOST (20.6 / 25.8 us), and reading
// four or eight descriptors per lane and round made it slower still.  Round 2's other single-launch attempts (a grid
// barrier on an arrival counter: 26.7 us; every workgroup summing ALL descriptors before its own: 19.6-21.1 us) are in DESIGN 3.
// Memory model (round 4): the descriptor carries value AND validity in one 64-bit word, so publishing and reading it are
// relaxed agent-scope atomics by right -- nothing else travels through it.  The one hand-off that orders DIFFERENT
// addresses is the reset by the last tile, and that is a release / acquire read-modify-write (below).
#ifndef LB_ITEMS
#define LB_ITEMS 64  // elements per thread: 16 K elements per tile (fewer, larger tiles: every look-back round is a device-coherent
#endif               // round trip of ~2 us on this 8-XCD part, and tile i needs ~i/64 of them until prefixes have spread)
#ifndef LB_LOOK
#define LB_LOOK 1    // descriptors per lane and look-back round
#endif
#define LB_TILE (JL_WG * LB_ITEMS)
JD unsigned long long lb_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
JD void lb_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(JL_WG) void k_scan_lookback(const uint32_t* __restrict__ in, uint32_t stride, uint32_t* __restrict__ out, uint32_t n_max,
                                                         const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ total_dev,
                                                         uint32_t* __restrict__ ctrl,            // [0] next tile id, [1] tiles finished
                                                         unsigned long long* __restrict__ desc) {
    __shared__ uint32_t sh[8];
    __shared__ uint32_t s_tile, s_prefix, s_last;
    const uint32_t n = n_dev ? umin_(*n_dev, n_max) : n_max;
    const uint32_t n_tiles = (n + LB_TILE - 1u) / LB_TILE;
    if (n_tiles == 0u) {
        if (blockIdx.x == 0u && threadIdx.x == 0u && total_dev) *total_dev = 0u;
        return;
    }
    if (blockIdx.x >= n_tiles) return;  // (uniform per workgroup: the counters only ever see n_tiles workgroups)
    if (threadIdx.x == 0u) s_tile = __hip_atomic_fetch_add(&ctrl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * LB_TILE + threadIdx.x * LB_ITEMS;
    uint32_t v[LB_ITEMS];
    uint32_t s = 0u;
    if (stride == 1u && base + LB_ITEMS <= n) {
        const uint4* p = (const uint4*)(in + base);  // (base is a multiple of 16 elements: 64-byte aligned when `in` is)
#pragma unroll
        for (int q = 0; q < LB_ITEMS / 4; q++) {
            const uint4 t = p[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < LB_ITEMS; i++) v[i] = base + i < n ? in[(size_t)(base + i) * stride] : 0u;
    }
#pragma unroll
    for (int i = 0; i < LB_ITEMS; i++) s += v[i];
    uint32_t tot;
    uint32_t excl = block_excl_scan_u32(s, sh, &tot);
    if (threadIdx.x == 0u) lb_store(&desc[tile], ((unsigned long long)tot << 2) | (tile == 0u ? 2ull : 1ull));
    if (tile != 0u && threadIdx.x < 64u) {
        const uint32_t lane = threadIdx.x;
        uint32_t acc = 0u;
        int32_t first = (int32_t)tile - 1;  // the descriptor lane 0 looks at
        for (;;) {
            // each lane folds LB_LOOK consecutive descriptors (nearest first) into one: the sum up to and including the
            // first inclusive prefix among them (status 2), or of all of them (status 1), or "not published yet" (0)
            uint32_t status = 1u, val = 0u;
            unsigned long long d[LB_LOOK];
#pragma unroll
            for (int q = 0; q < LB_LOOK; q++) {
                const int32_t idx = first - (int32_t)(lane * LB_LOOK) - q;
                d[q] = idx >= 0 ? lb_load(&desc[idx]) : 2ull;  // (in front of tile 0: an inclusive prefix of 0)
            }
#pragma unroll
            for (int q = 0; q < LB_LOOK; q++) {
                const uint32_t st = (uint32_t)d[q] & 3u;
                if (status == 1u) {
                    if (st == 0u) status = 0u;
                    else { val += (uint32_t)(d[q] >> 2); status = st; }
                }
            }
            const uint64_t incl = __builtin_amdgcn_ballot_w64(status == 2u);
            const uint64_t invalid = __builtin_amdgcn_ballot_w64(status == 0u);
            const uint32_t stop = incl != 0ull ? (uint32_t)__builtin_ctzll(incl) : 63u;  // the window is lanes 0..stop
            const uint64_t window = stop >= 63u ? ~0ull : ((2ull << stop) - 1ull);
            if ((invalid & window) != 0ull) { __builtin_amdgcn_s_sleep(1); continue; }  // a tile in the window has not published yet
            acc += wave_reduce_u32(lane <= stop ? val : 0u);
            if (incl != 0ull) break;
            first -= 64 * LB_LOOK;
        }
        if (lane == 0u) {
            s_prefix = acc;
            lb_store(&desc[tile], ((unsigned long long)(acc + tot) << 2) | 2ull);
        }
    }
    __syncthreads();
    const uint32_t prefix = tile != 0u ? s_prefix : 0u;
    excl += prefix;
    // Every tile counts itself out as soon as it has published its inclusive prefix and reads no descriptor any more -- BEFORE
    // it writes its share of `out`, which the release would otherwise have to wait for (and flush): 15.6 us per scan with the
    // count-out at the end of the kernel against 11.3 us with relaxed atomics.  The last one to count out resets the state.
    // Release: this tile's descriptor accesses are done before it counts out; acquire: the last tile's resets come after every
    // other tile's count -- the hand-off does not rest on instruction order.
    if (threadIdx.x == 0u) {
        if (tile == n_tiles - 1u && total_dev) *total_dev = prefix + tot;
        const uint32_t gone = __hip_atomic_fetch_add(&ctrl[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = gone + 1u == n_tiles ? 1u : 0u;
    }
    if (base + LB_ITEMS <= n) {
        uint4* q = (uint4*)(out + base);
#pragma unroll
        for (int i = 0; i < LB_ITEMS / 4; i++) {
            uint4 t;
            t.x = excl; excl += v[4 * i];
            t.y = excl; excl += v[4 * i + 1];
            t.z = excl; excl += v[4 * i + 2];
            t.w = excl; excl += v[4 * i + 3];
            q[i] = t;
        }
    } else {
#pragma unroll
        for (int i = 0; i < LB_ITEMS; i++) {
            if (base + i < n) out[base + i] = excl;
            excl += v[i];
        }
    }
    __syncthreads();
    if (s_last != 0u) {  // (uniform) the last tile to count out: nobody reads a descriptor any more
        for (uint32_t i = threadIdx.x; i < n_tiles; i += JL_WG) lb_store(&desc[i], 0ull);
        if (threadIdx.x == 0u) {
            __hip_atomic_store(&ctrl[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctrl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int jh_scan_u32(const JhLaunch& L, const uint32_t* in, uint32_t in_stride, uint32_t* out, uint32_t n_max, const uint32_t* n_dev,
                uint32_t* total_dev) {
    const uint64_t max_tiles = ((uint64_t)n_max + LB_TILE - 1u) / LB_TILE;
    // [ctrl: 256 bytes][one descriptor per tile]: zero between launches (the kernel cleans up after itself; the flag says
    // whether that has happened since the slot was allocated)
    char* st = (char*)jh_scratch_get(L.scratch, JH_SCR_SCAN_TMP, 256u + (max_tiles + 1u) * 8u);
    if (!st) return -5;
    uint32_t* clean = jh_scratch_flags(L.scratch);
    // (the WHOLE slot: a later scan of the frame may use more descriptors of the same allocation than this one)
    if ((*clean & JH_CLEAN_SCAN) == 0u) (void)hipMemsetAsync(st, 0, jh_scratch_cap(L.scratch, JH_SCR_SCAN_TMP), L.stream);
    *clean |= JH_CLEAN_SCAN;
    if (max_tiles == 0u) {
        if (total_dev) (void)hipMemsetAsync(total_dev, 0, 4, L.stream);
        return 0;
    }
    hipLaunchKernelGGL(k_scan_lookback, dim3((uint32_t)max_tiles), dim3(JL_WG), 0, L.stream, in, in_stride, out, n_max, n_dev, total_dev,
                       (uint32_t*)st, (unsigned long long*)(st + 256));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// pathtag (K1-K4)
// ------------------------------------------------------------------------------------------------
// (store_tm / load_tm / parent_prefix: kcommon.h -- flatten's classification kernel can stand in for the last scan)

// pathtag_reduce.wgsl:21-42
__global__ __launch_bounds__(JL_WG) void k_pathtag_reduce(const JlConfig* __restrict__ cfg, Buf<uint32_t> scene, Buf<JlTagMonoid> reduced) {
    __shared__ uint32_t sh[20];
    uint32_t ix = blockIdx.x * JL_WG + threadIdx.x;
    MonoidK<5> agg = reduce_tag(scene.rd(cfg->layout.pathtag_base + ix));
    MonoidK<5> t = block_reduce_monoid<5>(agg, sh);
    if (threadIdx.x == 0 && reduced.ok(blockIdx.x)) store_tm(&reduced.p[blockIdx.x], t);
}
// pathtag_reduce2.wgsl:23-41
__global__ __launch_bounds__(JL_WG) void k_pathtag_reduce2(Buf<JlTagMonoid> in, Buf<JlTagMonoid> out) {
    __shared__ uint32_t sh[20];
    uint32_t ix = blockIdx.x * JL_WG + threadIdx.x;
    MonoidK<5> t = block_reduce_monoid<5>(load_tm(in, ix), sh);
    if (threadIdx.x == 0 && out.ok(blockIdx.x)) store_tm(&out.p[blockIdx.x], t);
}
// Sum of parent[l] for l < wg (l < 256): the WGSL "reduce prefix of workgroups up to this one".
// pathtag_scan1.wgsl:26-67
// WITH_REDUCE2: the held-back pathtag_reduce2 dispatch rides along (a launch of its own is 4.5 us for four workgroups' worth of work).
// reduced2[w] is the sum of the w-th 256 entries of `reduced` -- exactly the block total this kernel's own scan produces -- and the
// prefix of workgroup w the sum of the entries in front of its 256, which it adds up itself (integer sums: any association gives
// the WGSL's words).  The reference dispatches reduce2 with 256 workgroups whatever the scene (render.go:186-190); the entries
// of reduced2 behind this grid -- sums over the unused (or absent: robust reads) tail of `reduced` -- are written by workgroup 0,
// one per thread.  Only for grids of at most PT_ABSORB_MAX workgroups (w * 256 loads per workgroup); the dispatcher decides.
#define PT_ABSORB_MAX 16u
template <bool WITH_REDUCE2>
__global__ __launch_bounds__(JL_WG) void k_pathtag_scan1(Buf<JlTagMonoid> reduced, Buf<JlTagMonoid> reduced2, Buf<JlTagMonoid> out, uint32_t n_red2) {
    __shared__ uint32_t sh[20];
    MonoidK<5> prefix;
    if (WITH_REDUCE2) {
        MonoidK<5> agg;
#pragma unroll
        for (int i = 0; i < 5; i++) agg.v[i] = 0;
        for (uint32_t i = threadIdx.x; i < blockIdx.x * JL_WG; i += JL_WG) agg = monoid_add(agg, load_tm(reduced, i));
        prefix = block_reduce_monoid<5>(agg, sh);
        __syncthreads();
    } else {
        prefix = parent_prefix(reduced2, blockIdx.x, sh);
    }
    uint32_t ix = blockIdx.x * JL_WG + threadIdx.x;
    MonoidK<5> tot;
    MonoidK<5> ex = block_excl_scan_monoid<5>(load_tm(reduced, ix), sh, &tot);
    if (out.ok(ix)) store_tm(&out.p[ix], monoid_add(prefix, ex));
    if (WITH_REDUCE2) {
        if (threadIdx.x == 0 && blockIdx.x < n_red2 && reduced2.ok(blockIdx.x)) store_tm(&reduced2.p[blockIdx.x], tot);
        const uint32_t w = threadIdx.x;  // (n_red2 <= 256: one thread per entry behind the grid)
        if (blockIdx.x == 0u && w >= gridDim.x && w < n_red2 && reduced2.ok(w)) {
            MonoidK<5> agg;
#pragma unroll
            for (int i = 0; i < 5; i++) agg.v[i] = 0;
            for (uint32_t i = w * JL_WG; i < (w + 1u) * JL_WG && i < reduced.n; i++) agg = monoid_add(agg, load_tm(reduced, i));
            store_tm(&reduced2.p[w], agg);
        }
    }
}
// (Round 3 ran pathtag_reduce + pathtag_reduce2 + pathtag_scan1 as ONE launch behind a finished-workgroups counter: 11.9 us
// against ~14 us for the three, with the hand-off ordered by relaxed atomics around an s_waitcnt.  Written as the memory model
// wants it -- release on every workgroup's count, acquire on the last one's -- the launch takes 21.6 us on C3: a release at
// agent scope writes back the L2 of its XCD, 782 times.  Three launches it is again: 3 x 4.7 us)
// pathtag_scan.wgsl:28-76
template <bool SMALL>
__global__ __launch_bounds__(JL_WG) void k_pathtag_scan(const JlConfig* __restrict__ cfg, Buf<uint32_t> scene, Buf<JlTagMonoid> reduced,
                                                        Buf<JlTagMonoid> out) {
    __shared__ uint32_t sh[20];
    MonoidK<5> prefix;
    if (SMALL) prefix = parent_prefix(reduced, blockIdx.x, sh); else prefix = load_tm(reduced, blockIdx.x);
    uint32_t ix = blockIdx.x * JL_WG + threadIdx.x;
    MonoidK<5> tot;
    MonoidK<5> ex = block_excl_scan_monoid<5>(reduce_tag(scene.rd(cfg->layout.pathtag_base + ix)), sh, &tot);
    if (out.ok(ix)) store_tm(&out.p[ix], monoid_add(prefix, ex));
}

int jh_launch_pathtag(const JhLaunch& L, int stage) {
    if (L.gx == 0) return 0;
    dim3 g(L.gx), blk(JL_WG);
    switch (stage) {
        case 0:
            if (L.nb < 3) return -1;
            hipLaunchKernelGGL(k_pathtag_reduce, g, blk, 0, L.stream, (const JlConfig*)L.b[0].ptr, mkbuf<uint32_t>(L.b[1].ptr, L.b[1].size),
                               mkbuf<JlTagMonoid>(L.b[2].ptr, L.b[2].size));
            break;
        case 1:
            if (L.nb < 2) return -1;
            hipLaunchKernelGGL(k_pathtag_reduce2, g, blk, 0, L.stream, mkbuf<JlTagMonoid>(L.b[0].ptr, L.b[0].size),
                               mkbuf<JlTagMonoid>(L.b[1].ptr, L.b[1].size));
            break;
        case 2:
            if (L.nb < 3) return -1;
            // (JH_ABSORB_SETUP here: the dispatcher held pathtag_reduce2 back and found this dispatch to be its consumer)
            // (L.extra.size: the workgroup count of the held-back dispatch = entries of reduced2 it would have written)
            if ((L.absorb & JH_ABSORB_SETUP) != 0u)
                hipLaunchKernelGGL(k_pathtag_scan1<true>, g, blk, 0, L.stream, mkbuf<JlTagMonoid>(L.b[0].ptr, L.b[0].size),
                                   mkbuf<JlTagMonoid>(L.b[1].ptr, L.b[1].size), mkbuf<JlTagMonoid>(L.b[2].ptr, L.b[2].size), (uint32_t)L.extra.size);
            else
                hipLaunchKernelGGL(k_pathtag_scan1<false>, g, blk, 0, L.stream, mkbuf<JlTagMonoid>(L.b[0].ptr, L.b[0].size),
                                   mkbuf<JlTagMonoid>(L.b[1].ptr, L.b[1].size), mkbuf<JlTagMonoid>(L.b[2].ptr, L.b[2].size), 0u);
            break;
        case 3:
        case 4: {
            if (L.nb < 4) return -1;
            auto cfg = (const JlConfig*)L.b[0].ptr;
            auto scene = mkbuf<uint32_t>(L.b[1].ptr, L.b[1].size);
            auto red = mkbuf<JlTagMonoid>(L.b[2].ptr, L.b[2].size);
            auto out = mkbuf<JlTagMonoid>(L.b[3].ptr, L.b[3].size);
            if (stage == 3) hipLaunchKernelGGL(k_pathtag_scan<true>, g, blk, 0, L.stream, cfg, scene, red, out);
            else hipLaunchKernelGGL(k_pathtag_scan<false>, g, blk, 0, L.stream, cfg, scene, red, out);
            break;
        }
        default: return -1;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// bbox_clear (K5)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(JL_WG) void k_bbox_clear(const JlConfig* __restrict__ cfg, Buf<JlPathBbox> bb) {
    uint32_t ix = blockIdx.x * JL_WG + threadIdx.x;
    if (ix < cfg->layout.n_path && bb.ok(ix)) {
        bb.p[ix].x0 = 0x7fffffff;
        bb.p[ix].y0 = 0x7fffffff;
        bb.p[ix].x1 = (int32_t)0x80000000;
        bb.p[ix].y1 = (int32_t)0x80000000;
    }
}
int jh_launch_bbox_clear(const JhLaunch& L) {
    if (L.nb < 2) return -1;
    if (L.gx == 0) return 0;
    hipLaunchKernelGGL(k_bbox_clear, dim3(L.gx), dim3(JL_WG), 0, L.stream, (const JlConfig*)L.b[0].ptr, mkbuf<JlPathBbox>(L.b[1].ptr, L.b[1].size));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// draw_reduce / draw_leaf (K7-K8)
// ------------------------------------------------------------------------------------------------
JD uint32_t read_draw_tag(const JlConfig* cfg, const Buf<uint32_t>& scene, uint32_t ix) {  // shared/util.wgsl:15-24
    return ix < cfg->layout.n_drawobj ? scene.rd(cfg->layout.drawtag_base + ix) : 0u;
}
JD void store_dm(JlDrawMonoid* d, const MonoidK<4>& m) { d->path_ix = m.v[0]; d->clip_ix = m.v[1]; d->scene_offset = m.v[2]; d->info_offset = m.v[3]; }
JD MonoidK<4> load_dm(const Buf<JlDrawMonoid>& b, uint32_t i) {
    JlDrawMonoid t = b.rd(i);
    MonoidK<4> m;
    m.v[0] = t.path_ix; m.v[1] = t.clip_ix; m.v[2] = t.scene_offset; m.v[3] = t.info_offset;
    return m;
}

// draw_reduce.wgsl:22-55
__global__ __launch_bounds__(JL_WG) void k_draw_reduce(const JlConfig* __restrict__ cfg, Buf<uint32_t> scene, Buf<JlDrawMonoid> reduced) {
    __shared__ uint32_t sh[16];
    uint32_t num_blocks_total = (cfg->layout.n_drawobj + (JL_WG - 1u)) / JL_WG;
    uint32_t n_blocks_base = num_blocks_total / JL_WG;
    uint32_t remainder = num_blocks_total % JL_WG;
    uint32_t first_block = n_blocks_base * blockIdx.x + umin_(blockIdx.x, remainder);
    uint32_t n_blocks = n_blocks_base + (blockIdx.x < remainder ? 1u : 0u);
    uint32_t block_index = first_block * JL_WG + threadIdx.x;
    MonoidK<4> agg;
#pragma unroll
    for (int i = 0; i < 4; i++) agg.v[i] = 0;
    for (uint32_t i = 0; i < n_blocks; i++) {
        agg = monoid_add(agg, map_draw_tag(read_draw_tag(cfg, scene, block_index)));
        block_index += JL_WG;
    }
    MonoidK<4> t = block_reduce_monoid<4>(agg, sh);
    if (threadIdx.x == 0 && reduced.ok(blockIdx.x)) store_dm(&reduced.p[blockIdx.x], t);
}

JD Xf from_poly2(V2 p0, V2 p1) {  // draw_leaf.wgsl:279-284
    Xf r;
    r.m0 = p1.y - p0.y; r.m1 = p0.x - p1.x; r.m2 = p1.x - p0.x; r.m3 = p1.y - p0.y; r.t0 = p0.x; r.t1 = p0.y;
    return r;
}
JD Xf two_point_to_unit_line(V2 p0, V2 p1) {  // draw_leaf.wgsl:272-277
    Xf tmp1 = from_poly2(p0, p1);
    Xf inv = xf_inverse(tmp1);
    Xf tmp2 = from_poly2(v2(0.0f, 0.0f), v2(1.0f, 0.0f));
    return xf_mul(tmp2, inv);
}
JD Xf xf_scale(float sx, float sy) { Xf r; r.m0 = sx; r.m1 = 0.0f; r.m2 = 0.0f; r.m3 = sy; r.t0 = 0.0f; r.t1 = 0.0f; return r; }
JD Xf xf_read_guarded(const Buf<uint32_t>& scene, uint32_t transform_base, uint32_t ix) {
    uint32_t base = transform_base + ix * 6u;
    Xf r;
    r.m0 = u2f(scene.rd(base)); r.m1 = u2f(scene.rd(base + 1u)); r.m2 = u2f(scene.rd(base + 2u)); r.m3 = u2f(scene.rd(base + 3u));
    r.t0 = u2f(scene.rd(base + 4u)); r.t1 = u2f(scene.rd(base + 5u));
    return r;
}

// draw_leaf.wgsl:52-270
__global__ __launch_bounds__(JL_WG) void k_draw_leaf(const JlConfig* __restrict__ cfg, Buf<uint32_t> scene, Buf<JlDrawMonoid> reduced,
                                                     Buf<JlPathBbox> path_bbox, Buf<JlDrawMonoid> draw_monoid, Buf<uint32_t> info,
                                                     Buf<JlClipInp> clip_inp) {
    __shared__ uint32_t sh[16];
    MonoidK<4> agg;
#pragma unroll
    for (int i = 0; i < 4; i++) agg.v[i] = 0;
    if (threadIdx.x < blockIdx.x) agg = load_dm(reduced, threadIdx.x);
    MonoidK<4> prefix = block_reduce_monoid<4>(agg, sh);

    uint32_t num_blocks_total = (cfg->layout.n_drawobj + JL_WG - 1u) / JL_WG;
    uint32_t n_blocks_base = num_blocks_total / JL_WG;
    uint32_t remainder = num_blocks_total % JL_WG;
    uint32_t first_block = n_blocks_base * blockIdx.x + umin_(blockIdx.x, remainder);
    uint32_t n_blocks = n_blocks_base + (blockIdx.x < remainder ? 1u : 0u);
    uint32_t block_start = first_block * JL_WG;
    for (uint32_t blk = 0; blk < n_blocks; blk++, block_start += JL_WG) {
        uint32_t ix = block_start + threadIdx.x;
        uint32_t tag_word = read_draw_tag(cfg, scene, ix);
        MonoidK<4> tot;
        MonoidK<4> m = monoid_add(prefix, block_excl_scan_monoid<4>(map_draw_tag(tag_word), sh, &tot));
        if (ix < cfg->layout.n_drawobj && draw_monoid.ok(ix)) store_dm(&draw_monoid.p[ix], m);
        uint32_t dd = cfg->layout.drawdata_base + m.v[2];
        uint32_t di = m.v[3];
        if (tag_word == 0x50u || tag_word == 0x114u || tag_word == 0x29cu || tag_word == 0x254u || tag_word == 0x248u || tag_word == 0x9u) {
            JlPathBbox bbox = path_bbox.rd(m.v[0]);
            uint32_t draw_flags = bbox.draw_flags;
            Xf transform = xf_scale(0.0f, 0.0f);
            if (tag_word == 0x114u || tag_word == 0x29cu || tag_word == 0x254u || tag_word == 0x248u)
                transform = xf_read_guarded(scene, cfg->layout.transform_base, bbox.trans_ix);
            if (tag_word == 0x50u) {
                info.wr(di, draw_flags);
            } else if (tag_word == 0x114u) {
                info.wr(di, draw_flags);
                V2 p0 = v2(u2f(scene.rd(dd + 1u)), u2f(scene.rd(dd + 2u)));
                V2 p1 = v2(u2f(scene.rd(dd + 3u)), u2f(scene.rd(dd + 4u)));
                p0 = xf_apply(transform, p0);
                p1 = xf_apply(transform, p1);
                V2 dxy = p1 - p0;
                float scale = 1.0f / dot(dxy, dxy);
                V2 line_xy = dxy * scale;
                float line_c = -dot(p0, line_xy);
                info.wr(di + 1u, f2u(line_xy.x));
                info.wr(di + 2u, f2u(line_xy.y));
                info.wr(di + 3u, f2u(line_c));
            } else if (tag_word == 0x29cu) {
                const float GRADIENT_EPSILON = 1.0f / (float)(1u << 12);
                info.wr(di, draw_flags);
                V2 p0 = v2(u2f(scene.rd(dd + 1u)), u2f(scene.rd(dd + 2u)));
                V2 p1 = v2(u2f(scene.rd(dd + 3u)), u2f(scene.rd(dd + 4u)));
                float r0 = u2f(scene.rd(dd + 5u));
                float r1 = u2f(scene.rd(dd + 6u));
                Xf user_to_gradient = xf_inverse(transform);
                Xf xform = xf_scale(0.0f, 0.0f);
                float focal_x = 0.0f, radius = 0.0f;
                uint32_t kind = 0u, flags = 0u;
                if (abs_(r0 - r1) <= GRADIENT_EPSILON) {
                    kind = JL_RAD_GRAD_KIND_STRIP;
                    float scaled = r0 / length(p0 - p1);
                    xform = xf_mul(two_point_to_unit_line(p0, p1), user_to_gradient);
                    radius = scaled * scaled;
                } else {
                    kind = JL_RAD_GRAD_KIND_CONE;
                    if (veq(p0, p1)) {
                        kind = JL_RAD_GRAD_KIND_CIRCULAR;
                        p0 = v2(p0.x + GRADIENT_EPSILON, p0.y + GRADIENT_EPSILON);
                    }
                    if (r1 == 0.0f) {
                        flags |= JL_RAD_GRAD_SWAPPED;
                        V2 tmp_p = p0; p0 = p1; p1 = tmp_p;
                        float tmp_r = r0; r0 = r1; r1 = tmp_r;
                    }
                    focal_x = r0 / (r0 - r1);
                    V2 cf = (1.0f - focal_x) * p0 + focal_x * p1;
                    radius = r1 / length(cf - p1);
                    Xf user_to_unit_line = xf_mul(two_point_to_unit_line(cf, p1), user_to_gradient);
                    if (abs_(radius - 1.0f) <= GRADIENT_EPSILON) {
                        kind = JL_RAD_GRAD_KIND_FOCAL_ON_CIRCLE;
                        float scale = 0.5f * abs_(1.0f - focal_x);
                        xform = xf_mul(xf_scale(scale, scale), user_to_unit_line);
                    } else {
                        float a = radius * radius - 1.0f;
                        float scale_ratio = abs_(1.0f - focal_x) / a;
                        float scale_x = radius * scale_ratio;
                        float scale_y = sqrt_(abs_(a)) * scale_ratio;
                        xform = xf_mul(xf_scale(scale_x, scale_y), user_to_unit_line);
                    }
                }
                info.wr(di + 1u, f2u(xform.m0)); info.wr(di + 2u, f2u(xform.m1)); info.wr(di + 3u, f2u(xform.m2)); info.wr(di + 4u, f2u(xform.m3));
                info.wr(di + 5u, f2u(xform.t0)); info.wr(di + 6u, f2u(xform.t1));
                info.wr(di + 7u, f2u(focal_x));
                info.wr(di + 8u, f2u(radius));
                info.wr(di + 9u, (flags << 3) | kind);
            } else if (tag_word == 0x254u) {
                info.wr(di, draw_flags);
                V2 p0 = v2(u2f(scene.rd(dd + 1u)), u2f(scene.rd(dd + 2u)));
                Xf tr = xf_identity();
                tr.t0 = p0.x; tr.t1 = p0.y;
                Xf inv = xf_inverse(xf_mul(transform, tr));
                info.wr(di + 1u, f2u(inv.m0)); info.wr(di + 2u, f2u(inv.m1)); info.wr(di + 3u, f2u(inv.m2)); info.wr(di + 4u, f2u(inv.m3));
                info.wr(di + 5u, f2u(inv.t0)); info.wr(di + 6u, f2u(inv.t1));
                info.wr(di + 7u, scene.rd(dd + 3u));
                info.wr(di + 8u, scene.rd(dd + 4u));
            } else if (tag_word == 0x248u) {
                info.wr(di, draw_flags);
                Xf inv = xf_inverse(transform);
                info.wr(di + 1u, f2u(inv.m0)); info.wr(di + 2u, f2u(inv.m1)); info.wr(di + 3u, f2u(inv.m2)); info.wr(di + 4u, f2u(inv.m3));
                info.wr(di + 5u, f2u(inv.t0)); info.wr(di + 6u, f2u(inv.t1));
                info.wr(di + 7u, scene.rd(dd));
                info.wr(di + 8u, scene.rd(dd + 1u));
            }
        }
        if (tag_word == 0x9u || tag_word == 0x21u) {
            uint32_t path_ix = ~ix;
            if (tag_word == 0x9u) path_ix = m.v[0];
            JlClipInp ci;
            ci.ix = ix;
            ci.path_ix = (int32_t)path_ix;
            clip_inp.wr(m.v[1], ci);
        }
        prefix = monoid_add(prefix, tot);
    }
}

int jh_launch_draw_reduce(const JhLaunch& L) {
    if (L.nb < 3) return -1;
    if (L.gx == 0) return 0;
    hipLaunchKernelGGL(k_draw_reduce, dim3(L.gx), dim3(JL_WG), 0, L.stream, (const JlConfig*)L.b[0].ptr, mkbuf<uint32_t>(L.b[1].ptr, L.b[1].size),
                       mkbuf<JlDrawMonoid>(L.b[2].ptr, L.b[2].size));
    return 0;
}
int jh_launch_draw_leaf(const JhLaunch& L) {
    if (L.nb < 7) return -1;
    if (L.gx == 0) return 0;
    hipLaunchKernelGGL(k_draw_leaf, dim3(L.gx), dim3(JL_WG), 0, L.stream, (const JlConfig*)L.b[0].ptr, mkbuf<uint32_t>(L.b[1].ptr, L.b[1].size),
                       mkbuf<JlDrawMonoid>(L.b[2].ptr, L.b[2].size), mkbuf<JlPathBbox>(L.b[3].ptr, L.b[3].size),
                       mkbuf<JlDrawMonoid>(L.b[4].ptr, L.b[4].size), mkbuf<uint32_t>(L.b[5].ptr, L.b[5].size),
                       mkbuf<JlClipInp>(L.b[6].ptr, L.b[6].size));
    return 0;
}
