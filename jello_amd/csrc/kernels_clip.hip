// kernels_clip.hip -- K9 clip_reduce and K10 clip_leaf for gfx950 (round 5: designed for wave64, not restated).
//
// What the two stages compute (orig/clip_reduce.wgsl:24-67, orig/clip_leaf.wgsl:80-207, shared/clip.wgsl:4-26): the clip
// stream is a sequence of BeginClip / EndClip records.  Every EndClip is paired with its BeginClip (its draw monoid is pointed
// at the BeginClip's path and scene data), every BeginClip gets the intersection of the path boxes of all layers open at that
// point, every EndClip the box of the layers that stay open behind it.  Results: `clip_bboxes`, two words of `draw_monoids`
// per EndClip; between the stages: one nesting summary per 256 records (`reduced`: EndClips that close something in front of
// the block, BeginClips the block leaves open) and the block's open BeginClips in stack order (`clip_els`).
//
// How it is done here.  The stream is cut into runs of 64 records, one per wave.
//   * Nesting depth in front of a record = a wave prefix sum (DPP), its running minimum a DPP min-scan.  A record's partner /
//     enclosing layer is the LAST BeginClip in front of it that sits one level below: one ballot per nesting level present in
//     the run gives the mask of that level's BeginClips, a lane masks it to the lanes below its own and counts leading zeros.
//     No search tree, no barrier.  Levels are visited bottom-up, so the box of a lane's enclosing layer is final when the lane
//     reads it (wave-private LDS, 16 bytes per lane).
//   * What a run cannot resolve lies in the stack it starts on.  A record's chain of enclosing layers leaves the run at the
//     running minimum of the depth, so ONE lookup per record completes it: the full box of the stack entry at that level.
//   * The stack a block starts on is rebuilt once per block by all four waves: thread = earlier block, the nesting summaries
//     are joined by a DPP scan running against the stream (the summaries form a non-commutative semigroup; the lane order is
//     reversed so that the scan's own direction is the suffix direction), which gives every earlier block the number of its
//     open layers that survive up to this block and where they sit in the stack.  The top entries the block can reach
//     (at most one per EndClip: 256) are fetched individually and intersected by one more scan; everything deeper only
//     matters as one box, accumulated per thread and reduced once.  There is no limit on the depth of the stack or on the
//     number of blocks (the WGSL sees 256 entries / 256 blocks; its authors left a TODO for the rest).
//   * The three runs in front of a wave's own in its block are covered by a three-step walk over their open lists in LDS.
// One barrier in clip_reduce, five in clip_leaf (the WGSL: 18 and 50).
#include "kcommon.h"

using namespace jk;
using namespace jd;

namespace {

// Nesting summary of a stretch of the stream: EndClips that close layers opened in front of it, BeginClips it leaves open.
struct Nest {
    uint32_t closes, opens;
};
// x in front of y (shared/clip.wgsl:9-12: the bicyclic semigroup)
JD Nest nest_join(Nest x, Nest y) {
    const uint32_t paired = umin_(x.opens, y.closes);
    return Nest{x.closes + y.closes - paired, x.opens + y.opens - paired};
}
struct Box {
    float x0, y0, x1, y1;
};
JD Box box_meet(Box a, Box b) { return Box{fmax_(a.x0, b.x0), fmax_(a.y0, b.y0), fmin_(a.x1, b.x1), fmin_(a.y1, b.y1)}; }  // shared/bbox.wgsl:21-23
JD Box box_everything() { return Box{-1e9f, -1e9f, 1e9f, 1e9f}; }
JD Box box_of_path(const JlPathBbox& pb) { return Box{(float)pb.x0, (float)pb.y0, (float)pb.x1, (float)pb.y1}; }

// ---- wave64 scans on the DPP path (row_shr 1/2/4/8, row_bcast 15/31); lanes without a source keep their own value ----
#define CLIP_DPP(old, v, ctrl, rows) __builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, rows, 0xf, false)
#define CLIP_SCAN_STEPS(STEP)       \
    STEP(JK_DPP_ROW_SHR(1), 0xf)    \
    STEP(JK_DPP_ROW_SHR(2), 0xf)    \
    STEP(JK_DPP_ROW_SHR(4), 0xf)    \
    STEP(JK_DPP_ROW_SHR(8), 0xf)    \
    STEP(JK_DPP_ROW_BCAST15, 0xa)   \
    STEP(JK_DPP_ROW_BCAST31, 0xc)

JD int32_t wave_running_min(int32_t v) {
#define STEP(ctrl, rows) v = imin_(v, CLIP_DPP(v, v, ctrl, rows));
    CLIP_SCAN_STEPS(STEP)
#undef STEP
    return v;
}
JD int32_t wave_running_max(int32_t v) {
#define STEP(ctrl, rows) v = imax_(v, CLIP_DPP(v, v, ctrl, rows));
    CLIP_SCAN_STEPS(STEP)
#undef STEP
    return v;
}
// Inclusive join over lanes 0..own, the own lane's stretch IN FRONT of the lower lanes' (the caller lays the stream out against
// the lanes).  Lanes without a source receive the neutral element (0, 0).
JD Nest wave_nest_scan(Nest v) {
#define STEP(ctrl, rows) v = nest_join(v, Nest{(uint32_t)CLIP_DPP(0, v.closes, ctrl, rows), (uint32_t)CLIP_DPP(0, v.opens, ctrl, rows)});
    CLIP_SCAN_STEPS(STEP)
#undef STEP
    return v;
}
#define CLIP_DPP_F(v, ctrl, rows) __int_as_float(CLIP_DPP(__float_as_int(v), __float_as_int(v), ctrl, rows))
JD Box wave_box_scan(Box v) {
#define STEP(ctrl, rows) \
    v = box_meet(v, Box{CLIP_DPP_F(v.x0, ctrl, rows), CLIP_DPP_F(v.y0, ctrl, rows), CLIP_DPP_F(v.x1, ctrl, rows), CLIP_DPP_F(v.y1, ctrl, rows)});
    CLIP_SCAN_STEPS(STEP)
#undef STEP
    return v;
}
JD Nest lane_below(Nest v) {  // the value of lane - 1; lane 0 receives the neutral element
    Nest r{(uint32_t)__shfl_up((int)v.closes, 1, 64), (uint32_t)__shfl_up((int)v.opens, 1, 64)};
    if (lane_id() == 0u) r = Nest{0u, 0u};
    return r;
}
JD Nest nest_of_lane(Nest v, int lane) { return Nest{(uint32_t)__builtin_amdgcn_readlane((int)v.closes, lane), (uint32_t)__builtin_amdgcn_readlane((int)v.opens, lane)}; }

constexpr uint32_t BLOCK = 256u;  // records per nesting summary (the reference's workgroup: the layout of `reduced` / `clip_els`)
constexpr uint32_t RUN = 64u;     // records per wave
constexpr uint32_t RUNS = BLOCK / RUN;
constexpr uint32_t NOBODY = 0xffffffffu;

// -------------------------------------------------------------------------------------------------------------------------
// K9.  One block of 256 records per workgroup, laid out AGAINST the threads (thread t holds record 255 - t), so that the
// scan's direction is "everything behind me".  A BeginClip stays open iff nothing behind it closes past it; its place in the
// block's open list is the number of open ones in front of it.
// -------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_clip_reduce(Buf<JlClipInp> clip_inp, Buf<JlPathBbox> path_bboxes, Buf<JlClipBic> reduced,
                                                       Buf<JlClipEl> clip_out) {
    __shared__ Nest sh_run[RUNS];
    const uint32_t t = threadIdx.x, w = t >> 6;
    const uint32_t rec = blockIdx.x * BLOCK + (BLOCK - 1u - t);
    const int32_t what = clip_inp.rd(rec).path_ix;  // >= 0: BeginClip of that path
    const bool begins = what >= 0;
    JlPathBbox pb;
    if (begins) pb = path_bboxes.rd((uint32_t)what);
    const Nest from_me = wave_nest_scan(Nest{begins ? 0u : 1u, begins ? 1u : 0u});  // this record .. last record of the wave
    if (lane_id() == RUN - 1u) sh_run[w] = from_me;
    Nest behind_me = lane_below(from_me);
    __syncthreads();
    Nest whole{0u, 0u};
#pragma unroll
    for (uint32_t i = 0; i < RUNS; i++) {  // wave i holds records in front of those of the waves below it
        const Nest r = sh_run[i];
        if (i == w) behind_me = nest_join(behind_me, whole);
        whole = nest_join(r, whole);
    }
    if (t == 0u) reduced.wr(blockIdx.x, JlClipBic{whole.closes, whole.opens});
    if (begins && behind_me.closes == 0u) {
        JlClipEl el;
        el.parent_ix = rec;
        el.pad[0] = 0u; el.pad[1] = 0u; el.pad[2] = 0u;
        el.bbox[0] = (float)pb.x0; el.bbox[1] = (float)pb.y0; el.bbox[2] = (float)pb.x1; el.bbox[3] = (float)pb.y1;
        clip_out.wr(blockIdx.x * BLOCK + (whole.opens - behind_me.opens - 1u), el);
    }
}

// -------------------------------------------------------------------------------------------------------------------------
// K10.
// -------------------------------------------------------------------------------------------------------------------------
struct LeafShared {
    // per run (wave-private until the first barrier)
    Box layer_box[RUNS][RUN];     // lane's box within its run: its own path box met with those of the enclosing layers of the run
    int32_t layer_up[RUNS][RUN];  // lane of the enclosing BeginClip in the run, -1: it lies in front of the run
    Nest run_nest[RUNS];
    uint32_t open_rec[RUNS][RUN];  // the run's open BeginClips, bottom first: record index ...
    Box open_box[RUNS][RUN];       // ... and box (met with the open ones below it in the run)
    // the stack the block starts on: entry 0 = top
    uint32_t top_src[BLOCK];       // (earlier block << 8 | place in its open list) of the entries this block can reach
    uint32_t top_rec[BLOCK];       // their record indices, NOBODY past the bottom of the stack
    Box top_box[BLOCK + 1u];       // full boxes (met with everything below); [n] with n = entries reachable: the rest of the stack
    Nest carry;
    Nest part_nest[RUNS];
    Box part_box[RUNS], part_rest[RUNS];
};

// Box (and, for below = true, record index) of stack entry t (0 = top) as the stack stands at the start of run w of this block:
// the open lists of the runs w-1 .. 0, then the block's own starting stack.  below: the box of what lies UNDER that entry.
JD Box stack_lookup(const LeafShared& sh, uint32_t w, uint32_t t, bool below, uint32_t* rec) {
    Box acc = box_everything();
    bool found = false;
    *rec = NOBODY;
    for (uint32_t r = w; r-- > 0u;) {
        const Nest n = sh.run_nest[r];
        if (t < n.opens) {
            const uint32_t q = n.opens - 1u - t;
            if (below && !found) {
                *rec = sh.open_rec[r][q];
                if (q > 0u) acc = box_meet(acc, sh.open_box[r][q - 1u]);
            } else {
                acc = box_meet(acc, sh.open_box[r][q]);
            }
            found = true;
            t = n.closes;  // go on with what the run itself started on, under the layers it closed
        } else {
            t = n.closes + (t - n.opens);
        }
    }
    if (below && !found) {
        if (t < BLOCK) *rec = sh.top_rec[t];
        t += 1u;
    }
    return box_meet(acc, sh.top_box[umin_(t, BLOCK)]);
}

__global__ __launch_bounds__(BLOCK) void k_clip_leaf(const JlConfig* __restrict__ cfg, Buf<JlClipInp> clip_inp, Buf<JlPathBbox> path_bboxes,
                                                     Buf<JlClipBic> reduced, Buf<JlClipEl> clip_els, Buf<JlDrawMonoid> draw_monoids,
                                                     Buf<Box> clip_bboxes) {
    __shared__ LeafShared sh;
    const uint32_t t = threadIdx.x, w = t >> 6, lane = t & 63u;
    const uint32_t n_clip = cfg->layout.n_clip;
    const uint32_t rec = blockIdx.x * BLOCK + t;
    const bool live = rec < n_clip;

    // ---- 1. the run on its own ---------------------------------------------------------------------------------------
    const int32_t what = live ? clip_inp.rd(rec).path_ix : -1;
    const bool begins = live && what >= 0, ends = live && what < 0;
    Box own = box_everything();
    if (begins) own = box_of_path(path_bboxes.rd((uint32_t)what));
    sh.top_src[t] = NOBODY;

    const int32_t step = begins ? 1 : (ends ? -1 : 0);
    const int32_t depth = (int32_t)wave_incl_scan_u32((uint32_t)step) - step;  // levels open in front of this record, relative to the run's start
    const int32_t lowest = wave_running_min(depth);                            // the lowest level the run has been on so far
    const int32_t after = depth + step;
    const int32_t lowest_after = __builtin_amdgcn_readlane(wave_running_min(after), 63);
    const int32_t net = __builtin_amdgcn_readlane(after, 63);
    const uint32_t run_closes = (uint32_t)imax_(0, -lowest_after);
    const Nest run{run_closes, (uint32_t)(net + (int32_t)run_closes)};
    const int32_t level_lo = __builtin_amdgcn_readlane(lowest, 63), level_hi = __builtin_amdgcn_readlane(wave_running_max(depth), 63);

    const uint64_t lanes_below = (1ull << lane) - 1ull, lanes_above = ~lanes_below << 1;
    int32_t up = -1;         // the BeginClip one level down in front of this record (its partner for an EndClip), as a lane of the run
    bool stays_open = false;
    Box box = own;
    sh.layer_box[w][lane] = own;
    for (int32_t level = level_lo; level <= level_hi; level++) {
        const uint64_t begins_here = __builtin_amdgcn_ballot_w64(begins && depth == level);
        if (begins_here == 0ull) continue;
        const uint64_t returns_here = __builtin_amdgcn_ballot_w64(ends && depth - 1 == level);
        if (begins && depth == level) stays_open = (returns_here & lanes_above) == 0ull;
        const uint64_t mine = begins_here & lanes_below;
        wave_sync();  // the boxes of this level's BeginClips were stored on the trip before
        if (depth - 1 == level && mine != 0ull) {
            up = 63 - (int32_t)__builtin_clzll(mine);
            if (begins) {
                box = box_meet(sh.layer_box[w][up], own);
                sh.layer_box[w][lane] = box;
            }
        }
    }
    wave_sync();
    sh.layer_up[w][lane] = up;
    if (lane == 0u) sh.run_nest[w] = run;
    if (begins && stays_open) {
        const uint32_t place = (uint32_t)(depth + (int32_t)run.closes) & (RUN - 1u);
        sh.open_rec[w][place] = rec;
        sh.open_box[w][place] = box;
    }
    wave_sync();
    // an EndClip whose partner is in the run: the layer that stays open behind it, if that one is in the run as well
    int32_t up2 = -1;
    Box box2 = box_everything();
    if (ends && up >= 0) {
        up2 = sh.layer_up[w][up];
        if (up2 >= 0) box2 = sh.layer_box[w][up2];
    }
    __syncthreads();

    // ---- 2. the stack this block starts on (all four waves) ----------------------------------------------------------
    const Nest block_nest = nest_join(nest_join(nest_join(sh.run_nest[0], sh.run_nest[1]), sh.run_nest[2]), sh.run_nest[3]);
    const uint32_t reach = block_nest.closes;  // entries of the stack the block's EndClips can get at
    Nest carry{0u, 0u};                        // summary of the earlier blocks already visited (they are behind the ones still to come)
    Box rest = box_everything();               // what this thread has seen of the stack under the reachable entries
    for (uint32_t hi = blockIdx.x; hi > 0u; hi -= umin_(hi, BLOCK)) {
        // thread t looks at block hi - 1 - t: the scan runs towards the start of the stream
        const bool have = t < hi;
        const uint32_t src = hi - 1u - t;
        Nest mine{0u, 0u};
        if (have) {
            const JlClipBic b = reduced.rd(src);
            mine = Nest{umin_(b.a, BLOCK), umin_(b.b, BLOCK)};  // a block of 256 records cannot close or open more (bounds the loops below)
        }
        const Nest incl = wave_nest_scan(mine);
        Nest behind = lane_below(incl);  // the blocks between `src` and this wave's first
        if (lane == RUN - 1u) sh.part_nest[w] = incl;
        __syncthreads();
        Nest group{0u, 0u};
#pragma unroll
        for (uint32_t i = 0; i < RUNS; i++) {  // wave i of this group holds blocks in front of those of the waves below it
            if (i == w) behind = nest_join(behind, group);
            group = nest_join(sh.part_nest[i], group);
        }
        behind = nest_join(behind, carry);
        carry = nest_join(group, carry);
        if (have) {
            // of the block's open layers, the top `closed` ones are closed again before this block starts
            const uint32_t closed = umin_(mine.opens, behind.closes);
            const uint32_t alive = mine.opens - closed;
            const uint32_t first = behind.opens;  // stack entry of its topmost surviving layer
            const uint32_t in_reach = first < reach ? umin_(alive, reach - first) : 0u;
            for (uint32_t i = 0; i < in_reach; i++) sh.top_src[first + i] = (src << 8) | (alive - 1u - i);
            for (uint32_t q = 0; q + in_reach < alive; q++) {
                const JlClipEl el = clip_els.rd(src * BLOCK + q);
                rest = box_meet(rest, Box{el.bbox[0], el.bbox[1], el.bbox[2], el.bbox[3]});
            }
        }
        __syncthreads();  // part_nest is written again by the next group
    }
    __syncthreads();
    {
        // thread t takes entry 255 - t, so that the scan adds the entries below to the ones above
        const uint32_t entry = BLOCK - 1u - t;
        const uint32_t from = sh.top_src[entry];
        Box b = box_everything();
        uint32_t r = NOBODY;
        if (from != NOBODY) {
            const JlClipEl el = clip_els.rd((from >> 8) * BLOCK + (from & 255u));
            r = el.parent_ix;
            b = Box{el.bbox[0], el.bbox[1], el.bbox[2], el.bbox[3]};
        }
        sh.top_rec[entry] = r;
        b = wave_box_scan(b);
        const Box rest_wave = wave_box_scan(rest);
        if (lane == RUN - 1u) {
            sh.part_box[w] = b;
            sh.part_rest[w] = rest_wave;
        }
        __syncthreads();
        Box under = box_meet(box_meet(sh.part_rest[0], sh.part_rest[1]), box_meet(sh.part_rest[2], sh.part_rest[3]));
        if (t == 0u) sh.top_box[BLOCK] = under;
#pragma unroll
        for (uint32_t i = 0; i < RUNS; i++)
            if (i < w) under = box_meet(under, sh.part_box[i]);
        sh.top_box[entry] = box_meet(b, under);
    }
    __syncthreads();

    // ---- 3. one lookup per record completes it -----------------------------------------------------------------------
    if (!live) return;
    // The chain of enclosing layers leaves the run at level lowest - 1, stack entry -lowest of the run's starting stack.  An
    // EndClip whose partner is that entry itself wants the partner's record and the box of what lies under it.
    const bool partner_outside = ends && up < 0;
    uint32_t partner = NOBODY;
    const Box outer = stack_lookup(sh, w, (uint32_t)(-lowest), partner_outside, &partner);
    Box result;
    if (begins) {
        result = box_meet(outer, box);
    } else {
        result = box_meet(outer, box2);
        if (!partner_outside) partner = blockIdx.x * BLOCK + w * RUN + (uint32_t)up;
        if (partner != NOBODY) {
            // the EndClip draws with its BeginClip's path and scene data (clip_leaf.wgsl:185-191)
            const JlClipInp opener = clip_inp.rd(partner);
            const uint32_t drawobj = ~(uint32_t)what;
            if (draw_monoids.ok(drawobj)) {
                draw_monoids.p[drawobj].path_ix = (uint32_t)opener.path_ix;
                draw_monoids.p[drawobj].scene_offset = draw_monoids.rd(opener.ix).scene_offset;
            }
        }
    }
    clip_bboxes.wr(rec, result);
}

}  // namespace

// [clip_inp, path_bboxes, reduced(bics), clip_out(els)]
int jh_launch_clip_reduce(const JhLaunch& L) {
    if (L.nb < 4) return -1;
    if (L.gx == 0) return 0;
    hipLaunchKernelGGL(k_clip_reduce, dim3(L.gx), dim3(BLOCK), 0, L.stream, mkbuf<JlClipInp>(L.b[0].ptr, L.b[0].size),
                       mkbuf<JlPathBbox>(L.b[1].ptr, L.b[1].size), mkbuf<JlClipBic>(L.b[2].ptr, L.b[2].size),
                       mkbuf<JlClipEl>(L.b[3].ptr, L.b[3].size));
    return 0;
}
// [config, clip_inp, path_bboxes, reduced, clip_els, draw_monoids, clip_bboxes]
int jh_launch_clip_leaf(const JhLaunch& L) {
    if (L.nb < 7) return -1;
    if (L.gx == 0) return 0;
    hipLaunchKernelGGL(k_clip_leaf, dim3(L.gx), dim3(BLOCK), 0, L.stream, (const JlConfig*)L.b[0].ptr, mkbuf<JlClipInp>(L.b[1].ptr, L.b[1].size),
                       mkbuf<JlPathBbox>(L.b[2].ptr, L.b[2].size), mkbuf<JlClipBic>(L.b[3].ptr, L.b[3].size),
                       mkbuf<JlClipEl>(L.b[4].ptr, L.b[4].size), mkbuf<JlDrawMonoid>(L.b[5].ptr, L.b[5].size),
                       mkbuf<Box>(L.b[6].ptr, L.b[6].size));
    return 0;
}
