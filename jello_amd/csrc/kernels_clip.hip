// kernels_clip.hip -- K9 clip_reduce (orig/clip_reduce.wgsl:24-67) and K10 clip_leaf
// (orig/clip_leaf.wgsl:80-207): match every EndClip with its BeginClip through the bicyclic
// semigroup (shared/clip.wgsl:9-12), intersect clip bounding boxes along the parent chain and
// redirect each EndClip's draw monoid to its BeginClip's path / scene data.
// Integer + f32 min/max only, deterministic as written; this is a direct gfx950 restatement with the
// 256-wide LDS ladders kept (n_clip is small: 2 x layers), one 256-thread workgroup per 256 clips.
#include "kcommon.h"

using namespace jk;
using namespace jd;

namespace {

struct Bic { uint32_t a, b; };
struct Bb4 { float v[4]; };

JD Bic bic_combine(Bic x, Bic y) {
    uint32_t m = umin_(x.b, y.a);
    Bic r;
    r.a = x.a + y.a - m;
    r.b = x.b + y.b - m;
    return r;
}
JD Bb4 bbox_intersect(Bb4 a, Bb4 b) {
    Bb4 r;
    r.v[0] = fmax_(a.v[0], b.v[0]); r.v[1] = fmax_(a.v[1], b.v[1]); r.v[2] = fmin_(a.v[2], b.v[2]); r.v[3] = fmin_(a.v[3], b.v[3]);
    return r;
}
JD Bb4 bb_inf() { Bb4 r; r.v[0] = -1e9f; r.v[1] = -1e9f; r.v[2] = 1e9f; r.v[3] = 1e9f; return r; }

// Reverse (suffix) scan of the bicyclic semigroup over 256 LDS entries, as the WGSL ladder does.
JD Bic reverse_scan_bic(Bic bic, Bic* sh) {
    uint32_t lid = threadIdx.x;
    sh[lid] = bic;
    for (uint32_t i = 0; i < 8u; i++) {
        __syncthreads();
        if (lid + (1u << i) < JL_WG) {
            Bic other = sh[lid + (1u << i)];
            bic = bic_combine(bic, other);
        }
        __syncthreads();
        sh[lid] = bic;
    }
    return bic;
}

__global__ __launch_bounds__(JL_WG) void k_clip_reduce(Buf<JlClipInp> clip_inp, Buf<JlPathBbox> path_bboxes, Buf<JlClipBic> reduced,
                                                       Buf<JlClipEl> clip_out) {
    __shared__ Bic sh_bic[JL_WG];
    __shared__ uint32_t sh_parent[JL_WG];
    __shared__ uint32_t sh_path_ix[JL_WG];
    uint32_t lid = threadIdx.x, gid = blockIdx.x * JL_WG + lid;
    int32_t inp = clip_inp.rd(gid).path_ix;
    bool is_push = inp >= 0;
    Bic bic;
    bic.a = 1u - (is_push ? 1u : 0u);
    bic.b = is_push ? 1u : 0u;
    bic = reverse_scan_bic(bic, sh_bic);
    if (lid == 0u) {
        JlClipBic o;
        o.a = bic.a; o.b = bic.b;
        reduced.wr(blockIdx.x, o);
    }
    __syncthreads();
    uint32_t size = sh_bic[0].b;
    Bic nb;
    nb.a = 0u; nb.b = 0u;
    if (lid + 1u < JL_WG) nb = sh_bic[lid + 1u];
    if (is_push && nb.a == 0u) {
        uint32_t local_ix = size - nb.b - 1u;
        if (local_ix < JL_WG) {
            sh_parent[local_ix] = lid;
            sh_path_ix[local_ix] = (uint32_t)inp;
        }
    }
    __syncthreads();
    if (lid < size) {
        uint32_t path_ix = sh_path_ix[lid];
        JlPathBbox pb = path_bboxes.rd(path_ix);
        JlClipEl el;
        el.parent_ix = sh_parent[lid] + blockIdx.x * JL_WG;
        el.pad[0] = 0u; el.pad[1] = 0u; el.pad[2] = 0u;
        el.bbox[0] = (float)pb.x0; el.bbox[1] = (float)pb.y0; el.bbox[2] = (float)pb.x1; el.bbox[3] = (float)pb.y1;
        clip_out.wr(gid, el);
    }
}

// clip_leaf.wgsl:38-66
JD int32_t search_link(Bic* bic, uint32_t ix_in, const Bic* sh_bic) {
    uint32_t ix = ix_in;
    uint32_t j = 0u;
    while (j < 8u) {
        uint32_t base = 2u * JL_WG - (2u << (8u - j));
        if (((ix >> j) & 1u) != 0u) {
            Bic test = bic_combine(sh_bic[base + (ix >> j) - 1u], *bic);
            if (test.b > 0u) break;
            *bic = test;
            ix -= 1u << j;
        }
        j += 1u;
    }
    if (ix > 0u) {
        while (j > 0u) {
            j -= 1u;
            uint32_t base = 2u * JL_WG - (2u << (8u - j));
            Bic test = bic_combine(sh_bic[base + (ix >> j) - 1u], *bic);
            if (test.b == 0u) {
                *bic = test;
                ix -= 1u << j;
            }
        }
    }
    if (ix > 0u) return (int32_t)ix - 1;
    return (int32_t)(~0u - bic->a);
}

__global__ __launch_bounds__(JL_WG) void k_clip_leaf(const JlConfig* __restrict__ cfg, Buf<JlClipInp> clip_inp, Buf<JlPathBbox> path_bboxes,
                                                     Buf<JlClipBic> reduced, Buf<JlClipEl> clip_els, Buf<JlDrawMonoid> draw_monoids,
                                                     Buf<Bb4> clip_bboxes) {
    __shared__ Bic sh_bic[510];
    __shared__ uint32_t sh_stack[JL_WG];
    __shared__ Bb4 sh_stack_bbox[JL_WG];
    __shared__ Bb4 sh_bbox[JL_WG];
    __shared__ int32_t sh_link[JL_WG];
    uint32_t lid = threadIdx.x, gid = blockIdx.x * JL_WG + lid;
    Bic bic;
    bic.a = 0u; bic.b = 0u;
    if (lid < blockIdx.x) {
        JlClipBic r = reduced.rd(lid);
        bic.a = r.a; bic.b = r.b;
    }
    bic = reverse_scan_bic(bic, sh_bic);
    __syncthreads();
    uint32_t stack_size = sh_bic[0].b;
    // binary search in stack
    uint32_t sp = JL_WG - 1u - lid;
    uint32_t ix = 0u;
    for (uint32_t i = 0; i < 8u; i++) {
        uint32_t probe = ix + (128u >> i);
        if (sp < sh_bic[probe].b) ix = probe;
    }
    uint32_t b = sh_bic[ix].b;
    Bb4 bbox = bb_inf();
    sh_stack[lid] = 0u;
    if (sp < b) {
        JlClipEl el = clip_els.rd(ix * JL_WG + b - sp - 1u);
        sh_stack[lid] = el.parent_ix;
        bbox.v[0] = el.bbox[0]; bbox.v[1] = el.bbox[1]; bbox.v[2] = el.bbox[2]; bbox.v[3] = el.bbox[3];
    }
    // forward scan of bbox values of prefix stack
    for (uint32_t i = 0; i < 8u; i++) {
        sh_stack_bbox[lid] = bbox;
        __syncthreads();
        if (lid >= (1u << i)) bbox = bbox_intersect(sh_stack_bbox[lid - (1u << i)], bbox);
        __syncthreads();
    }
    sh_stack_bbox[lid] = bbox;

    // Read input and compute Bic binary tree
    int32_t inp = (gid < cfg->layout.n_clip) ? clip_inp.rd(gid).path_ix : (int32_t)0x80000000;
    bool is_push = inp >= 0;
    bic.a = 1u - (is_push ? 1u : 0u);
    bic.b = is_push ? 1u : 0u;
    __syncthreads();  // all reads of sh_bic above are done
    sh_bic[lid] = bic;
    if (is_push) {
        JlPathBbox pb = path_bboxes.rd((uint32_t)inp);
        bbox.v[0] = (float)pb.x0; bbox.v[1] = (float)pb.y0; bbox.v[2] = (float)pb.x1; bbox.v[3] = (float)pb.y1;
    } else {
        bbox = bb_inf();
    }
    uint32_t inbase = 0u;
    for (uint32_t i = 0; i < 7u; i++) {
        uint32_t outbase = 2u * JL_WG - (1u << (8u - i));
        __syncthreads();
        if (lid < (1u << (7u - i))) {
            uint32_t in_off = inbase + lid * 2u;
            sh_bic[outbase + lid] = bic_combine(sh_bic[in_off], sh_bic[in_off + 1u]);
        }
        inbase = outbase;
    }
    __syncthreads();
    // search for predecessor node
    bic.a = 0u; bic.b = 0u;
    int32_t link = search_link(&bic, lid, sh_bic);
    sh_link[lid] = link;
    __syncthreads();
    int32_t grandparent = (link >= 0) ? sh_link[link] : (link - 1);
    int32_t parent;
    if (link >= 0) {
        parent = (int32_t)(blockIdx.x * JL_WG) + link;
    } else if (link + (int32_t)stack_size >= 0) {
        parent = (int32_t)sh_stack[(int32_t)JL_WG + link];
    } else {
        parent = -1;
    }
    // bbox scan (intersect) across parent links
    for (uint32_t i = 0; i < 8u; i++) {
        if (i != 0u) sh_link[lid] = link;
        sh_bbox[lid] = bbox;
        __syncthreads();
        if (link >= 0) {
            bbox = bbox_intersect(sh_bbox[link], bbox);
            link = sh_link[link];
        }
        __syncthreads();
    }
    if (link + (int32_t)stack_size >= 0) bbox = bbox_intersect(sh_stack_bbox[(int32_t)JL_WG + link], bbox);
    sh_bbox[lid] = bbox;
    __syncthreads();
    if (!is_push && gid < cfg->layout.n_clip) {
        if (parent >= 0) {
            JlClipInp parent_clip = clip_inp.rd((uint32_t)parent);
            uint32_t dix = ~(uint32_t)inp;
            if (draw_monoids.ok(dix)) {
                draw_monoids.p[dix].path_ix = (uint32_t)parent_clip.path_ix;
                draw_monoids.p[dix].scene_offset = draw_monoids.rd(parent_clip.ix).scene_offset;
            }
        }
        if (grandparent >= 0) {
            bbox = sh_bbox[grandparent];
        } else if (grandparent + (int32_t)stack_size >= 0) {
            bbox = sh_stack_bbox[(int32_t)JL_WG + grandparent];
        } else {
            bbox = bb_inf();
        }
    }
    if (gid < cfg->layout.n_clip) clip_bboxes.wr(gid, bbox);
}

}  // namespace

// [clip_inp, path_bboxes, reduced(bics), clip_out(els)]
int jh_launch_clip_reduce(const JhLaunch& L) {
    if (L.nb < 4) return -1;
    if (L.gx == 0) return 0;
    hipLaunchKernelGGL(k_clip_reduce, dim3(L.gx), dim3(JL_WG), 0, L.stream, mkbuf<JlClipInp>(L.b[0].ptr, L.b[0].size),
                       mkbuf<JlPathBbox>(L.b[1].ptr, L.b[1].size), mkbuf<JlClipBic>(L.b[2].ptr, L.b[2].size),
                       mkbuf<JlClipEl>(L.b[3].ptr, L.b[3].size));
    return 0;
}
// [config, clip_inp, path_bboxes, reduced, clip_els, draw_monoids, clip_bboxes]
int jh_launch_clip_leaf(const JhLaunch& L) {
    if (L.nb < 7) return -1;
    if (L.gx == 0) return 0;
    hipLaunchKernelGGL(k_clip_leaf, dim3(L.gx), dim3(JL_WG), 0, L.stream, (const JlConfig*)L.b[0].ptr, mkbuf<JlClipInp>(L.b[1].ptr, L.b[1].size),
                       mkbuf<JlPathBbox>(L.b[2].ptr, L.b[2].size), mkbuf<JlClipBic>(L.b[3].ptr, L.b[3].size),
                       mkbuf<JlClipEl>(L.b[4].ptr, L.b[4].size), mkbuf<JlDrawMonoid>(L.b[5].ptr, L.b[5].size),
                       mkbuf<Bb4>(L.b[6].ptr, L.b[6].size));
    return 0;
}
