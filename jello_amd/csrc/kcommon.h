// kcommon.h -- shared device helpers: bounds-checked buffer views (the WGSL robust-access rule:
// out-of-range reads give zero, writes are dropped -- also what keeps a malformed scene from
// faulting the GPU), wave64/LDS block scans, and the launcher declarations used by jello_hip.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/jello_formats.h"
#include "dmath.h"

#define JL_WG 256

namespace jk {

template <typename T>
struct Buf {
    T* p;
    uint32_t n;  // element count
    JD T rd(uint32_t i) const {
        if (i < n) return p[i];
        T z;
        __builtin_memset(&z, 0, sizeof(T));
        return z;
    }
    JD void wr(uint32_t i, const T& v) const {
        if (i < n) p[i] = v;
    }
    JD bool ok(uint32_t i) const { return i < n; }
};

template <typename T>
static inline Buf<T> mkbuf(void* p, uint64_t bytes) {
    Buf<T> b;
    b.p = (T*)p;
    uint64_t n = bytes / sizeof(T);
    b.n = n > 0xffffffffull ? 0xffffffffu : (uint32_t)n;
    return b;
}

// ---- wave64 primitives ----
JD uint32_t lane_id() { return threadIdx.x & 63u; }
JD uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }  // a value that is the same in every lane, as a scalar
// Inclusive wave64 prefix operations on the DPP path (row_shr 1,2,4,8 inside each 16-lane row, then
// row_bcast:15 / row_bcast:31 across rows): six VALU instructions, no LDS crossbar round trips
// (__shfl_up compiles to ds_bpermute_b32, ~100 cycles each and six of them dependent).
#define JK_DPP_ROW_SHR(n) (0x110 + (n))
#define JK_DPP_ROW_BCAST15 0x142
#define JK_DPP_ROW_BCAST31 0x143
JD uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_SHR(8), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_BCAST15, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return v;
}
JD uint32_t wave_incl_max_u32(uint32_t v) {
    v = jd::umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_SHR(1), 0xf, 0xf, false));
    v = jd::umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_SHR(2), 0xf, 0xf, false));
    v = jd::umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_SHR(4), 0xf, 0xf, false));
    v = jd::umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_SHR(8), 0xf, 0xf, false));
    v = jd::umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_BCAST15, 0xa, 0xf, false));
    v = jd::umax_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, JK_DPP_ROW_BCAST31, 0xc, 0xf, false));
    return v;
}
// Orders this wave's LDS accesses (a wave's DS instructions execute in issue order; the fence only has to stop the
// compiler from moving them) -- the synchronisation primitive of kernels whose waves own private LDS regions.
JD void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// [p, p + n) of a bump counter for every ACTIVE lane of the wave, with ONE atomic: the lanes are served value by value (the
// lanes of a call site mostly ask for the same n: one round), a lane's offset = what the rounds before it and the lanes below
// it in its own round take.  By hand, because LLVM's atomic optimizer in its DPP strategy -- which does the same with a wave
// prefix sum -- returned wrong offsets for flatten's call sites (divergent branches inside loops; test_c2_blobs_all_joins_caps_evenodd
// failed with it and passes with the strategies None and Iterative), and None means 64 atomics per call.  The build no longer
// passes that strategy (round 6, csrc/Makefile); jh_selftest_atomics holds whatever the compiler does with this function and
// with the plain per-lane form to a serial sum.
JD uint32_t wave_bump(uint32_t* ctr, uint32_t n) {
    const uint64_t below = (1ull << lane_id()) - 1ull;
    uint64_t todo = __builtin_amdgcn_ballot_w64(true);
    const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
    uint32_t off = 0u, total = 0u;
    while (todo != 0ull) {  // uniform among the active lanes
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)n, __builtin_ctzll(todo));
        const uint64_t m = __builtin_amdgcn_ballot_w64(n == v);
        if (n == v) off = total + v * (uint32_t)__builtin_popcountll(m & below);
        total += v * (uint32_t)__builtin_popcountll(m);
        todo &= ~m;
    }
    uint32_t p = 0u;
    if (lane_id() == leader) p = atomicAdd(ctr, total);
    return (uint32_t)__builtin_amdgcn_readlane((int)p, (int)leader) + off;
}
JD uint32_t wave_reduce_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
}

// Exclusive scan of one u32 per thread across a 256-thread block.  `sh` needs 5 words.
// Returns the exclusive prefix; *total receives the block sum.
JD uint32_t block_excl_scan_u32(uint32_t v, uint32_t* sh, uint32_t* total) {
    uint32_t incl = wave_incl_scan_u32(v);
    uint32_t w = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 63u) sh[w] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        uint32_t s = sh[i];
        if (i < w) base += s;
    }
    *total = sh[0] + sh[1] + sh[2] + sh[3];
    return base + incl - v;
}

// K-word monoid (component-wise u32 add): exclusive block scan.  sh needs 4*K words.
template <int K>
struct MonoidK {
    uint32_t v[K];
};
template <int K>
JD MonoidK<K> monoid_add(const MonoidK<K>& a, const MonoidK<K>& b) {
    MonoidK<K> c;
#pragma unroll
    for (int i = 0; i < K; i++) c.v[i] = a.v[i] + b.v[i];
    return c;
}
template <int K>
JD MonoidK<K> block_excl_scan_monoid(const MonoidK<K>& in, uint32_t* sh, MonoidK<K>* total) {
    MonoidK<K> incl;
#pragma unroll
    for (int i = 0; i < K; i++) incl.v[i] = wave_incl_scan_u32(in.v[i]);
    uint32_t w = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 63u) {
#pragma unroll
        for (int i = 0; i < K; i++) sh[w * K + i] = incl.v[i];
    }
    __syncthreads();
    MonoidK<K> out, tot;
#pragma unroll
    for (int i = 0; i < K; i++) {
        uint32_t base = 0, t = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            uint32_t s = sh[j * K + i];
            if (j < w) base += s;
            t += s;
        }
        out.v[i] = base + incl.v[i] - in.v[i];
        tot.v[i] = t;
    }
    *total = tot;
    return out;
}
template <int K>
JD MonoidK<K> block_reduce_monoid(const MonoidK<K>& in, uint32_t* sh) {
    MonoidK<K> r = in;
#pragma unroll
    for (int i = 0; i < K; i++) r.v[i] = wave_reduce_u32(r.v[i]);
    uint32_t w = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 0u) {
#pragma unroll
        for (int i = 0; i < K; i++) sh[w * K + i] = r.v[i];
    }
    __syncthreads();
    MonoidK<K> t;
#pragma unroll
    for (int i = 0; i < K; i++) t.v[i] = sh[i] + sh[K + i] + sh[2 * K + i] + sh[3 * K + i];
    return t;
}

// shared/pathtag.wgsl:58-71
JD MonoidK<5> reduce_tag(uint32_t tag_word) {
    MonoidK<5> c;
    uint32_t point_count = tag_word & 0x3030303u;
    c.v[1] = __popc((point_count * 7u) & 0x4040404u);          // pathseg_ix
    c.v[0] = __popc(tag_word & (0x20u * 0x1010101u));           // trans_ix
    uint32_t n_points = point_count + ((tag_word >> 2) & 0x1010101u);
    uint32_t a = n_points + (n_points & (((tag_word >> 3) & 0x1010101u) * 15u));
    a += a >> 8;
    a += a >> 16;
    c.v[2] = a & 0xffu;                                          // pathseg_offset
    c.v[4] = __popc(tag_word & (0x10u * 0x1010101u));            // path_ix
    c.v[3] = __popc(tag_word & (0x40u * 0x1010101u)) * 2u;       // style_ix
    return c;
}
JD void store_tm(JlTagMonoid* dst, const MonoidK<5>& m) {
    dst->trans_ix = m.v[0]; dst->pathseg_ix = m.v[1]; dst->pathseg_offset = m.v[2]; dst->style_ix = m.v[3]; dst->path_ix = m.v[4];
}
JD MonoidK<5> load_tm(const Buf<JlTagMonoid>& b, uint32_t i) {
    JlTagMonoid t = b.rd(i);
    MonoidK<5> m;
    m.v[0] = t.trans_ix; m.v[1] = t.pathseg_ix; m.v[2] = t.pathseg_offset; m.v[3] = t.style_ix; m.v[4] = t.path_ix;
    return m;
}
// pathtag_scan.wgsl (small variant): the prefix of workgroup wg = the sum of the wg entries of `parent` in front of it (sh: 20 words)
JD MonoidK<5> parent_prefix(const Buf<JlTagMonoid>& parent, uint32_t wg, uint32_t* sh) {
    MonoidK<5> agg;
#pragma unroll
    for (int i = 0; i < 5; i++) agg.v[i] = 0;
    if (threadIdx.x < wg) agg = load_tm(parent, threadIdx.x);
    return block_reduce_monoid<5>(agg, sh);
}
// shared/drawtag.wgsl:46-53
JD MonoidK<4> map_draw_tag(uint32_t t) {
    MonoidK<4> c;
    c.v[0] = (t != 0u) ? 1u : 0u;
    c.v[1] = t & 1u;
    c.v[2] = (t >> 2) & 7u;
    c.v[3] = (t >> 6) & 0xfu;
    return c;
}

}  // namespace jk

// ------------------------------------------------------------------------------------------------
// Host-side launcher interface (implemented in the kernels_*.hip files, called by jello_hip.cpp)
// ------------------------------------------------------------------------------------------------
struct JhBound {
    void* ptr;
    uint64_t size;  // bytes
    uint32_t width, height;
    int format;
};

// fine's image array: up to JH_FINE_INLINE_IMAGES descriptors travel in the kernel arguments, larger arrays as a device table
#define JH_FINE_INLINE_IMAGES 8
struct JhImageDesc {
    const void* ptr;  // RGBA8 texels, row-major; nullptr = never written (samples as transparent black)
    uint32_t width, height;
    uint32_t srgb;    // 1: JL_RGBA8_SRGB (decode to linear when sampled)
    uint32_t pad;
};

struct JhScratch;  // per-context scratch allocator, defined in jello_hip.cpp
void* jh_scratch_get(JhScratch* s, int slot, uint64_t bytes);  // grows on demand, returns device pointer (nullptr on OOM)

struct JhLaunch {
    hipStream_t stream;
    JhScratch* scratch;
    uint32_t gx, gy, gz;
    const JhBound* b;
    int nb;
    const JhBound* images;  // JH_BIND_IMAGE_ARRAY contents for fine
    int n_images;
    const uint32_t* indirect;  // device pointer to IndirectCount (indirect dispatch) or nullptr
    int num_cus;
    const JlConfig* cfg_host;  // host shadow of the uploaded ConfigUniform bound at index 0, or nullptr
    uint32_t band_row0, band_row1;  // jh_set_band: bin rows [row0, row1) this context writes PTCL for and rasterises (0, ~0u = all)
    const JhImageDesc* image_table;  // device table of all n_images descriptors when n_images > JH_FINE_INLINE_IMAGES, else nullptr
    uint32_t clip_depth_hint;  // jh_set_clip_depth_hint: upper bound of the clip layers' nesting depth, 0 = unknown
    uint32_t* hint_overflow;   // device counter of the blend-stack saves dropped because that hint was too small (or nullptr)
    uint32_t debug_flatten;    // jh_debug_flatten_regions (tests): bit 0 = every wave of k_flatten_items starts in region 0 of the temporary, bit 1 = always 8 regions, bit 2 = batches allocate job by job
    uint32_t absorb;  // JH_ABSORB_*: held-back commands this stage performs in passing (jello_hip.cpp, Deferred)
    JhBound extra;    // JH_ABSORB_SETUP of path_tiling: the ptcl buffer of path_tiling_setup (ptcl[0] = ~0 on failure); JH_ABSORB_PATHTAG_SCAN: see there
};
enum { JH_ABSORB_BBOX_CLEAR = 1u, JH_ABSORB_BUMP_CLEAR = 2u, JH_ABSORB_SETUP = 4u,
       // flatten: the last pathtag scan (pathtag_scan_small / _large) was held back; flatten's classification kernel produces the tag
       // monoids in passing.  L.extra = the scan's `reduced` binding, extra.width = its workgroup count, extra.height = 1: the small variant
       JH_ABSORB_PATHTAG_SCAN = 8u };

enum {  // scratch slots
    JH_SCR_SCAN_TMP = 0,
    JH_SCR_A = 1,
    JH_SCR_B = 2,
    JH_SCR_C = 3,
    JH_SCR_D = 4,
    JH_SCR_E = 5,
    JH_SCR_F = 6,
    JH_SCR_G = 7,
    JH_SCR_H = 8,
    JH_SCR_I = 9,
    JH_SCR_J = 10,
    JH_SCR_FL_CTR = 11,  // flatten's list counters / chunk fills: NOT shared with other stages (they survive between frames)
    JH_SCR_BD_CTR = 12,  // backdrop's wide-row counter: likewise
    JH_SCR_PC_TOT = 13,  // path_count's crossings per path (atomic sums): likewise, zeroed by the stage's last kernel
    JH_SCR_COUNT = 14
};
// Counters a stage needs zeroed when it starts are zeroed by the LAST kernel that runs before without using them (the
// stage's own last kernel of the frame before, or a kernel of the stage in front) instead of by a fill launch of
// ~4.4 us; a host-side flag per counter says whether that has happened since the counter was last used.  A stage that
// finds its flag down (first frame, an aborted frame, a stage run on its own, a scratch reallocation) fills as before.
enum { JH_CLEAN_FL_CTR = 1u, JH_CLEAN_BD_CTR = 2u, JH_CLEAN_SCAN = 8u, JH_CLEAN_PC_TOT = 16u };
uint32_t* jh_scratch_flags(JhScratch* s);
uint64_t jh_scratch_cap(JhScratch* s, int slot);  // bytes the slot holds (>= what was last asked for)

// Generic device-side exclusive scan of u32 (stride in words between consecutive inputs).
// n is read from *n_dev when n_dev != nullptr (clamped to n_max), else n_max.  Writes out[0..n) and
// *total_dev (if non-null).  Three launches, no inter-workgroup spinning.
int jh_scan_u32(const JhLaunch& L, const uint32_t* in, uint32_t in_stride, uint32_t* out, uint32_t n_max, const uint32_t* n_dev,
                uint32_t* total_dev);


int jh_launch_pathtag(const JhLaunch& L, int stage);
int jh_launch_bbox_clear(const JhLaunch& L);
int jh_launch_flatten(const JhLaunch& L);
int jh_launch_draw_reduce(const JhLaunch& L);
int jh_launch_draw_leaf(const JhLaunch& L);
int jh_launch_clip_reduce(const JhLaunch& L);
int jh_launch_clip_leaf(const JhLaunch& L);
int jh_launch_binning(const JhLaunch& L);
int jh_launch_tile_alloc(const JhLaunch& L);
int jh_launch_path_count_setup(const JhLaunch& L);
int jh_launch_path_count(const JhLaunch& L);
int jh_launch_backdrop_dyn(const JhLaunch& L);
int jh_launch_coarse(const JhLaunch& L);
int jh_launch_path_tiling_setup(const JhLaunch& L);
int jh_launch_path_tiling(const JhLaunch& L);
int jh_launch_fine_area(const JhLaunch& L);
int jh_launch_fine_msaa(const JhLaunch& L, int samples);  // 8 or 16
