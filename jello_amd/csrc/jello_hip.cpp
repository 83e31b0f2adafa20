// jello_hip.cpp -- implementation of the C ABI in include/jello_hip.h: context, pooled device
// buffers keyed by the recording's ResourceIDs, images in linear device memory, per-stage dispatch
// onto the HIP kernels, hipEvent profiling.  Replaces engine/wgpu_engine (wgpu.go:322-643) below
// the renderer.Recording boundary.  There is no CPU fallback here: if HIP is unavailable every
// call fails with JH_ERR_NO_DEVICE / JH_ERR_DEVICE.
#include "../../include/jello_hip.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "kcommon.h"

#ifndef JH_SCR_SKEW
#define JH_SCR_SKEW 1280u
#endif
struct JhScratch {
    void* ptr[JH_SCR_COUNT];
    void* base[JH_SCR_COUNT];  // what hipMalloc returned (ptr may be offset into it)
    uint64_t cap[JH_SCR_COUNT];
    std::vector<void*> retired;  // old allocations kept until the next sync (kernels may still use them)
    uint32_t clean_flags;        // JH_CLEAN_*: see kcommon.h
    jh_ctx* ctx;
};

struct Alloc {
    void* ptr = nullptr;
    uint64_t size = 0;      // logical size (bytes)
    uint64_t capacity = 0;  // allocation size class
    bool owned = true;
    uint32_t width = 0, height = 0;
    int format = 0;
    bool pending_clear = false;
    bool written = false;   // images: has content (uploaded / imported); a created-only image reads as zero like a new wgpu texture
};

// One node of the profile tree (engine/wgpu_engine/profiler.go:96-158): a group (Start/Nest ... End: CPU interval and
// children) or a GPU query (ProfilerGroup.Compute: one hipEvent pair around a dispatch).
struct ProfEntry {
    int kind;    // JH_PROF_GROUP / JH_PROF_QUERY
    int parent;  // index of the enclosing group, -1 at top level
    int stage;   // query: jh_stage
    std::string label;
    double cpu_start_ms = 0.0, cpu_end_ms = 0.0;
    hipEvent_t start = nullptr, stop = nullptr;  // query only
};

// A recorded command that has not been launched yet.  Five of the recording's commands are 1-thread or trivially small
// kernels in front of a stage that can do their work in passing (the last pathtag scan, bbox_clear and Clear(bump) in front of flatten,
// path_count_setup / path_tiling_setup in front of their indirect dispatches, pathtag_reduce2 in front of pathtag_scan1;
// render.go:186-197,230-237,369-374,415-420): the
// engine holds them back until the next command and lets that stage absorb them when it is the one they were waiting for
// (same buffers) -- or launches them as recorded before anything else happens (any other command, a download, a sync, the
// end of a graph capture).  What every buffer holds after each command is what the recording says.
struct Deferred {
    bool is_clear = false;
    int stage = -1;
    uint32_t gx = 0, gy = 0, gz = 0;
    std::vector<JhBound> b;
    void* clear_ptr = nullptr;
    uint64_t clear_bytes = 0;
};

struct JhGraph {  // a captured frame plus the resource generation it was captured against
    hipGraphExec_t exec;
    uint64_t generation;
    uint32_t kernel_nodes, other_nodes;  // what the capture recorded (jh_graph_node_counts)
    // JH_CLEAN_* flags that were up when the capture began: the graph holds NO fill for those counters and relies on the
    // frame before it having left them zeroed.  jh_graph_launch zeroes whichever of them is not clean at that moment (a frame
    // that failed half-way, jh_debug_poison_scratch) before the replay.
    uint32_t assumes_clean;
};

struct Staging {  // pinned host arena for uploads: the caller's slice is copied once, the DMA runs asynchronously
    char* base = nullptr;
    uint64_t cap = 0, used = 0;
};

struct jh_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    std::string name;
    uint64_t total_mem = 0;
    std::unordered_map<uint64_t, Alloc> buffers;
    std::unordered_map<uint64_t, Alloc> images;
    std::unordered_map<uint64_t, JlConfig> config_shadow;  // host copies of uploaded ConfigUniforms (by buffer id)
    std::multimap<uint64_t, void*> pool;  // capacity -> free allocation
    uint64_t pool_bytes = 0;
    JhScratch scratch;
    bool profiling = false;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> free_events;
    std::vector<int> prof_stack;  // open groups (indices into prof)
    std::string last_error;
    uint32_t band_row0 = 0u, band_row1 = 0xffffffffu;  // jh_set_band
    uint32_t clip_depth_hint = 0u;                     // jh_set_clip_depth_hint
    uint32_t debug_flatten = 0u;                       // jh_debug_flatten_regions
    // Bumped whenever a device pointer a captured graph may have baked in goes away or moves: buffer / image free,
    // regrow or import, scratch regrow.  jh_graph_launch refuses a graph captured against an older generation.
    uint64_t generation = 0;
    bool capturing = false;
    uint32_t clean_at_capture_begin = 0u;
    uint64_t graph_self_cleans = 0;  // replays that had to zero a counter first (jh_debug_graph_self_cleans)
    Staging staging;
    // fine: descriptor table of the bound image array when it has more entries than fit in the kernel arguments
    void* image_table = nullptr;
    uint64_t image_table_cap = 0;
    uint32_t* hint_overflow = nullptr;  // device word: blend-stack saves fine had to drop because the clip-depth hint was too small
    std::vector<JhImageDesc> image_table_host;
    std::vector<Deferred> deferred;  // held-back commands, in recording order
};

static int flush_deferred(jh_ctx* ctx);
#define JH_FLUSH(ctx)                              \
    do {                                           \
        int frc__ = flush_deferred(ctx);           \
        if (frc__ != JH_OK) return frc__;          \
    } while (0)

static int fail(jh_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    return code;
}
static int hip_fail(jh_ctx* ctx, hipError_t e, const char* what) {
    return fail(ctx, JH_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(ctx, expr)                                \
    do {                                                  \
        hipError_t e__ = (expr);                          \
        if (e__ != hipSuccess) return hip_fail(ctx, e__, #expr); \
    } while (0)

// engine/wgpu_engine/wgpu.go:800-808 poolSizeClass with sizeClassBits = 1
static uint64_t pool_size_class(uint64_t x) {
    const uint32_t num_bits = 1;
    if (x > (1ull << num_bits)) {
        int a = __builtin_clzll(x - 1);
        uint64_t b = (x - 1) | (((~0ull / 2) >> num_bits) >> a);
        return b + 1;
    }
    return 1ull << num_bits;
}

static int pool_get(jh_ctx* ctx, uint64_t size, void** out, uint64_t* cap_out) {
    uint64_t cap = pool_size_class(size < 16 ? 16 : size);
    auto it = ctx->pool.find(cap);
    if (it != ctx->pool.end()) {
        *out = it->second;
        *cap_out = cap;
        ctx->pool.erase(it);
        return 0;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) return fail(ctx, JH_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(e));
    ctx->pool_bytes += cap;
    *out = p;
    *cap_out = cap;
    return 0;
}

void* jh_scratch_get(JhScratch* s, int slot, uint64_t bytes) {
    if (slot < 0 || slot >= JH_SCR_COUNT) return nullptr;
    if (bytes < 256) bytes = 256;
    if (s->cap[slot] >= bytes) return s->ptr[slot];
    // A capture records pointers: growing an array now would leave the graph with a mix of old and new ones, and
    // jh_graph_end would put the "left clean" flag of the NEW, uninitialised memory back up.  The contract --
    // run the recording once eagerly before capturing it -- makes every array big enough; a capture that still has to grow
    // one is refused (the dispatch answers JH_ERR_OOM with a message that says so).
    if (s->ctx->capturing) return nullptr;
    // (what is asked for + 1/16, in whole MiB: these arrays are internal, they do not go through the reference's pool and its
    // size classes -- 2^k and 1.5 * 2^k, up to a third more than asked -- and the sixteenth keeps a slowly growing scene from
    // reallocating every frame)
    uint64_t cap = bytes + bytes / 16;
    cap = cap < (1ull << 20) ? pool_size_class(cap) : (cap + ((1ull << 20) - 1)) & ~((1ull << 20) - 1);
    void* p = nullptr;
    // Every slot starts at its own offset (a multiple of 256 B) into its allocation: the flatten kernels walk up to
    // seven scratch arrays with the same index at once, and with all of them on 2 MiB boundaries k_flatten_lines
    // measured 117-124 us on C3 (4 of 4 processes) against 106-109 us (6 of 8) with the offsets.
    if (hipMalloc(&p, cap + (uint64_t)JH_SCR_COUNT * JH_SCR_SKEW) != hipSuccess) return nullptr;
    if (s->base[slot]) {
        s->retired.push_back(s->base[slot]);
        s->ctx->generation++;  // a captured graph may hold the old pointer
    }
    s->base[slot] = p;
    // new memory holds anything: the slot's "left clean by its kernels" flag goes down (the other slots keep theirs -- with one
    // flag word for all, a frame that grew ANY array late made the next frame fill every counter again, and a graph captured
    // from that frame kept the fills)
    switch (slot) {
        case JH_SCR_SCAN_TMP: s->clean_flags &= ~(uint32_t)JH_CLEAN_SCAN; break;
        case JH_SCR_FL_CTR: s->clean_flags &= ~(uint32_t)JH_CLEAN_FL_CTR; break;
        case JH_SCR_BD_CTR: s->clean_flags &= ~(uint32_t)JH_CLEAN_BD_CTR; break;
        case JH_SCR_PC_TOT: s->clean_flags &= ~(uint32_t)JH_CLEAN_PC_TOT; break;
        default: break;  // (the other arrays are written before they are read in every frame)
    }
    p = (char*)p + (uint64_t)slot * JH_SCR_SKEW;
    s->ptr[slot] = p;
    s->cap[slot] = cap;
    return p;
}

uint32_t* jh_scratch_flags(JhScratch* s) { return &s->clean_flags; }
uint64_t jh_scratch_cap(JhScratch* s, int slot) { return (slot >= 0 && slot < JH_SCR_COUNT) ? s->cap[slot] : 0; }

static void scratch_release_retired(JhScratch* s) {
    for (void* p : s->retired) (void)hipFree(p);
    s->retired.clear();
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Copies `size` host bytes into the pinned arena and returns the pinned address (valid until the arena wraps, which
// first waits for the stream).  nullptr on allocation failure.
static void* stage_copy(jh_ctx* ctx, const void* data, uint64_t size) {
    Staging& st = ctx->staging;
    uint64_t need = (size + 255u) & ~255ull;
    if (st.used + need > st.cap) {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;  // every earlier copy out of the arena is done
        st.used = 0;
        if (need > st.cap) {
            uint64_t cap = st.cap ? st.cap : (8ull << 20);
            while (cap < need) cap *= 2;
            if (st.base) (void)hipHostFree(st.base);
            st.base = nullptr;
            st.cap = 0;
            void* p = nullptr;
            if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
            st.base = (char*)p;
            st.cap = cap;
        }
    }
    void* dst = st.base + st.used;
    std::memcpy(dst, data, size);
    st.used += need;
    return dst;
}

// Per-stage binding contract (SURVEY Appendix C = renderer/render.go dispatch order): minimum binding count and the
// slots whose contents the launchers dereference as fixed structs -- checked before every launch so that a buffer that
// is too small yields JH_ERR_INVALID instead of an out-of-bounds read.
struct StageContract { int n_min, cfg, bump, indirect; };
static const StageContract kContract[JH_STAGE_COUNT] = {
    /* pathtag_reduce */ {3, 0, -1, -1}, /* pathtag_reduce2 */ {2, -1, -1, -1}, /* pathtag_scan1 */ {3, -1, -1, -1},
    /* pathtag_scan_small */ {4, 0, -1, -1}, /* pathtag_scan_large */ {4, 0, -1, -1}, /* bbox_clear */ {2, 0, -1, -1},
    /* flatten */ {6, 0, 4, -1}, /* draw_reduce */ {3, 0, -1, -1}, /* draw_leaf */ {7, 0, -1, -1}, /* clip_reduce */ {4, -1, -1, -1},
    /* clip_leaf */ {7, 0, -1, -1}, /* binning */ {8, 0, 5, -1}, /* tile_alloc */ {6, 0, 3, -1}, /* backdrop_dyn */ {4, 0, 1, -1},
    /* path_count_setup */ {2, -1, 0, 1}, /* path_count */ {6, 0, 1, -1}, /* coarse */ {9, 0, 7, -1},
    /* path_tiling_setup */ {3, -1, 0, 1}, /* path_tiling */ {6, -1, 0, -1}, /* fine_area */ {7, 0, -1, -1},
    /* fine_msaa8 */ {9, 0, -1, -1}, /* fine_msaa16 */ {9, 0, -1, -1}};

extern "C" {

const char* jh_stage_name(int stage) {
    static const char* names[JH_STAGE_COUNT] = {
        "pathtag_reduce", "pathtag_reduce2", "pathtag_scan1", "pathtag_scan_small", "pathtag_scan_large", "bbox_clear", "flatten",
        "draw_reduce", "draw_leaf", "clip_reduce", "clip_leaf", "binning", "tile_alloc", "backdrop_dyn", "path_count_setup", "path_count",
        "coarse", "path_tiling_setup", "path_tiling", "fine_area", "fine_msaa8", "fine_msaa16"};
    return (stage >= 0 && stage < JH_STAGE_COUNT) ? names[stage] : "?";
}

int jh_create(jh_ctx** out, int device) {
    if (!out) return JH_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return JH_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return JH_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return JH_ERR_DEVICE;
    jh_ctx* ctx = new jh_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->num_cus = prop.multiProcessorCount;
        ctx->name = prop.name;
        ctx->total_mem = prop.totalGlobalMem;
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return JH_ERR_DEVICE;
    }
    ctx->stream = ctx->own_stream;
    std::memset(&ctx->scratch.ptr, 0, sizeof ctx->scratch.ptr);
    std::memset(&ctx->scratch.cap, 0, sizeof ctx->scratch.cap);
    std::memset(&ctx->scratch.base, 0, sizeof ctx->scratch.base);
    ctx->scratch.clean_flags = 0u;
    ctx->scratch.ctx = ctx;
    // (zeroed on the context's OWN stream: a hipMemset here would be the process's first use of the legacy default stream, and
    // from then on the frames of two contexts no longer overlapped at all -- bench.py's two frames in flight fell from 0.91 to
    // 1.05 ms per frame, found by bisection in round 5)
#ifndef JH_NO_HINT_COUNTER
    if (hipMalloc((void**)&ctx->hint_overflow, 256) != hipSuccess || hipMemsetAsync(ctx->hint_overflow, 0, 256, ctx->own_stream) != hipSuccess ||
        hipStreamSynchronize(ctx->own_stream) != hipSuccess)
        ctx->hint_overflow = nullptr;
#endif
    *out = ctx;
    return JH_OK;
}

void jh_destroy(jh_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->buffers)
        if (kv.second.owned && kv.second.ptr) (void)hipFree(kv.second.ptr);
    for (auto& kv : ctx->images)
        if (kv.second.owned && kv.second.ptr) (void)hipFree(kv.second.ptr);
    for (auto& kv : ctx->pool) (void)hipFree(kv.second);
    for (int i = 0; i < JH_SCR_COUNT; i++)
        if (ctx->scratch.base[i]) (void)hipFree(ctx->scratch.base[i]);
    scratch_release_retired(&ctx->scratch);
    if (ctx->staging.base) (void)hipHostFree(ctx->staging.base);
    if (ctx->image_table) (void)hipFree(ctx->image_table);
    if (ctx->hint_overflow) (void)hipFree(ctx->hint_overflow);
    for (auto& p : ctx->prof)
        if (p.kind == JH_PROF_QUERY) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
    for (auto& e : ctx->free_events) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* jh_last_error(jh_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int jh_set_stream(jh_ctx* ctx, void* hip_stream) {
    if (!ctx) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (next != ctx->stream && ctx->staging.used) {  // uploads still in flight on the old stream read the pinned arena
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->staging.used = 0;
    }
    ctx->stream = next;
    return JH_OK;
}

// A stream of the lowest (level < 0), the default (0) or the highest (> 0) launch priority of the context's device, for callers that
// want a stage on a stream of its own (tools/fine_priority.py).  Destroy with jh_stream_destroy.
int jh_stream_create(jh_ctx* ctx, int level, void** hip_stream) {
    if (!ctx || !hip_stream) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int least = 0, greatest = 0;
    HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t st = nullptr;
    HIP_TRY(ctx, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, level < 0 ? least : (level > 0 ? greatest : 0)));
    *hip_stream = (void*)st;
    return JH_OK;
}
int jh_stream_destroy(jh_ctx* ctx, void* hip_stream) {
    if (!ctx || !hip_stream) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamDestroy((hipStream_t)hip_stream));
    return JH_OK;
}

int jh_set_band(jh_ctx* ctx, uint32_t bin_row0, uint32_t bin_row1) {
    if (!ctx || bin_row1 < bin_row0) return JH_ERR_INVALID;
    ctx->band_row0 = bin_row0;
    ctx->band_row1 = bin_row1;
    return JH_OK;
}

int jh_sync(jh_ctx* ctx) {
    if (!ctx) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    scratch_release_retired(&ctx->scratch);
    ctx->staging.used = 0;
    return JH_OK;
}

// ---- buffers ----
static int buffer_get_or_create(jh_ctx* ctx, uint64_t id, uint64_t size, Alloc** out) {
    auto it = ctx->buffers.find(id);
    if (it != ctx->buffers.end()) {
        if (it->second.size >= size) {
            *out = &it->second;
            return 0;
        }
        if (!it->second.owned) return fail(ctx, JH_ERR_INVALID, "imported buffer is smaller than the requested size (the caller owns it: cannot grow)");
        // grow: return the old allocation to the pool (a held-back command may still refer to it: launch those first)
        JH_FLUSH(ctx);
        ctx->pool.insert({it->second.capacity, it->second.ptr});
        ctx->buffers.erase(it);
        ctx->generation++;
    }
    Alloc a;
    int rc = pool_get(ctx, size, &a.ptr, &a.capacity);
    if (rc) return rc;
    a.size = size;
    a.owned = true;
    auto ins = ctx->buffers.emplace(id, a);
    *out = &ins.first->second;
    return 0;
}

int jh_buffer_create(jh_ctx* ctx, uint64_t id, uint64_t size) {
    if (!ctx) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Alloc* a;
    return buffer_get_or_create(ctx, id, size, &a);
}

int jh_buffer_import(jh_ctx* ctx, uint64_t id, void* device_ptr, uint64_t size) {
    if (!ctx || !device_ptr) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    auto it = ctx->buffers.find(id);
    if (it != ctx->buffers.end() && it->second.owned) {
        ctx->pool.insert({it->second.capacity, it->second.ptr});
        ctx->generation++;  // pool memory a graph may point to goes back into circulation
    }
    // (Re-importing caller-owned memory does not invalidate graphs: a graph captured while another pointer was bound keeps
    // using that pointer, which stays valid for as long as its owner keeps it -- e.g. double-buffered output images.)
    Alloc a;
    a.ptr = device_ptr;
    a.size = size;
    a.capacity = size;
    a.owned = false;
    ctx->buffers[id] = a;
    ctx->config_shadow.erase(id);  // whatever was uploaded under this id is not what the new memory holds
    return JH_OK;
}

int jh_upload(jh_ctx* ctx, uint64_t id, const void* data, uint64_t size) {
    if (!ctx || (!data && size)) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Alloc* a;
    int rc = buffer_get_or_create(ctx, id, size, &a);
    if (rc) return rc;
    // The launchers pick kernel instantiations and grids from the host shadow of the ConfigUniform (clip / no-clip fine and
    // coarse, ...), so a captured frame is only valid for the uniform it was captured with: a different uniform under the
    // same id makes every captured graph stale (jh_graph_launch then answers JH_ERR_INVALID instead of replaying kernels
    // that would stop at the first BEGIN_CLIP).
    {
        auto sh = ctx->config_shadow.find(id);
        const bool had = sh != ctx->config_shadow.end();
        if (size == sizeof(JlConfig)) {
            if (!had || std::memcmp(&sh->second, data, sizeof(JlConfig)) != 0) {
                if (had) ctx->generation++;
                std::memcpy(&ctx->config_shadow[id], data, sizeof(JlConfig));
            }
        } else if (had) {
            ctx->config_shadow.erase(sh);
            ctx->generation++;
        }
    }
    if (size) {
        // The host slice is only valid for the duration of the call (reference: queue.WriteBuffer copies, wgpu.go:360):
        // it is copied into the pinned arena here and the DMA is left in flight -- no stream synchronisation per upload.
        void* src = stage_copy(ctx, data, size);
        if (!src) return fail(ctx, JH_ERR_OOM, "jh_upload: pinned staging allocation failed");
        HIP_TRY(ctx, hipMemcpyAsync(a->ptr, src, size, hipMemcpyHostToDevice, ctx->stream));
    }
    return JH_OK;
}

int jh_clear(jh_ctx* ctx, uint64_t id, uint64_t offset, int64_t size) {
    if (!ctx) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto it = ctx->buffers.find(id);
    if (it == ctx->buffers.end()) return fail(ctx, JH_ERR_INVALID, "jh_clear: unknown buffer id (create it first)");
    Alloc& a = it->second;
    if (offset > a.size) return fail(ctx, JH_ERR_INVALID, "jh_clear: offset out of range");
    uint64_t n = size < 0 ? a.size - offset : (uint64_t)size;
    if (offset + n > a.size) n = a.size - offset;
    const bool bump_like = !ctx->profiling && n && offset == 0 && n == a.size && a.size == sizeof(JlBump);  // (profiling: as recorded)
    for (const Deferred& d : ctx->deferred)
        if (!(bump_like && !d.is_clear && (d.stage == JH_BBOX_CLEAR || d.stage == JH_PATHTAG_SCAN_SMALL || d.stage == JH_PATHTAG_SCAN_LARGE))) { JH_FLUSH(ctx); break; }
    if (bump_like) {
        // (the recording's Clear(bump) in front of flatten, render.go:237: flatten's first kernel does it in passing)
        Deferred d;
        d.is_clear = true; d.clear_ptr = a.ptr; d.clear_bytes = n;
        ctx->deferred.push_back(d);
    } else if (n) {
        HIP_TRY(ctx, hipMemsetAsync((char*)a.ptr + offset, 0, n, ctx->stream));
    }
    if (n && offset < sizeof(JlConfig) && ctx->config_shadow.erase(id)) ctx->generation++;  // the shadow no longer describes the device copy
    return JH_OK;
}

int jh_download(jh_ctx* ctx, uint64_t id, void* dst, uint64_t offset, uint64_t size) {
    if (!ctx || !dst) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto it = ctx->buffers.find(id);
    if (it == ctx->buffers.end()) return fail(ctx, JH_ERR_INVALID, "jh_download: unknown buffer id");
    if (offset + size > it->second.size) return fail(ctx, JH_ERR_INVALID, "jh_download: range out of bounds");
    HIP_TRY(ctx, hipMemcpyAsync(dst, (char*)it->second.ptr + offset, size, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return JH_OK;
}

int jh_free(jh_ctx* ctx, uint64_t id) {
    if (!ctx) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    auto it = ctx->buffers.find(id);
    if (it == ctx->buffers.end()) return JH_OK;  // the reference ignores frees of unknown ids (wgpu.go:601-603)
    if (it->second.owned) {
        ctx->pool.insert({it->second.capacity, it->second.ptr});
        ctx->generation++;
    }
    ctx->buffers.erase(it);
    ctx->config_shadow.erase(id);
    return JH_OK;
}

void* jh_buffer_device_ptr(jh_ctx* ctx, uint64_t id) {
    if (!ctx) return nullptr;
    auto it = ctx->buffers.find(id);
    return it == ctx->buffers.end() ? nullptr : it->second.ptr;
}
uint64_t jh_buffer_size(jh_ctx* ctx, uint64_t id) {
    if (!ctx) return 0;
    auto it = ctx->buffers.find(id);
    return it == ctx->buffers.end() ? 0 : it->second.size;
}

// ---- images ----
static uint64_t format_bpp(int format) {
    switch (format) {
        case JL_RGBA8: case JL_RGBA8_SRGB: case JL_BGRA8: return 4;
        case JL_RGBA16_FLOAT: return 8;
        default: return 0;
    }
}

int jh_image_create(jh_ctx* ctx, uint64_t id, uint32_t width, uint32_t height, int format) {
    if (!ctx) return JH_ERR_INVALID;
    uint64_t bpp = format_bpp(format);
    if (!bpp) return fail(ctx, JH_ERR_INVALID, "jh_image_create: bad format");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint64_t size = (uint64_t)width * height * bpp;
    auto it = ctx->images.find(id);
    if (it != ctx->images.end()) {
        if (it->second.width == width && it->second.height == height && it->second.format == format) return JH_OK;
        JH_FLUSH(ctx);
        if (it->second.owned) ctx->pool.insert({it->second.capacity, it->second.ptr});
        ctx->images.erase(it);
        ctx->generation++;
    }
    Alloc a;
    int rc = pool_get(ctx, size ? size : 8, &a.ptr, &a.capacity);
    if (rc) return rc;
    a.size = size; a.width = width; a.height = height; a.format = format; a.owned = true;
    ctx->images[id] = a;
    return JH_OK;
}

int jh_image_import(jh_ctx* ctx, uint64_t id, void* device_ptr, uint32_t width, uint32_t height, int format) {
    if (!ctx || !device_ptr) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    uint64_t bpp = format_bpp(format);
    if (!bpp) return fail(ctx, JH_ERR_INVALID, "jh_image_import: bad format");
    auto it = ctx->images.find(id);
    if (it != ctx->images.end() && it->second.owned) {
        ctx->pool.insert({it->second.capacity, it->second.ptr});
        ctx->generation++;
    }
    Alloc a;
    a.ptr = device_ptr; a.size = (uint64_t)width * height * bpp; a.capacity = a.size; a.owned = false;
    a.width = width; a.height = height; a.format = format; a.written = true;
    ctx->images[id] = a;
    return JH_OK;
}

int jh_image_upload(jh_ctx* ctx, uint64_t id, uint32_t width, uint32_t height, int format, const void* data, uint64_t size) {
    int rc = jh_image_create(ctx, id, width, height, format);
    if (rc) return rc;
    JH_FLUSH(ctx);
    if (!data && size) return JH_ERR_INVALID;
    Alloc& a = ctx->images[id];
    if (size > a.size) size = a.size;
    if (size) {
        void* src = stage_copy(ctx, data, size);
        if (!src) return fail(ctx, JH_ERR_OOM, "jh_image_upload: pinned staging allocation failed");
        HIP_TRY(ctx, hipMemcpyAsync(a.ptr, src, size, hipMemcpyHostToDevice, ctx->stream));
    }
    // an all-zero image (the 1x1 placeholder of render.go:115-124) contributes zero texels: bind it as absent
    bool nonzero = false;
    const uint8_t* bytes = (const uint8_t*)data;
    for (uint64_t i = 0; i < size && !nonzero; i++) nonzero = bytes[i] != 0;
    if (a.written != nonzero) ctx->generation++;  // fine binds a never-written image as absent: the choice is baked into a graph
    a.written = nonzero;
    return JH_OK;
}

// WriteImage (renderer/recording.go:204-208; wgpu.go:422-452 queue.WriteTexture): rows of `width` texels, tightly
// packed in `data`, into the rectangle (x, y, width, height) of the image.
int jh_image_write(jh_ctx* ctx, uint64_t id, uint32_t x, uint32_t y, uint32_t width, uint32_t height, const void* data, uint64_t size) {
    if (!ctx || (!data && size)) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto it = ctx->images.find(id);
    if (it == ctx->images.end()) return fail(ctx, JH_ERR_INVALID, "jh_image_write: unknown image id (create it first)");
    Alloc& a = it->second;
    const uint64_t bpp = format_bpp(a.format);
    if ((uint64_t)x + width > a.width || (uint64_t)y + height > a.height) return fail(ctx, JH_ERR_INVALID, "jh_image_write: rectangle outside the image");
    const uint64_t row = (uint64_t)width * bpp;
    if (size < row * height) return fail(ctx, JH_ERR_INVALID, "jh_image_write: data smaller than the rectangle");
    if (row == 0 || height == 0) return JH_OK;
    void* src = stage_copy(ctx, data, row * height);
    if (!src) return fail(ctx, JH_ERR_OOM, "jh_image_write: pinned staging allocation failed");
    HIP_TRY(ctx, hipMemcpy2DAsync((char*)a.ptr + ((uint64_t)y * a.width + x) * bpp, (uint64_t)a.width * bpp, src, row, row, height,
                                  hipMemcpyHostToDevice, ctx->stream));
    if (!a.written) { a.written = true; ctx->generation++; }
    return JH_OK;
}

int jh_image_download(jh_ctx* ctx, uint64_t id, void* dst, uint64_t size) {
    if (!ctx || !dst) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto it = ctx->images.find(id);
    if (it == ctx->images.end()) return fail(ctx, JH_ERR_INVALID, "jh_image_download: unknown image id");
    if (size > it->second.size) size = it->second.size;
    HIP_TRY(ctx, hipMemcpyAsync(dst, it->second.ptr, size, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return JH_OK;
}

int jh_image_free(jh_ctx* ctx, uint64_t id) {
    if (!ctx) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    auto it = ctx->images.find(id);
    if (it == ctx->images.end()) return JH_OK;
    if (it->second.owned) {
        ctx->pool.insert({it->second.capacity, it->second.ptr});
        ctx->generation++;
    }
    ctx->images.erase(it);
    return JH_OK;
}

void* jh_image_device_ptr(jh_ctx* ctx, uint64_t id) {
    if (!ctx) return nullptr;
    auto it = ctx->images.find(id);
    return it == ctx->images.end() ? nullptr : it->second.ptr;
}

// ---- dispatch ----
static int resolve_bindings(jh_ctx* ctx, const jh_binding* bindings, int n, std::vector<JhBound>& out, std::vector<JhBound>& images) {
    out.clear();
    images.clear();
    for (int i = 0; i < n; i++) {
        const jh_binding& b = bindings[i];
        JhBound r;
        std::memset(&r, 0, sizeof r);
        if (b.kind == JH_BIND_BUFFER) {
            auto it = ctx->buffers.find(b.id);
            if (it == ctx->buffers.end()) return fail(ctx, JH_ERR_INVALID, "dispatch: unknown buffer id in binding " + std::to_string(i));
            r.ptr = it->second.ptr;
            r.size = it->second.size;
            out.push_back(r);
        } else if (b.kind == JH_BIND_IMAGE) {
            auto it = ctx->images.find(b.id);
            if (it == ctx->images.end()) return fail(ctx, JH_ERR_INVALID, "dispatch: unknown image id in binding " + std::to_string(i));
            r.ptr = it->second.ptr; r.size = it->second.size; r.width = it->second.width; r.height = it->second.height; r.format = it->second.format;
            out.push_back(r);
        } else if (b.kind == JH_BIND_IMAGE_ARRAY) {
            for (uint32_t k = 0; k < b.count; k++) {
                auto it = ctx->images.find(b.ids[k]);
                if (it == ctx->images.end()) return fail(ctx, JH_ERR_INVALID, "dispatch: unknown image id in image array");
                JhBound im;
                std::memset(&im, 0, sizeof im);
                // a never-written image samples as transparent black: no pointer, the texel fetch returns zero
                im.ptr = it->second.written ? it->second.ptr : nullptr; im.size = it->second.size; im.width = it->second.width; im.height = it->second.height; im.format = it->second.format;
                images.push_back(im);
            }
            out.push_back(r);  // placeholder keeps binding indices aligned with the WGSL @binding order
        } else {
            return fail(ctx, JH_ERR_INVALID, "dispatch: bad binding kind");
        }
    }
    return 0;
}

static int launch_stage(int stage, const JhLaunch& L) {
    switch (stage) {
        case JH_PATHTAG_REDUCE: case JH_PATHTAG_REDUCE2: case JH_PATHTAG_SCAN1: case JH_PATHTAG_SCAN_SMALL: case JH_PATHTAG_SCAN_LARGE:
            return jh_launch_pathtag(L, stage);
        case JH_BBOX_CLEAR: return jh_launch_bbox_clear(L);
        case JH_FLATTEN: return jh_launch_flatten(L);
        case JH_DRAW_REDUCE: return jh_launch_draw_reduce(L);
        case JH_DRAW_LEAF: return jh_launch_draw_leaf(L);
        case JH_CLIP_REDUCE: return jh_launch_clip_reduce(L);
        case JH_CLIP_LEAF: return jh_launch_clip_leaf(L);
        case JH_BINNING: return jh_launch_binning(L);
        case JH_TILE_ALLOC: return jh_launch_tile_alloc(L);
        case JH_BACKDROP_DYN: return jh_launch_backdrop_dyn(L);
        case JH_PATH_COUNT_SETUP: return jh_launch_path_count_setup(L);
        case JH_PATH_COUNT: return jh_launch_path_count(L);
        case JH_COARSE: return jh_launch_coarse(L);
        case JH_PATH_TILING_SETUP: return jh_launch_path_tiling_setup(L);
        case JH_PATH_TILING: return jh_launch_path_tiling(L);
        case JH_FINE_AREA: return jh_launch_fine_area(L);
        case JH_FINE_MSAA8: return jh_launch_fine_msaa(L, 8);
        case JH_FINE_MSAA16: return jh_launch_fine_msaa(L, 16);
        default: return -9;
    }
}

// Launches every held-back command as recorded (see Deferred).
static int flush_deferred(jh_ctx* ctx) {
    if (ctx->deferred.empty()) return JH_OK;
    std::vector<Deferred> pending;
    pending.swap(ctx->deferred);
    for (const Deferred& d : pending) {
        if (d.is_clear) {
            HIP_TRY(ctx, hipMemsetAsync(d.clear_ptr, 0, d.clear_bytes, ctx->stream));
            continue;
        }
        JhLaunch L;
        std::memset(&L, 0, sizeof L);
        L.stream = ctx->stream;
        L.scratch = &ctx->scratch;
        L.gx = d.gx; L.gy = d.gy; L.gz = d.gz;
        L.b = d.b.data();
        L.nb = (int)d.b.size();
        L.num_cus = ctx->num_cus;
        L.band_row0 = ctx->band_row0;
        L.band_row1 = ctx->band_row1;
        int rc = launch_stage(d.stage, L);
        if (rc) return fail(ctx, JH_ERR_INVALID, std::string("deferred stage failed: ") + jh_stage_name(d.stage));
    }
    return JH_OK;
}

static int dispatch_common(jh_ctx* ctx, int stage, uint32_t gx, uint32_t gy, uint32_t gz, const uint32_t* indirect, const jh_binding* bindings,
                           int n_bindings) {
    if (!ctx || stage < 0 || stage >= JH_STAGE_COUNT || (n_bindings && !bindings)) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<JhBound> b, images;
    int rc = resolve_bindings(ctx, bindings, n_bindings, b, images);
    if (rc) return rc;
    {   // the launchers read these slots as fixed structs: refuse buffers that cannot hold them
        const StageContract& sc = kContract[stage];
        if ((int)b.size() < sc.n_min) return fail(ctx, JH_ERR_INVALID, std::string("bad bindings for stage ") + jh_stage_name(stage) + ": too few");
        auto too_small = [&](int slot, uint64_t need) { return slot >= 0 && (b[slot].ptr == nullptr || b[slot].size < need); };
        if (too_small(sc.cfg, sizeof(JlConfig)) || too_small(sc.bump, sizeof(JlBump)) || too_small(sc.indirect, 12))
            return fail(ctx, JH_ERR_INVALID, std::string("bad bindings for stage ") + jh_stage_name(stage) + ": config / bump / indirect buffer too small");
    }
    // Hold back the small stages a following stage can absorb (see Deferred); absorb or launch what is being held.
    // (not while profiling: a held-back stage's query would bracket nothing and its work would be charged to the stage that absorbs
    // it -- with the profiler on every command is launched as recorded and every query times its own stage)
    // (pathtag_reduce2: its consumer pathtag_scan1 redoes its sums in passing if it runs at most 16 workgroups itself --
    // PT_ABSORB_MAX in kernels_scan.hip; the reference always dispatches reduce2 with 256, one thread per entry of reduced2)
    // (the last pathtag scan: flatten's classification kernel, which reads every tag word and its monoid back, produces the monoids
    // in passing instead -- k_pathtag_scan_classify)
    const bool is_last_scan = (stage == JH_PATHTAG_SCAN_SMALL || stage == JH_PATHTAG_SCAN_LARGE) && gx > 0u && b.size() >= 4;
    const bool deferrable = !ctx->profiling && (stage == JH_BBOX_CLEAR || stage == JH_PATH_COUNT_SETUP || stage == JH_PATH_TILING_SETUP ||
                                                (stage == JH_PATHTAG_REDUCE2 && gx > 0u && gx <= 256u && b.size() >= 2) || is_last_scan);
    uint32_t absorb = 0u;
    JhBound extra;
    std::memset(&extra, 0, sizeof extra);
    if (deferrable) {
        // (only one kind of thing waits at a time, except what waits for flatten: the last pathtag scan, bbox_clear, Clear(bump) --
        // one of each)
        for (const Deferred& d : ctx->deferred) {
            const bool d_scan = !d.is_clear && (d.stage == JH_PATHTAG_SCAN_SMALL || d.stage == JH_PATHTAG_SCAN_LARGE);
            const bool joins = (stage == JH_BBOX_CLEAR && (d.is_clear || d_scan)) || (is_last_scan && (d.is_clear || d.stage == JH_BBOX_CLEAR));
            if (!joins) { JH_FLUSH(ctx); break; }
        }
    } else if (!ctx->deferred.empty()) {
        bool all = true;
        for (const Deferred& d : ctx->deferred) {
            bool ok = false;
            // (a dispatch with no workgroups launches nothing that could do the held-back work in passing -- an empty scene's
            // flatten, renderer.cpp: flatten_wgs = 0 -- so bbox_clear and Clear(bump) run as recorded: ADVICE r03)
            if (stage == JH_FLATTEN && b.size() >= 6 && gx > 0u) {
                if (d.is_clear) ok = d.clear_ptr == b[4].ptr && d.clear_bytes == b[4].size && b[4].size == sizeof(JlBump);
                else if (d.stage == JH_BBOX_CLEAR) ok = d.b[0].ptr == b[0].ptr && d.b[1].ptr == b[3].ptr && d.b[1].size == b[3].size;
                else  // the last pathtag scan: same config and scene, its output is the tag monoids flatten reads, and its threads
                      // (one per tag word = four tag bytes) cover flatten's (one per tag byte)
                    ok = (d.stage == JH_PATHTAG_SCAN_SMALL || d.stage == JH_PATHTAG_SCAN_LARGE) && d.b.size() >= 4 && d.b[0].ptr == b[0].ptr &&
                         d.b[1].ptr == b[1].ptr && d.b[1].size == b[1].size && d.b[3].ptr == b[2].ptr && d.b[3].size == b[2].size &&
                         (uint64_t)d.gx * 4u >= (uint64_t)gx && d.gy <= 1u && d.gz <= 1u;
            } else if (stage == JH_PATHTAG_SCAN1 && b.size() >= 3) {
                ok = !d.is_clear && d.stage == JH_PATHTAG_REDUCE2 && gx > 0u && gx <= 16u && gx <= d.gx && d.b[0].ptr == b[0].ptr &&
                     d.b[0].size == b[0].size && d.b[1].ptr == b[1].ptr && d.b[1].size == b[1].size;
            } else if (stage == JH_PATH_COUNT && indirect && b.size() >= 6) {
                ok = !d.is_clear && d.stage == JH_PATH_COUNT_SETUP && d.b[0].ptr == b[1].ptr && d.b[1].ptr == (void*)indirect;
            } else if (stage == JH_PATH_TILING && indirect && b.size() >= 6) {
                ok = !d.is_clear && d.stage == JH_PATH_TILING_SETUP && d.b[0].ptr == b[0].ptr && d.b[1].ptr == (void*)indirect;
            }
            all = all && ok;
        }
        if (all) {
            for (const Deferred& d : ctx->deferred) {
                if (d.is_clear) absorb |= JH_ABSORB_BUMP_CLEAR;
                else if (d.stage == JH_BBOX_CLEAR) absorb |= JH_ABSORB_BBOX_CLEAR;
                else if (d.stage == JH_PATHTAG_SCAN_SMALL || d.stage == JH_PATHTAG_SCAN_LARGE) {
                    absorb |= JH_ABSORB_PATHTAG_SCAN;
                    extra = d.b[2];
                    extra.width = d.gx;
                    extra.height = d.stage == JH_PATHTAG_SCAN_SMALL ? 1u : 0u;
                } else {
                    absorb |= JH_ABSORB_SETUP;
                    if (d.stage == JH_PATH_TILING_SETUP) extra = d.b[2];
                    if (d.stage == JH_PATHTAG_REDUCE2) extra.size = d.gx;  // (entries of reduced2 the held-back dispatch writes)
                }
            }
            ctx->deferred.clear();
        } else {
            JH_FLUSH(ctx);
        }
    }
    JhLaunch L;
    L.stream = ctx->stream;
    L.scratch = &ctx->scratch;
    L.gx = gx; L.gy = gy; L.gz = gz;
    L.absorb = absorb;
    L.extra = extra;
    L.b = b.data();
    L.nb = (int)b.size();
    L.images = images.data();
    L.n_images = (int)images.size();
    L.indirect = indirect;
    L.num_cus = ctx->num_cus;
    L.cfg_host = nullptr;
    L.band_row0 = ctx->band_row0;
    L.band_row1 = ctx->band_row1;
    L.clip_depth_hint = ctx->clip_depth_hint;
    L.hint_overflow = ctx->hint_overflow;
    L.debug_flatten = ctx->debug_flatten;
    L.image_table = nullptr;
    if ((int)images.size() > JH_FINE_INLINE_IMAGES && stage >= JH_FINE_AREA) {
        // More images than fit in the kernel arguments: fine indexes a device table of descriptors (the reference binds
        // an array of up to 2048 textures, wgpu.go:278).  The table is re-uploaded only when the bound set changes.
        std::vector<JhImageDesc> t(images.size());
        for (size_t i = 0; i < images.size(); i++) {
            t[i].ptr = images[i].ptr; t[i].width = images[i].width; t[i].height = images[i].height;
            t[i].srgb = images[i].format == JL_RGBA8_SRGB ? 1u : 0u; t[i].pad = 0u;
        }
        const bool same = t.size() == ctx->image_table_host.size() &&
                          std::memcmp(t.data(), ctx->image_table_host.data(), t.size() * sizeof(JhImageDesc)) == 0;
        if (!same) {
            if (ctx->capturing) return fail(ctx, JH_ERR_INVALID, "fine: the image table changed during graph capture (run the recording once eagerly first)");
            const uint64_t bytes = t.size() * sizeof(JhImageDesc);
            if (bytes > ctx->image_table_cap) {
                void* np = nullptr;
                if (hipMalloc(&np, pool_size_class(bytes)) != hipSuccess) return fail(ctx, JH_ERR_OOM, "fine: image table allocation failed");
                if (ctx->image_table) ctx->scratch.retired.push_back(ctx->image_table);
                ctx->image_table = np;
                ctx->image_table_cap = pool_size_class(bytes);
                ctx->generation++;
            }
            void* src = stage_copy(ctx, t.data(), bytes);
            if (!src) return fail(ctx, JH_ERR_OOM, "fine: pinned staging allocation failed");
            HIP_TRY(ctx, hipMemcpyAsync(ctx->image_table, src, bytes, hipMemcpyHostToDevice, ctx->stream));
            ctx->image_table_host = t;
        }
        L.image_table = (const JhImageDesc*)ctx->image_table;
    }
    if (n_bindings > 0 && bindings[0].kind == JH_BIND_BUFFER) {
        auto sh = ctx->config_shadow.find(bindings[0].id);
        if (sh != ctx->config_shadow.end()) L.cfg_host = &sh->second;
    }
    // binning and coarse address a bin by its index in a 256-entry table (one per lane of the 256-thread workgroup:
    // binning.wgsl:52,131; coarse.wgsl:153-176): a target of more than 256 bins of 256 x 256 px -- beyond 4096 px in a
    // direction -- silently loses the bins past the 256th in the reference.  Here the dispatch is refused instead.
    if ((stage == JH_BINNING || stage == JH_COARSE) && L.cfg_host) {
        const uint64_t wb = (L.cfg_host->width_in_tiles + 15u) / 16u, hb = (L.cfg_host->height_in_tiles + 15u) / 16u;
        if (wb * hb > 256u)
            return fail(ctx, JH_ERR_INVALID, std::string(jh_stage_name(stage)) + ": the target has " + std::to_string(wb * hb) +
                                                 " bins of 256 x 256 px, the pipeline addresses 256 (at most 4096 x 4096 px or any shape of <= 256 bins)");
    }
    ProfEntry pe;
    if (ctx->profiling) {
        auto get_event = [&](hipEvent_t* e) {
            if (!ctx->free_events.empty()) { *e = ctx->free_events.back(); ctx->free_events.pop_back(); return hipSuccess; }
            return hipEventCreate(e);
        };
        HIP_TRY(ctx, get_event(&pe.start));
        HIP_TRY(ctx, get_event(&pe.stop));
        pe.kind = JH_PROF_QUERY;
        pe.parent = ctx->prof_stack.empty() ? -1 : ctx->prof_stack.back();
        pe.stage = stage;
        pe.label = jh_stage_name(stage);  // ProfilerGroup.Compute(arena, shader.Label), wgpu.go:486,537
        pe.cpu_start_ms = now_ms();
        HIP_TRY(ctx, hipEventRecord(pe.start, ctx->stream));
    }
    if (deferrable) {  // held back: launched by the stage it waits for, or as recorded by the next flush
        Deferred d;
        d.stage = stage; d.gx = gx; d.gy = gy; d.gz = gz; d.b = b;
        ctx->deferred.push_back(d);
        rc = 0;
    } else {
        rc = launch_stage(stage, L);
        if (rc == -9) {
            if (ctx->profiling) { ctx->free_events.push_back(pe.start); ctx->free_events.push_back(pe.stop); }
            return fail(ctx, JH_ERR_UNSUPPORTED, std::string("stage not implemented: ") + jh_stage_name(stage));
        }
    }
    if (ctx->profiling) {
        HIP_TRY(ctx, hipEventRecord(pe.stop, ctx->stream));
        pe.cpu_end_ms = now_ms();
        ctx->prof.push_back(pe);
    }
    if (rc == -5)
        return fail(ctx, JH_ERR_OOM, ctx->capturing ? "a scratch array would have to grow during graph capture: run the recording once eagerly first"
                                                    : "scratch allocation failed");
    if (rc) return fail(ctx, JH_ERR_INVALID, std::string("bad bindings for stage ") + jh_stage_name(stage));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(ctx, e, jh_stage_name(stage));
    return JH_OK;
}

int jh_dispatch(jh_ctx* ctx, int stage, uint32_t gx, uint32_t gy, uint32_t gz, const jh_binding* bindings, int n_bindings) {
    return dispatch_common(ctx, stage, gx, gy, gz, nullptr, bindings, n_bindings);
}

int jh_dispatch_indirect(jh_ctx* ctx, int stage, uint64_t indirect_buffer_id, uint64_t offset, const jh_binding* bindings, int n_bindings) {
    if (!ctx) return JH_ERR_INVALID;
    auto it = ctx->buffers.find(indirect_buffer_id);
    if (it == ctx->buffers.end()) return fail(ctx, JH_ERR_INVALID, "dispatch_indirect: unknown indirect buffer");
    if (offset + 12 > it->second.size || (offset & 3)) return fail(ctx, JH_ERR_INVALID, "dispatch_indirect: bad offset");
    return dispatch_common(ctx, stage, 0, 1, 1, (const uint32_t*)((char*)it->second.ptr + offset), bindings, n_bindings);
}

// ---- hipGraph capture ----
int jh_graph_begin(jh_ctx* ctx) {
    if (!ctx) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    if (ctx->profiling) return fail(ctx, JH_ERR_INVALID, "jh_graph_begin: disable profiling first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    ctx->clean_at_capture_begin = ctx->scratch.clean_flags;
    return JH_OK;
}
int jh_graph_end(jh_ctx* ctx, void** graph_exec) {
    if (!ctx || !graph_exec) return JH_ERR_INVALID;
    *graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    const int frc = flush_deferred(ctx);  // held-back commands belong to the captured frame
    ctx->capturing = false;
    // Nothing ran: the device holds what it held when the capture began, and so must the host's picture of it (a launcher that
    // found a flag down recorded its fill INTO the graph and raised the flag -- the fill has not happened).
    ctx->scratch.clean_flags = ctx->clean_at_capture_begin;
    HIP_TRY(ctx, hipStreamEndCapture(ctx->stream, &graph));
    if (frc != JH_OK) { if (graph) (void)hipGraphDestroy(graph); return frc; }
    // the launches of the frame, counted on the captured graph itself
    uint32_t n_kernel = 0u, n_other = 0u;
    {
        size_t n = 0;
        if (hipGraphGetNodes(graph, nullptr, &n) == hipSuccess && n > 0) {
            std::vector<hipGraphNode_t> nodes(n);
            if (hipGraphGetNodes(graph, nodes.data(), &n) == hipSuccess)
                for (size_t i = 0; i < n; i++) {
                    hipGraphNodeType ty = hipGraphNodeTypeEmpty;
                    if (hipGraphNodeGetType(nodes[i], &ty) != hipSuccess) continue;
                    if (ty == hipGraphNodeTypeKernel) n_kernel++; else if (ty != hipGraphNodeTypeEmpty) n_other++;
                }
        }
    }
    // (Launch priorities per kernel node -- hipGraphKernelNodeSetAttribute(hipKernelNodeAttributePriority) -- are refused by this ROCm for
    // every node; priorities through streams do not help either: profiles/r04_fine_priority.txt.)
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipGraphInstantiate");
    *graph_exec = (void*)new JhGraph{exec, ctx->generation, n_kernel, n_other, ctx->clean_at_capture_begin};
    return JH_OK;
}
int jh_graph_node_counts(jh_ctx* ctx, void* graph_exec, uint32_t* kernel_nodes, uint32_t* other_nodes) {
    if (!ctx || !graph_exec) return JH_ERR_INVALID;
    const JhGraph* g = (const JhGraph*)graph_exec;
    if (kernel_nodes) *kernel_nodes = g->kernel_nodes;
    if (other_nodes) *other_nodes = g->other_nodes;
    return JH_OK;
}
int jh_graph_launch(jh_ctx* ctx, void* graph_exec) {
    if (!ctx || !graph_exec) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    JhGraph* g = (JhGraph*)graph_exec;
    // The graph holds raw device pointers (buffers, scratch, image table) and the kernel instantiations picked at capture
    // time: it is only valid while none of them has been freed, regrown or re-imported since.
    if (g->generation != ctx->generation)
        return fail(ctx, JH_ERR_INVALID, "jh_graph_launch: stale graph (a buffer, image or scratch array it refers to was freed, regrown or "
                                         "re-imported after the capture) -- capture again");
    // Counters the graph expects zeroed (it has no fill for them) and that are not: the frame before did not run to its end
    // (a stage failed half-way) or the scratch was poisoned.  Zero them here -- the eager path's own remedy -- then replay.
    const uint32_t dirty = g->assumes_clean & ~ctx->scratch.clean_flags;
    if (dirty) {
        static const struct { uint32_t flag; int slot; } kSlots[] = {
            {JH_CLEAN_FL_CTR, JH_SCR_FL_CTR}, {JH_CLEAN_BD_CTR, JH_SCR_BD_CTR}, {JH_CLEAN_SCAN, JH_SCR_SCAN_TMP}, {JH_CLEAN_PC_TOT, JH_SCR_PC_TOT}};
        for (const auto& k : kSlots)
            if ((dirty & k.flag) && ctx->scratch.ptr[k.slot] && ctx->scratch.cap[k.slot])
                HIP_TRY(ctx, hipMemsetAsync(ctx->scratch.ptr[k.slot], 0, ctx->scratch.cap[k.slot], ctx->stream));
        ctx->scratch.clean_flags |= dirty;
        ctx->graph_self_cleans++;
    }
    HIP_TRY(ctx, hipGraphLaunch(g->exec, ctx->stream));
    return JH_OK;
}
int jh_graph_destroy(jh_ctx* ctx, void* graph_exec) {
    if (!ctx) return JH_ERR_INVALID;
    if (graph_exec) {
        JhGraph* g = (JhGraph*)graph_exec;
        hipError_t e = hipGraphExecDestroy(g->exec);
        delete g;
        if (e != hipSuccess) return hip_fail(ctx, e, "hipGraphExecDestroy");
    }
    return JH_OK;
}

// ---- profiling ----
int jh_profile_enable(jh_ctx* ctx, int on) {
    if (!ctx) return JH_ERR_INVALID;
    ctx->profiling = on != 0;
    return JH_OK;
}

// Profiler.Start / ProfilerGroup.Nest (profiler.go:49-65, 138-158): opens a group under the innermost open one.
int jh_profile_group_begin(jh_ctx* ctx, const char* label) {
    if (!ctx) return JH_ERR_INVALID;
    if (!ctx->profiling) return JH_OK;  // a nil profiler accepts every call (profiler.go:45-52)
    ProfEntry g;
    g.kind = JH_PROF_GROUP;
    g.parent = ctx->prof_stack.empty() ? -1 : ctx->prof_stack.back();
    g.stage = -1;
    g.label = label ? label : "";
    g.cpu_start_ms = now_ms();
    ctx->prof_stack.push_back((int)ctx->prof.size());
    ctx->prof.push_back(g);
    return JH_OK;
}
// ProfilerGroup.End (profiler.go:113-125); ending a group that was never begun is an error (the reference panics).
int jh_profile_group_end(jh_ctx* ctx) {
    if (!ctx) return JH_ERR_INVALID;
    if (!ctx->profiling && ctx->prof_stack.empty()) return JH_OK;
    if (ctx->prof_stack.empty()) return fail(ctx, JH_ERR_INVALID, "jh_profile_group_end: no open group");
    ctx->prof[ctx->prof_stack.back()].cpu_end_ms = now_ms();
    ctx->prof_stack.pop_back();
    return JH_OK;
}

static void prof_recycle(jh_ctx* ctx) {
    for (auto& p : ctx->prof)
        if (p.kind == JH_PROF_QUERY) { ctx->free_events.push_back(p.start); ctx->free_events.push_back(p.stop); }
    ctx->prof.clear();
    ctx->prof_stack.clear();
}

int jh_profile_collect(jh_ctx* ctx, jh_profile_record* out, int max) {
    if (!ctx) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int n = 0;
    for (auto& p : ctx->prof) {
        if (p.kind != JH_PROF_QUERY) continue;
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, p.start, p.stop);
        if (out && n < max) {
            out[n].stage = p.stage;
            out[n].pad = 0;
            out[n].ms = ms;
            n++;
        }
    }
    prof_recycle(ctx);
    return n;
}

// Profiler.Collect (profiler.go:337-385): the tree, flattened in creation order (a node's parent always precedes it).
int jh_profile_collect_tree(jh_ctx* ctx, jh_profile_node* out, int max) {
    if (!ctx) return JH_ERR_INVALID;
    if (!ctx->prof_stack.empty()) return fail(ctx, JH_ERR_INVALID, "jh_profile_collect_tree: a group is still open");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    hipEvent_t base = nullptr;
    double cpu0 = 0.0;
    bool have_cpu0 = false;
    for (auto& p : ctx->prof) {
        if (!have_cpu0) { cpu0 = p.cpu_start_ms; have_cpu0 = true; }
        if (p.kind == JH_PROF_QUERY && !base) base = p.start;
    }
    int n = 0;
    for (auto& p : ctx->prof) {
        if (!out || n >= max) break;
        jh_profile_node& o = out[n++];
        std::memset(&o, 0, sizeof o);
        o.kind = p.kind;
        o.parent = p.parent;
        o.stage = p.stage;
        std::strncpy(o.label, p.label.c_str(), sizeof(o.label) - 1);
        o.cpu_start_ms = p.cpu_start_ms - cpu0;
        o.cpu_end_ms = p.cpu_end_ms - cpu0;
        if (p.kind == JH_PROF_QUERY) {
            float a = 0.0f, d = 0.0f;
            (void)hipEventElapsedTime(&a, base, p.start);
            (void)hipEventElapsedTime(&d, p.start, p.stop);
            o.gpu_start_ms = a;
            o.gpu_end_ms = a + d;
        }
    }
    // a group's GPU interval = the hull of the queries below it (filled bottom-up: children follow their parents)
    for (int i = n - 1; i >= 0; i--) {
        if (out[i].kind == JH_PROF_QUERY || (out[i].kind == JH_PROF_GROUP && out[i].gpu_end_ms > out[i].gpu_start_ms)) {
            int par = out[i].parent;
            if (par >= 0 && par < n) {
                if (out[par].gpu_end_ms <= out[par].gpu_start_ms) { out[par].gpu_start_ms = out[i].gpu_start_ms; out[par].gpu_end_ms = out[i].gpu_end_ms; }
                else {
                    if (out[i].gpu_start_ms < out[par].gpu_start_ms) out[par].gpu_start_ms = out[i].gpu_start_ms;
                    if (out[i].gpu_end_ms > out[par].gpu_end_ms) out[par].gpu_end_ms = out[i].gpu_end_ms;
                }
            }
        }
    }
    prof_recycle(ctx);
    return n;
}

int jh_selftest_math_launch(hipStream_t stream, int op, const float* a, const float* b, float* out, uint32_t n);

int jh_selftest_math(jh_ctx* ctx, int op, const float* a, const float* b, float* out, uint32_t n) {
    if (!ctx || !a || !out) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    size_t bytes = (size_t)n * 4;
    HIP_TRY(ctx, hipMalloc(&da, bytes ? bytes : 4));
    HIP_TRY(ctx, hipMalloc(&dout, bytes ? bytes : 4));
    if (b) HIP_TRY(ctx, hipMalloc(&db, bytes ? bytes : 4));
    HIP_TRY(ctx, hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (b) HIP_TRY(ctx, hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = jh_selftest_math_launch(ctx->stream, op, da, db, dout, n);
    if (rc == 0) {
        HIP_TRY(ctx, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    (void)hipFree(da); (void)hipFree(dout);
    if (db) (void)hipFree(db);
    return rc == 0 ? JH_OK : fail(ctx, JH_ERR_DEVICE, "selftest launch failed");
}

int jh_selftest_atomics_launch(hipStream_t stream, int form, uint32_t seed, uint32_t n_waves);

int jh_selftest_atomics(jh_ctx* ctx, int form, uint32_t seed, uint32_t n_waves) {
    if (!ctx) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = jh_selftest_atomics_launch(ctx->stream, form, seed, n_waves);
    if (rc == -1) return fail(ctx, JH_ERR_INVALID, "selftest_atomics: form 0..2, 1..4096 waves");
    if (rc < 0) return fail(ctx, JH_ERR_DEVICE, "selftest_atomics: runtime error");
    return rc;
}

int jh_debug_poison_scratch(jh_ctx* ctx, int byte) {
    if (!ctx) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < JH_SCR_COUNT; i++)
        if (ctx->scratch.ptr[i] && ctx->scratch.cap[i]) HIP_TRY(ctx, hipMemsetAsync(ctx->scratch.ptr[i], byte, ctx->scratch.cap[i], ctx->stream));
    ctx->scratch.clean_flags = 0u;
    return JH_OK;
}

int jh_set_clip_depth_hint(jh_ctx* ctx, uint32_t max_depth) {
    if (!ctx) return JH_ERR_INVALID;
    // (a captured graph has the layout of its capture baked in and never looks at the hint again; its scratch pointer stays
    // valid until the array is regrown, which bumps the generation by itself)
    ctx->clip_depth_hint = max_depth;
    return JH_OK;
}
// Tests: flatten's temporary is cut into regions that a frame only fills up (and leaves behind, marking the slots at their ends
// empty) when it comes close to the capacity of its line buffer.  Bit 0: every wave starts in region 0; bit 1: eight regions
// whatever the capacity -- so that ordinary scenes take that path; bit 2: batches allocate their slots job by job (the product does
// above 51 200 lines per batch).  Results never depend on it.
int jh_debug_flatten_regions(jh_ctx* ctx, uint32_t flags) {
    if (!ctx) return JH_ERR_INVALID;
    ctx->debug_flatten = flags;
    ctx->generation++;  // (a captured graph holds the old kernel arguments)
    return JH_OK;
}
// Blend-stack saves the fine stage dropped since the last reset because jh_set_clip_depth_hint promised a shallower scene than
// it got (every one of them is a wrong pixel colour): 0 for every correct hint.  Synchronises the stream.
int jh_debug_clip_hint_overflows(jh_ctx* ctx, uint32_t* count, int reset) {
    if (!ctx || !ctx->hint_overflow) return JH_ERR_INVALID;
    JH_FLUSH(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (count) {
        HIP_TRY(ctx, hipMemcpyAsync(count, ctx->hint_overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (reset) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->hint_overflow, 0, 4, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return JH_OK;
}
#if defined(FINE_TIMING) || defined(FINE_EB_STATS)
extern "C" int jh_debug_fine_timing(jh_ctx* ctx, unsigned long long* out6, int reset) {
    if (!ctx || !ctx->hint_overflow) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (out6) { HIP_TRY(ctx, hipMemcpyAsync(out6, ctx->hint_overflow + 8, 48, hipMemcpyDeviceToHost, ctx->stream)); HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); }
    if (reset) { HIP_TRY(ctx, hipMemsetAsync(ctx->hint_overflow + 8, 0, 48, ctx->stream)); HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); }
    return JH_OK;
}
#endif
uint64_t jh_debug_scratch_bytes(jh_ctx* ctx, int slot) {
    if (!ctx) return 0;
    if (slot == -1) {  // all of them
        uint64_t sum = 0;
        for (int i = 0; i < JH_SCR_COUNT; i++) sum += ctx->scratch.cap[i];
        return sum;
    }
    return (slot >= 0 && slot < JH_SCR_COUNT) ? ctx->scratch.cap[slot] : 0;
}
// The scratch arrays only grow; a context that has rendered one much larger frame (or one frame with the estimator's generous
// bump sizes before the regrow loop settled on the real ones) keeps that size.  This gives all of it back: waits for the
// stream, frees every array; the next frame allocates what it needs.  Captured graphs hold the old pointers: they are stale
// afterwards (generation check) and must be captured again.
int jh_scratch_trim(jh_ctx* ctx) {
    if (!ctx) return JH_ERR_INVALID;
    if (ctx->capturing) return fail(ctx, JH_ERR_INVALID, "jh_scratch_trim during a graph capture");
    JH_FLUSH(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    scratch_release_retired(&ctx->scratch);
    for (int i = 0; i < JH_SCR_COUNT; i++) {
        if (ctx->scratch.base[i]) (void)hipFree(ctx->scratch.base[i]);
        ctx->scratch.base[i] = nullptr;
        ctx->scratch.ptr[i] = nullptr;
        ctx->scratch.cap[i] = 0;
    }
    ctx->scratch.clean_flags = 0;  // (new memory holds anything: every self-cleaned array is filled again on its next use)
    ctx->generation++;
    return JH_OK;
}
uint64_t jh_debug_graph_self_cleans(jh_ctx* ctx) { return ctx ? ctx->graph_self_cleans : 0; }

int jh_device_info(jh_ctx* ctx, char* name, int name_len, int* compute_units, uint64_t* total_mem) {
    if (!ctx) return JH_ERR_INVALID;
    if (name && name_len > 0) {
        std::strncpy(name, ctx->name.c_str(), (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (compute_units) *compute_units = ctx->num_cus;
    if (total_mem) *total_mem = ctx->total_mem;
    return JH_OK;
}

uint64_t jh_pool_bytes(jh_ctx* ctx) { return ctx ? ctx->pool_bytes : 0; }

}  // extern "C"
