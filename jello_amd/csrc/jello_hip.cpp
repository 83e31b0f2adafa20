// jello_hip.cpp -- implementation of the C ABI in include/jello_hip.h: context, pooled device
// buffers keyed by the recording's ResourceIDs, images in linear device memory, per-stage dispatch
// onto the HIP kernels, hipEvent profiling.  Replaces engine/wgpu_engine (wgpu.go:322-643) below
// the renderer.Recording boundary.  There is no CPU fallback here: if HIP is unavailable every
// call fails with JH_ERR_NO_DEVICE / JH_ERR_DEVICE.
#include "../../include/jello_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "kcommon.h"

struct JhScratch {
    void* ptr[JH_SCR_COUNT];
    uint64_t cap[JH_SCR_COUNT];
    std::vector<void*> retired;  // old allocations kept until the next sync (kernels may still use them)
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

struct ProfEntry {
    int stage;
    hipEvent_t start, stop;
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
    std::string last_error;
    uint32_t band_row0 = 0u, band_row1 = 0xffffffffu;  // jh_set_band
};

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
    uint64_t cap = pool_size_class(bytes);
    void* p = nullptr;
    if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
    if (s->ptr[slot]) s->retired.push_back(s->ptr[slot]);
    s->ptr[slot] = p;
    s->cap[slot] = cap;
    return p;
}

static void scratch_release_retired(JhScratch* s) {
    for (void* p : s->retired) (void)hipFree(p);
    s->retired.clear();
}

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
    ctx->scratch.ctx = ctx;
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
        if (ctx->scratch.ptr[i]) (void)hipFree(ctx->scratch.ptr[i]);
    scratch_release_retired(&ctx->scratch);
    for (auto& p : ctx->prof) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
    for (auto& e : ctx->free_events) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* jh_last_error(jh_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int jh_set_stream(jh_ctx* ctx, void* hip_stream) {
    if (!ctx) return JH_ERR_INVALID;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    scratch_release_retired(&ctx->scratch);
    return JH_OK;
}

// ---- buffers ----
static int buffer_get_or_create(jh_ctx* ctx, uint64_t id, uint64_t size, Alloc** out) {
    auto it = ctx->buffers.find(id);
    if (it != ctx->buffers.end()) {
        if (it->second.size >= size || !it->second.owned) {
            *out = &it->second;
            return 0;
        }
        // grow: return the old allocation to the pool
        ctx->pool.insert({it->second.capacity, it->second.ptr});
        ctx->buffers.erase(it);
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
    auto it = ctx->buffers.find(id);
    if (it != ctx->buffers.end() && it->second.owned) ctx->pool.insert({it->second.capacity, it->second.ptr});
    Alloc a;
    a.ptr = device_ptr;
    a.size = size;
    a.capacity = size;
    a.owned = false;
    ctx->buffers[id] = a;
    return JH_OK;
}

int jh_upload(jh_ctx* ctx, uint64_t id, const void* data, uint64_t size) {
    if (!ctx || (!data && size)) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Alloc* a;
    int rc = buffer_get_or_create(ctx, id, size, &a);
    if (rc) return rc;
    if (size == sizeof(JlConfig)) std::memcpy(&ctx->config_shadow[id], data, sizeof(JlConfig));
    if (size) HIP_TRY(ctx, hipMemcpyAsync(a->ptr, data, size, hipMemcpyHostToDevice, ctx->stream));
    // The host slice is only valid for the duration of the call (reference: queue.WriteBuffer copies).
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
    if (n) HIP_TRY(ctx, hipMemsetAsync((char*)a.ptr + offset, 0, n, ctx->stream));
    return JH_OK;
}

int jh_download(jh_ctx* ctx, uint64_t id, void* dst, uint64_t offset, uint64_t size) {
    if (!ctx || !dst) return JH_ERR_INVALID;
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
    auto it = ctx->buffers.find(id);
    if (it == ctx->buffers.end()) return JH_OK;  // the reference ignores frees of unknown ids (wgpu.go:601-603)
    if (it->second.owned) ctx->pool.insert({it->second.capacity, it->second.ptr});
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
        if (it->second.owned) ctx->pool.insert({it->second.capacity, it->second.ptr});
        ctx->images.erase(it);
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
    uint64_t bpp = format_bpp(format);
    if (!bpp) return fail(ctx, JH_ERR_INVALID, "jh_image_import: bad format");
    auto it = ctx->images.find(id);
    if (it != ctx->images.end() && it->second.owned) ctx->pool.insert({it->second.capacity, it->second.ptr});
    Alloc a;
    a.ptr = device_ptr; a.size = (uint64_t)width * height * bpp; a.capacity = a.size; a.owned = false;
    a.width = width; a.height = height; a.format = format; a.written = true;
    ctx->images[id] = a;
    return JH_OK;
}

int jh_image_upload(jh_ctx* ctx, uint64_t id, uint32_t width, uint32_t height, int format, const void* data, uint64_t size) {
    int rc = jh_image_create(ctx, id, width, height, format);
    if (rc) return rc;
    Alloc& a = ctx->images[id];
    if (size > a.size) size = a.size;
    if (size) HIP_TRY(ctx, hipMemcpyAsync(a.ptr, data, size, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // an all-zero image (the 1x1 placeholder of render.go:115-124) contributes zero texels: bind it as absent
    bool nonzero = false;
    const uint8_t* bytes = (const uint8_t*)data;
    for (uint64_t i = 0; i < size && !nonzero; i++) nonzero = bytes[i] != 0;
    a.written = nonzero;
    return JH_OK;
}

int jh_image_download(jh_ctx* ctx, uint64_t id, void* dst, uint64_t size) {
    if (!ctx || !dst) return JH_ERR_INVALID;
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
    auto it = ctx->images.find(id);
    if (it == ctx->images.end()) return JH_OK;
    if (it->second.owned) ctx->pool.insert({it->second.capacity, it->second.ptr});
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

static int dispatch_common(jh_ctx* ctx, int stage, uint32_t gx, uint32_t gy, uint32_t gz, const uint32_t* indirect, const jh_binding* bindings,
                           int n_bindings) {
    if (!ctx || stage < 0 || stage >= JH_STAGE_COUNT || (n_bindings && !bindings)) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<JhBound> b, images;
    int rc = resolve_bindings(ctx, bindings, n_bindings, b, images);
    if (rc) return rc;
    JhLaunch L;
    L.stream = ctx->stream;
    L.scratch = &ctx->scratch;
    L.gx = gx; L.gy = gy; L.gz = gz;
    L.b = b.data();
    L.nb = (int)b.size();
    L.images = images.data();
    L.n_images = (int)images.size();
    L.indirect = indirect;
    L.num_cus = ctx->num_cus;
    L.cfg_host = nullptr;
    L.band_row0 = ctx->band_row0;
    L.band_row1 = ctx->band_row1;
    if (n_bindings > 0 && bindings[0].kind == JH_BIND_BUFFER) {
        auto sh = ctx->config_shadow.find(bindings[0].id);
        if (sh != ctx->config_shadow.end()) L.cfg_host = &sh->second;
    }
    ProfEntry pe;
    if (ctx->profiling) {
        auto get_event = [&](hipEvent_t* e) {
            if (!ctx->free_events.empty()) { *e = ctx->free_events.back(); ctx->free_events.pop_back(); return hipSuccess; }
            return hipEventCreate(e);
        };
        HIP_TRY(ctx, get_event(&pe.start));
        HIP_TRY(ctx, get_event(&pe.stop));
        pe.stage = stage;
        HIP_TRY(ctx, hipEventRecord(pe.start, ctx->stream));
    }
    switch (stage) {
        case JH_PATHTAG_REDUCE: case JH_PATHTAG_REDUCE2: case JH_PATHTAG_SCAN1: case JH_PATHTAG_SCAN_SMALL: case JH_PATHTAG_SCAN_LARGE:
            rc = jh_launch_pathtag(L, stage);
            break;
        case JH_BBOX_CLEAR: rc = jh_launch_bbox_clear(L); break;
        case JH_FLATTEN: rc = jh_launch_flatten(L); break;
        case JH_DRAW_REDUCE: rc = jh_launch_draw_reduce(L); break;
        case JH_DRAW_LEAF: rc = jh_launch_draw_leaf(L); break;
        case JH_CLIP_REDUCE: rc = jh_launch_clip_reduce(L); break;
        case JH_CLIP_LEAF: rc = jh_launch_clip_leaf(L); break;
        case JH_BINNING: rc = jh_launch_binning(L); break;
        case JH_TILE_ALLOC: rc = jh_launch_tile_alloc(L); break;
        case JH_BACKDROP_DYN: rc = jh_launch_backdrop_dyn(L); break;
        case JH_PATH_COUNT_SETUP: rc = jh_launch_path_count_setup(L); break;
        case JH_PATH_COUNT: rc = jh_launch_path_count(L); break;
        case JH_COARSE: rc = jh_launch_coarse(L); break;
        case JH_PATH_TILING_SETUP: rc = jh_launch_path_tiling_setup(L); break;
        case JH_PATH_TILING: rc = jh_launch_path_tiling(L); break;
        case JH_FINE_AREA: rc = jh_launch_fine_area(L); break;
        case JH_FINE_MSAA8: rc = jh_launch_fine_msaa(L, 8); break;
        case JH_FINE_MSAA16: rc = jh_launch_fine_msaa(L, 16); break;
        default:
            if (ctx->profiling) { ctx->free_events.push_back(pe.start); ctx->free_events.push_back(pe.stop); }
            return fail(ctx, JH_ERR_UNSUPPORTED, std::string("stage not implemented: ") + jh_stage_name(stage));
    }
    if (ctx->profiling) {
        HIP_TRY(ctx, hipEventRecord(pe.stop, ctx->stream));
        ctx->prof.push_back(pe);
    }
    if (rc == -5) return fail(ctx, JH_ERR_OOM, "scratch allocation failed");
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
    if (ctx->profiling) return fail(ctx, JH_ERR_INVALID, "jh_graph_begin: disable profiling first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    return JH_OK;
}
int jh_graph_end(jh_ctx* ctx, void** graph_exec) {
    if (!ctx || !graph_exec) return JH_ERR_INVALID;
    hipGraph_t graph = nullptr;
    HIP_TRY(ctx, hipStreamEndCapture(ctx->stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipGraphInstantiate");
    *graph_exec = (void*)exec;
    return JH_OK;
}
int jh_graph_launch(jh_ctx* ctx, void* graph_exec) {
    if (!ctx || !graph_exec) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipGraphLaunch((hipGraphExec_t)graph_exec, ctx->stream));
    return JH_OK;
}
int jh_graph_destroy(jh_ctx* ctx, void* graph_exec) {
    if (!ctx) return JH_ERR_INVALID;
    if (graph_exec) HIP_TRY(ctx, hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return JH_OK;
}

// ---- profiling ----
int jh_profile_enable(jh_ctx* ctx, int on) {
    if (!ctx) return JH_ERR_INVALID;
    ctx->profiling = on != 0;
    return JH_OK;
}

int jh_profile_collect(jh_ctx* ctx, jh_profile_record* out, int max) {
    if (!ctx) return JH_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int n = 0;
    for (auto& p : ctx->prof) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, p.start, p.stop);
        if (out && n < max) {
            out[n].stage = p.stage;
            out[n].pad = 0;
            out[n].ms = ms;
            n++;
        }
        ctx->free_events.push_back(p.start);
        ctx->free_events.push_back(p.stop);
    }
    ctx->prof.clear();
    return n;
}

int jh_selftest_math_launch(hipStream_t stream, int op, const float* a, const float* b, float* out, uint32_t n);

int jh_selftest_math(jh_ctx* ctx, int op, const float* a, const float* b, float* out, uint32_t n) {
    if (!ctx || !a || !out) return JH_ERR_INVALID;
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
