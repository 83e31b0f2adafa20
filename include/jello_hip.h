/* jello_hip.h -- C ABI of libjello_hip.so, the MI355X (gfx950) replacement for Jello's
 * engine/wgpu_engine.  These are exactly the entry points a Go `engine/hip_engine` package binds
 * with cgo to replay a renderer.Recording (see INTEGRATION.md for the cgo stub).
 *
 * Mapping to the reference interface each call replaces:
 *   jh_create / jh_destroy        wgpu_engine.New                      engine/wgpu_engine/wgpu.go:157-178
 *   jh_upload                     Upload / UploadUniform commands      wgpu.go:353-370 (queue.WriteBuffer)
 *   jh_image_upload               UploadImage command                  wgpu.go:372-418
 *   jh_clear                      Clear command                        wgpu.go:565-585
 *   jh_dispatch                   Dispatch command                     wgpu.go:454-497
 *   jh_dispatch_indirect          DispatchIndirect command             wgpu.go:499-552
 *   jh_download                   Download command + map               wgpu.go:554-563, 645-657
 *   jh_free / jh_image_free       FreeBuffer / FreeImage (pool return) wgpu.go:587-616, 772-808
 *   jh_image_import/buffer_import ExternalResource{ExternalImage,..}   wgpu.go:81-93, lib.go:257-262
 *   jh_image_write                WriteImage command                   wgpu.go:422-452 (queue.WriteTexture)
 *   jh_profile_enable/collect     ProfilerGroup.Compute timestamps     engine/wgpu_engine/profiler.go:160-177
 *   jh_profile_group_begin/end    Profiler.Start / Nest / End          profiler.go:49-65, 113-158
 *   jh_profile_collect_tree       Profiler.Collect (nested results)    profiler.go:304-385
 *   jh_stage                      renderer.FullShaders field order     renderer/render.go:17-43
 * Binding order for every stage is the WGSL @binding order = renderer/render.go dispatch order.
 *
 * Conventions: plain pointers and sizes only; every call returns 0 on success or a negative
 * jh_status (never aborts -- the reference panics, wgpu.go:77,213,282,544,558,594,955);
 * one jh_ctx = one device + one stream, not thread-safe; several contexts (one per GPU -- or two on
 * ONE GPU that take a stream of frames in turn, which fills the holes a single chain of dependent
 * launches leaves: INTEGRATION.md) may be used concurrently.  Work is enqueued asynchronously; only
 * jh_download, jh_image_download, jh_sync and jh_profile_collect[_tree] wait for the device.
 * Uploads copy the caller's bytes into a pinned staging arena before returning (the slice may be
 * reused at once, as with queue.WriteBuffer) and leave the DMA in flight.
 */
#ifndef JELLO_HIP_H
#define JELLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jh_ctx jh_ctx;

typedef enum jh_status {
    JH_OK = 0,
    JH_ERR_INVALID = -1,     /* bad argument / unknown id / wrong binding count */
    JH_ERR_DEVICE = -2,      /* HIP runtime error (jh_last_error has the text) */
    JH_ERR_NO_DEVICE = -3,   /* no usable GPU */
    JH_ERR_UNSUPPORTED = -4, /* stage not implemented by this build */
    JH_ERR_OOM = -5
} jh_status;

/* renderer.FullShaders field order (renderer/render.go:17-43) */
typedef enum jh_stage {
    JH_PATHTAG_REDUCE = 0,
    JH_PATHTAG_REDUCE2 = 1,
    JH_PATHTAG_SCAN1 = 2,
    JH_PATHTAG_SCAN_SMALL = 3,
    JH_PATHTAG_SCAN_LARGE = 4,
    JH_BBOX_CLEAR = 5,
    JH_FLATTEN = 6,
    JH_DRAW_REDUCE = 7,
    JH_DRAW_LEAF = 8,
    JH_CLIP_REDUCE = 9,
    JH_CLIP_LEAF = 10,
    JH_BINNING = 11,
    JH_TILE_ALLOC = 12,
    JH_BACKDROP_DYN = 13,
    JH_PATH_COUNT_SETUP = 14,
    JH_PATH_COUNT = 15,
    JH_COARSE = 16,
    JH_PATH_TILING_SETUP = 17,
    JH_PATH_TILING = 18,
    JH_FINE_AREA = 19,
    JH_FINE_MSAA8 = 20,
    JH_FINE_MSAA16 = 21,
    JH_STAGE_COUNT = 22
} jh_stage;

/* renderer.ResourceProxy (renderer/recording.go:23-36) */
typedef enum jh_binding_kind { JH_BIND_BUFFER = 1, JH_BIND_IMAGE = 2, JH_BIND_IMAGE_ARRAY = 3 } jh_binding_kind;
typedef struct jh_binding {
    uint32_t kind;       /* jh_binding_kind */
    uint32_t count;      /* JH_BIND_IMAGE_ARRAY: number of ids */
    uint64_t id;         /* buffer / image ResourceID */
    const uint64_t* ids; /* JH_BIND_IMAGE_ARRAY */
} jh_binding;

typedef struct jh_profile_record {
    int32_t stage; /* jh_stage */
    uint32_t pad;
    float ms; /* device time of the whole stage (all of its kernels), hipEvent pair */
} jh_profile_record;

/* One node of the nested profile (ProfilerResult, profiler.go:289-302): a group (label, CPU interval, children) or a GPU
 * query (one per dispatch, label = the stage's name as in wgpu.go:486).  Nodes come in creation order, so a node's
 * parent always precedes it; times are milliseconds relative to the first node (CPU clock) / the first query (GPU). */
typedef enum jh_profile_kind { JH_PROF_GROUP = 0, JH_PROF_QUERY = 1 } jh_profile_kind;
typedef struct jh_profile_node {
    int32_t kind;   /* jh_profile_kind */
    int32_t parent; /* index of the enclosing group in the same array, -1 at top level */
    int32_t stage;  /* query: jh_stage; group: -1 */
    uint32_t pad;
    char label[48];
    double cpu_start_ms, cpu_end_ms; /* group: Start/Nest .. End; query: the enqueue call */
    float gpu_start_ms, gpu_end_ms;  /* query: its hipEvent pair; group: hull of the queries below it (0,0 if none) */
} jh_profile_node;

/* ---- context ---- */
int jh_create(jh_ctx** out, int device);
void jh_destroy(jh_ctx* ctx);
const char* jh_last_error(jh_ctx* ctx);
const char* jh_stage_name(int stage);
/* Run on a caller-owned HIP stream (e.g. torch's current stream); NULL = the context's own. */
int jh_set_stream(jh_ctx* ctx, void* hip_stream);
/* A stream of the lowest (level < 0), default (0) or highest (> 0) launch priority of the context's device -- for a caller that puts
 * a stage on a stream of its own (no reference counterpart; tools/fine_priority.py).  jh_stream_destroy releases it. */
int jh_stream_create(jh_ctx* ctx, int level, void** hip_stream);
int jh_stream_destroy(jh_ctx* ctx, void* hip_stream);
int jh_sync(jh_ctx* ctx);
/* Band mode (sharding ONE target over several GPUs): this context writes the PTCL and rasterises only the
 * bin rows [bin_row0, bin_row1) (a bin row = 16 tile rows = 256 pixel rows).  Every other stage, and the counting pass
 * of coarse, still covers the whole scene, so all allocation offsets -- hence every PTCL word and segment index of the
 * band -- are those of the unsharded run; the target image is written in the band's rows only.  (0, UINT32_MAX) = whole
 * target (default).  The reference has no counterpart: one wgpu device renders the whole target (render.go:399-434). */
int jh_set_band(jh_ctx* ctx, uint32_t bin_row0, uint32_t bin_row1);
/* Optional: an upper bound of the nesting depth of clip / blend layers (open BEGIN_CLIPs at any point of the draw tag stream)
 * in the recordings run next; 0 = unknown.  fine keeps the blend-stack levels between its own LDS level and the WGSL's
 * blend_spill (fine.wgsl:938-949, BLEND_STACK_SPLIT = 4) in a per-tile scratch slice of 4 KiB per level: without the hint it
 * reserves the worst case, three levels (768 MiB for a 4096 x 4096 target), with it 0 / 1 / 2 / 3 levels for depth <= 1 / 2 /
 * 3 / >= 4.  A hint that is too SMALL loses the saved colours of the deeper levels (wrong pixels, no out-of-bounds access).  The
 * engine shims count the depth off encoding.DrawTags in RenderToTexture; the reference has no counterpart (its blend stack is
 * `var blend_stack: array<array<vec4<f32>, 4>, 4>` in registers). */
int jh_set_clip_depth_hint(jh_ctx* ctx, uint32_t max_depth);
/* Blend-stack saves fine dropped since the last reset because the hint above was smaller than the scene's real nesting depth
 * (each one is a wrong pixel colour; memory safety is never at stake): 0 after every frame rendered with a correct hint or
 * with no hint.  A caller that sets hints should check it in debug builds.  Synchronises the context's stream. */
int jh_debug_clip_hint_overflows(jh_ctx* ctx, uint32_t* count, int reset);

/* ---- buffers (ids are the recording's ResourceIDs; sizes in bytes) ---- */
int jh_buffer_create(jh_ctx* ctx, uint64_t id, uint64_t size);
int jh_buffer_import(jh_ctx* ctx, uint64_t id, void* device_ptr, uint64_t size); /* caller keeps ownership */
int jh_upload(jh_ctx* ctx, uint64_t id, const void* data, uint64_t size);        /* creates the buffer if needed */
int jh_clear(jh_ctx* ctx, uint64_t id, uint64_t offset, int64_t size);           /* size < 0: to the end */
int jh_download(jh_ctx* ctx, uint64_t id, void* dst, uint64_t offset, uint64_t size);
int jh_free(jh_ctx* ctx, uint64_t id); /* returns the allocation to the pool */
void* jh_buffer_device_ptr(jh_ctx* ctx, uint64_t id);
uint64_t jh_buffer_size(jh_ctx* ctx, uint64_t id);

/* ---- images: linear device memory, row-major, format = renderer.ImageFormat ----
 * An image that was only created (never uploaded / imported), or whose upload was all zero bytes (the 1x1 placeholder
 * of render.go:115-124), samples as transparent black like a fresh wgpu texture.  JL_RGBA8_SRGB texels are decoded to
 * linear when fine samples them (an rgba8unorm-srgb texture view); alpha is linear. */
int jh_image_create(jh_ctx* ctx, uint64_t id, uint32_t width, uint32_t height, int format);
int jh_image_import(jh_ctx* ctx, uint64_t id, void* device_ptr, uint32_t width, uint32_t height, int format);
int jh_image_upload(jh_ctx* ctx, uint64_t id, uint32_t width, uint32_t height, int format, const void* data, uint64_t size);
/* WriteImage: `data` holds height rows of width texels, tightly packed; written to the rectangle at (x, y). */
int jh_image_write(jh_ctx* ctx, uint64_t id, uint32_t x, uint32_t y, uint32_t width, uint32_t height, const void* data, uint64_t size);
int jh_image_download(jh_ctx* ctx, uint64_t id, void* dst, uint64_t size);
int jh_image_free(jh_ctx* ctx, uint64_t id);
void* jh_image_device_ptr(jh_ctx* ctx, uint64_t id);

/* ---- dispatch ----
 * One call per recorded Dispatch: `stage` is a jh_stage (= the field order of renderer.FullShaders), (gx, gy, gz) the
 * recorded workgroup counts, bindings in WGSL @binding order.  All 22 stages are implemented (fine_msaa8/16 included).
 * Several stages pick a specialised instantiation from the HOST SHADOW of the ConfigUniform that was uploaded to the
 * buffer bound at index 0 (n_clip == 0: no clip stack) and from what is bound (no ramp / no non-zero image: no
 * gradient code); a config that only exists on the device selects the general instantiation. */
int jh_dispatch(jh_ctx* ctx, int stage, uint32_t gx, uint32_t gy, uint32_t gz, const jh_binding* bindings, int n_bindings);
int jh_dispatch_indirect(jh_ctx* ctx, int stage, uint64_t indirect_buffer_id, uint64_t offset, const jh_binding* bindings,
                         int n_bindings);

/* ---- hipGraph capture of a replayed recording (a frame is ~35 short launches: launch-bound on the host) ----
 * jh_graph_begin starts capturing everything enqueued on the context's stream (dispatches, clears);
 * jh_graph_end returns an executable graph; jh_graph_launch replays it.  Uploads/downloads/frees and
 * profiling must not be issued while capturing, and every buffer and scratch array must already exist
 * (run the recording once eagerly first).  A graph bakes in device pointers and the kernel instantiations chosen at
 * capture time: it stays valid only until a buffer or image it uses is freed, regrown or re-imported, or an eager run
 * makes a scratch array grow.  jh_graph_launch detects this (a generation counter) and returns JH_ERR_INVALID.
 * Two internal counters (flatten's work lists, backdrop's wide-row list) are zeroed by kernels of the frame itself
 * instead of by fill launches, so a captured frame contains no fill for them: it assumes, like every frame, that the
 * frame before it ran to its end.  A stage that fails half-way leaves a host-side flag down: the next eager stage fills
 * again, and jh_graph_launch zeroes the counters its graph has no fill for before it replays (counted:
 * jh_debug_graph_self_cleans). */
int jh_graph_begin(jh_ctx* ctx);
int jh_graph_end(jh_ctx* ctx, void** graph_exec);
int jh_graph_launch(jh_ctx* ctx, void* graph_exec);
/* What the capture recorded: kernel launches and other nodes (fills, copies) of one replay -- the launches per frame. */
int jh_graph_node_counts(jh_ctx* ctx, void* graph_exec, uint32_t* kernel_nodes, uint32_t* other_nodes);
int jh_graph_destroy(jh_ctx* ctx, void* graph_exec);

/* ---- profiling ---- */
int jh_profile_enable(jh_ctx* ctx, int on);
/* Waits for the device, writes up to max records (one per dispatch since the last collect), returns the count. */
int jh_profile_collect(jh_ctx* ctx, jh_profile_record* out, int max);
/* Nested spans as in the reference's profiler: group_begin opens a group under the innermost open one (Profiler.Start at
 * top level, ProfilerGroup.Nest below), group_end closes it; every dispatch issued in between becomes a query of that
 * group.  No-ops while profiling is disabled (the reference's nil profiler).  collect_tree = Profiler.Collect: waits for
 * the device and returns everything since the last collect as a flattened tree (and clears it, like jh_profile_collect). */
int jh_profile_group_begin(jh_ctx* ctx, const char* label);
int jh_profile_group_end(jh_ctx* ctx);
int jh_profile_collect_tree(jh_ctx* ctx, jh_profile_node* out, int max);

/* ---- diagnostics (not used by the render path) ----
 * Evaluates one of the kernels' scalar math routines on n host floats: op 0 sin, 1 cos, 2 atan2(a,b),
 * 3 acos, 4 asin, 5 |a|^(2/3), 6 a/b, 7 sqrt, 8 round-to-even, 9 u32(a), 10 i32(a), 11 f32->f16 bits,
 * 12 a*b+a (uncontracted), 13 floor(a*b+0.5), 14 min(a,b), 15 max(a,b), 16 clamp(a,0,1), 17 clamp(a*b,0,1). */
int jh_selftest_math(jh_ctx* ctx, int op, const float* a, const float* b, float* out, uint32_t n);
/* Runs the allocation patterns the kernels rely on -- a returning atomic add with a different value per lane on one address, inside
 * a loop that lanes skip and leave at different trips -- on n_waves waves (1..4096) and compares with a serial execution on the
 * host: form 0 the plain per-lane atomic (what the compiler's atomic optimizer makes of it), 1 the hand-aggregated wave_bump
 * (kcommon.h) the hot call sites use, 2 the wave-private LDS forms (XOR into one word, 64-bit OR into a word per lane).
 * Returns 0 when everything agrees, the number of violations (> 0) otherwise, a negative jh_status on misuse / device errors.
 * (Round 5 met a compiler strategy that returned overlapping ranges for form 0; tests/test_gpu_math.py runs all three.) */
int jh_selftest_atomics(jh_ctx* ctx, int form, uint32_t seed, uint32_t n_waves);
/* Fills every per-context scratch allocation (the count / offset arrays and counters of the deterministic allocators)
 * with `byte` and forgets that any counter was left clean -- the state of freshly allocated device memory that happens
 * not to be zero.  Tests use it to show that no stage relies on what an earlier frame (or hipMalloc) left behind, the
 * way the reference's pooled buffers hold stale data (engine/wgpu_engine/wgpu.go:772-808). */
uint64_t jh_debug_scratch_bytes(jh_ctx* ctx, int slot);   /* capacity of an internal scratch array (tests); slot -1: all of them */
/* Internal scratch arrays (count / offset arrays of the deterministic allocators, flatten's temporary) grow on demand and are
   kept.  jh_scratch_trim waits for the stream and frees them all: after one frame that was much larger than the ones to come,
   or after a first attempt with the estimator's generous bump sizes (the reference's pool keeps its buffers in the same way,
   wgpu.go:601-616; this is the counterpart of dropping that pool).  Captured graphs become stale. */
int jh_scratch_trim(jh_ctx* ctx);
/* tests: bit 0 = every wave of flatten starts in region 0 of its temporary, bit 1 = always eight regions (kernels_flatten.hip,
   FlTemp): ordinary scenes then fill regions up and move on, which the product only does close to the line buffer's capacity;
   bit 2 = a batch of more than 48 lines allocates its slots job by job (the product: more than 51 200) */
int jh_debug_flatten_regions(jh_ctx* ctx, uint32_t flags);
uint64_t jh_debug_graph_self_cleans(jh_ctx* ctx);  /* replays that had to zero an internal counter first (tests) */
int jh_debug_poison_scratch(jh_ctx* ctx, int byte);

/* ---- introspection ---- */
int jh_device_info(jh_ctx* ctx, char* name, int name_len, int* compute_units, uint64_t* total_mem);
uint64_t jh_pool_bytes(jh_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* JELLO_HIP_H */
