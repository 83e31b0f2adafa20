// hip_engine.cpp -- see hip_engine.h.
#include "hip_engine.h"

#include <algorithm>
#include <cstring>

namespace jello {

Engine::Engine(int device) {
    int rc = jh_create(&ctx_, device);
    if (rc != JH_OK) throw EngineError(rc, "jh_create failed (no MI355X / HIP runtime available?) rc=" + std::to_string(rc));
}

Engine::~Engine() { jh_destroy(ctx_); }

void Engine::check(int rc, const char* what) {
    if (rc != JH_OK) throw EngineError(rc, std::string(what) + ": " + jh_last_error(ctx_));
}

void Engine::run_recording(const Recording& rec, const std::vector<ExternalImage>& ext_images, const std::vector<ExternalBuffer>& ext_buffers,
                           unsigned flags) {
    // pgroup = pgroup.Nest("RunRecording"); defer pgroup.End()  (wgpu.go:330-331) -- a no-op unless profiling is on
    check(jh_profile_group_begin(ctx_, "RunRecording"), "profile_group_begin");
    struct GroupEnd { jh_ctx* c; ~GroupEnd() { (void)jh_profile_group_end(c); } } group_end{ctx_};
    // every recording brings its own bound (0 = unknown: fine reserves the worst case): a stale hint must never outlive its scene
    check(jh_set_clip_depth_hint(ctx_, rec.max_clip_depth), "set_clip_depth_hint");
    for (const ExternalImage& e : ext_images)
        check(jh_image_import(ctx_, e.proxy.id, e.device_ptr, e.proxy.width, e.proxy.height, (int)e.proxy.format), "image_import");
    for (const ExternalBuffer& e : ext_buffers) check(jh_buffer_import(ctx_, e.proxy.id, e.device_ptr, e.proxy.size), "buffer_import");
    std::vector<ResourceID> free_bufs, free_images;
    std::set<ResourceID> pending_clears;
    std::vector<jh_binding> bindings;
    std::vector<std::vector<uint64_t>> id_arrays;
    auto bind = [&](const std::vector<ResourceProxy>& res) {
        bindings.clear();
        id_arrays.clear();
        id_arrays.reserve(res.size());
        for (const ResourceProxy& r : res) {
            jh_binding b;
            std::memset(&b, 0, sizeof b);
            switch (r.kind) {
                case ResourceProxy::Buffer:
                    b.kind = JH_BIND_BUFFER;
                    b.id = r.buffer.id;
                    // transient buffers are materialised on first use (wgpu.go:877-925)
                    if (jh_buffer_size(ctx_, r.buffer.id) == 0 && jh_buffer_device_ptr(ctx_, r.buffer.id) == nullptr) {
                        check(jh_buffer_create(ctx_, r.buffer.id, r.buffer.size), "buffer_create");
                        if (pending_clears.erase(r.buffer.id)) check(jh_clear(ctx_, r.buffer.id, 0, -1), "clear");
                    }
                    break;
                case ResourceProxy::Image:
                    b.kind = JH_BIND_IMAGE;
                    b.id = r.image.id;
                    if (jh_image_device_ptr(ctx_, r.image.id) == nullptr)
                        check(jh_image_create(ctx_, r.image.id, r.image.width, r.image.height, (int)r.image.format), "image_create");
                    break;
                case ResourceProxy::ImageArray: {
                    b.kind = JH_BIND_IMAGE_ARRAY;
                    id_arrays.emplace_back();
                    for (const ImageProxy& ip : r.image_array) {
                        if (jh_image_device_ptr(ctx_, ip.id) == nullptr)
                            check(jh_image_create(ctx_, ip.id, ip.width, ip.height, (int)ip.format), "image_create");
                        id_arrays.back().push_back(ip.id);
                    }
                    b.count = (uint32_t)id_arrays.back().size();
                    b.ids = id_arrays.back().data();
                    break;
                }
                default: throw EngineError(JH_ERR_INVALID, "unhandled resource kind in binding");
            }
            bindings.push_back(b);
        }
    };
    for (const Command& cmd : rec.commands) {
        switch (cmd.kind) {
            case Command::Upload:
            case Command::UploadUniform:
                if (flags & kRunUploads) check(jh_upload(ctx_, cmd.buffer.id, cmd.data.data(), cmd.data.size()), "upload");
                break;
            case Command::UploadImage:
                if (flags & kRunUploads)
                    check(jh_image_upload(ctx_, cmd.image.id, cmd.image.width, cmd.image.height, (int)cmd.image.format, cmd.data.data(),
                                          cmd.data.size()),
                          "image_upload");
                break;
            case Command::WriteImage:  // wgpu.go:422-452 (RenderFull never records one; Recording::write_image does)
                if (flags & kRunUploads) {
                    if (jh_image_device_ptr(ctx_, cmd.image.id) == nullptr)
                        check(jh_image_create(ctx_, cmd.image.id, cmd.image.width, cmd.image.height, (int)cmd.image.format), "image_create");
                    check(jh_image_write(ctx_, cmd.image.id, cmd.coords[0], cmd.coords[1], cmd.coords[2], cmd.coords[3], cmd.data.data(),
                                         cmd.data.size()),
                          "image_write");
                }
                break;
            case Command::Dispatch:
                if (flags & kRunDispatches) {
                    const bool is_fine = cmd.shader == JH_FINE_AREA || cmd.shader == JH_FINE_MSAA8 || cmd.shader == JH_FINE_MSAA16;
                    if ((is_fine && (flags & kRunSkipFine)) || (!is_fine && (flags & kRunOnlyFine))) break;
                    bind(cmd.bindings);
                    check(jh_dispatch(ctx_, cmd.shader, cmd.wg_count[0], cmd.wg_count[1], cmd.wg_count[2], bindings.data(), (int)bindings.size()),
                          jh_stage_name(cmd.shader));
                }
                break;
            case Command::DispatchIndirect:
                if ((flags & kRunDispatches) && !(flags & kRunOnlyFine)) {
                    bind(cmd.bindings);
                    check(jh_dispatch_indirect(ctx_, cmd.shader, cmd.buffer.id, cmd.offset, bindings.data(), (int)bindings.size()),
                          jh_stage_name(cmd.shader));
                }
                break;
            case Command::Download:
                if ((flags & kRunDispatches) && !(flags & kRunOnlyFine)) {
                    std::vector<uint8_t>& dst = downloads_[cmd.buffer.id];
                    dst.resize(cmd.buffer.size);
                    check(jh_download(ctx_, cmd.buffer.id, dst.data(), 0, cmd.buffer.size), "download");
                }
                break;
            case Command::Clear:
                if ((flags & kRunDispatches) && !(flags & kRunOnlyFine)) {
                    if (jh_buffer_device_ptr(ctx_, cmd.buffer.id) != nullptr) {
                        check(jh_clear(ctx_, cmd.buffer.id, cmd.offset, cmd.size), "clear");
                    } else {
                        pending_clears.insert(cmd.buffer.id);  // wgpu.go:583-585
                    }
                }
                break;
            case Command::FreeBuffer: free_bufs.push_back(cmd.buffer.id); break;
            case Command::FreeImage: free_images.push_back(cmd.image.id); break;
        }
    }
    if (flags & kRunFrees) {
        for (ResourceID id : free_bufs) check(jh_free(ctx_, id), "free");
        for (ResourceID id : free_images) check(jh_image_free(ctx_, id), "image_free");
    }
}

static uint32_t grow(uint32_t have, uint32_t need) {
    if (need <= have) return have;
    uint64_t g = (uint64_t)need + need / 4 + 1024;
    return g > 0xffffffffull ? 0xffffffffu : (uint32_t)g;
}

Engine::Frame Engine::render_to_texture(const Encoding& enc, RenderParams params, void* out_device, bool robust, bool retain) {
    Frame f;
    for (int attempt = 1; attempt <= 6; attempt++) {
        Renderer::Result r = renderer_.render_full(enc, resolver_, shaders_, params, robust);
        f.recording = std::move(r.recording);
        f.config = r.config;
        f.target = r.out_image.image;
        f.buffers = renderer_.last_buffers;
        f.attempts = attempt;
        std::vector<ExternalImage> ext;
        if (out_device) ext.push_back(ExternalImage{f.target, out_device});
        run_recording(f.recording, ext, {}, retain ? (kRunUploads | kRunDispatches) : kRunAll);
        if (!robust) break;
        const std::vector<uint8_t>* d = get_download(f.buffers["bumpBuf"].id);
        if (!d || d->size() < sizeof(JlBump)) break;
        std::memcpy(&f.bump, d->data(), sizeof(JlBump));
        if (f.bump.failed == 0) break;
        // grow whatever overflowed; bump.* report the required element counts (failure detection)
        BumpSizes& bsz = params.bump_sizes;
        BumpSizes before = bsz;
        const JlConfig& g = f.config.gpu;
        bsz.lines = grow(bsz.lines, f.bump.lines);
        bsz.bin_data = grow(bsz.bin_data, f.bump.binning + g.layout.bin_data_start);
        bsz.tiles = grow(bsz.tiles, f.bump.tile);
        bsz.seg_counts = grow(bsz.seg_counts, f.bump.seg_counts);
        bsz.segments = grow(bsz.segments, std::max(f.bump.segments, f.bump.seg_counts));
        bsz.blend_spill = grow(bsz.blend_spill, f.bump.blend);
        bsz.ptcl = grow(bsz.ptcl, f.bump.ptcl + g.width_in_tiles * g.height_in_tiles * JL_PTCL_INITIAL_ALLOC);
        if (retain) release(f);
        if (std::memcmp(&before, &bsz, sizeof bsz) == 0) break;  // nothing left to grow: give up
    }
    return f;
}

void Engine::download_target(const Frame& f, void* dst, size_t bytes) { check(jh_image_download(ctx_, f.target.id, dst, bytes), "image_download"); }

void Engine::release(const Frame& f) {
    for (const Command& cmd : f.recording.commands) {
        if (cmd.kind == Command::FreeBuffer) check(jh_free(ctx_, cmd.buffer.id), "free");
        if (cmd.kind == Command::FreeImage) check(jh_image_free(ctx_, cmd.image.id), "image_free");
    }
}

}  // namespace jello
