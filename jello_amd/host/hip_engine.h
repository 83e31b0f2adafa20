// hip_engine.h -- replays a renderer Recording on an MI355X through the C ABI of
// include/jello_hip.h.  Drop-in for engine/wgpu_engine: Engine::run_recording mirrors
// RunRecording (wgpu.go:322-643) command by command, Engine::render_to_texture mirrors
// RenderToTexture (lib.go:244-264).  A Go `engine/hip_engine` package does the same walk through
// cgo; this C++ twin exists because the build image has no Go toolchain.
#pragma once
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "jello_hip.h"
#include "renderer.h"

namespace jello {

struct EngineError : std::runtime_error {
    int code;
    EngineError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct ExternalImage { ImageProxy proxy; void* device_ptr; };   // wgpu.go:90-93
struct ExternalBuffer { BufferProxy proxy; void* device_ptr; }; // wgpu.go:85-88

enum RunFlags : unsigned {
    kRunUploads = 1,     // Upload / UploadUniform / UploadImage
    kRunDispatches = 2,  // Dispatch / DispatchIndirect / Clear / Download
    kRunFrees = 4,       // FreeBuffer / FreeImage (deferred to the end of the recording, wgpu.go:601-616)
    kRunSkipFine = 8,    // with kRunDispatches: every dispatch but the fine stage's (the last one of a RenderFull recording)
    kRunOnlyFine = 16,   // with kRunDispatches: the fine stage's dispatch alone -- the two let a caller put fine on a stream of its own
    kRunAll = 7
};

class Engine {
   public:
    explicit Engine(int device);
    ~Engine();
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    jh_ctx* ctx() { return ctx_; }
    FullShaders& shaders() { return shaders_; }

    // RunRecording.  Buffers that the recording never frees stay resident (wgpu.go:631-640).
    void run_recording(const Recording& rec, const std::vector<ExternalImage>& images = {}, const std::vector<ExternalBuffer>& buffers = {},
                       unsigned flags = kRunAll);

    // RenderToTexture: record + run.  If `out_device` is non-null it must point to width*height*8
    // bytes of device memory (RGBA16F) and receives the image; otherwise the engine owns the target
    // and `download_target` reads it back.  robust=true re-runs with grown bump buffers while
    // bump.failed is set (the regrow loop Vello has and Jello lacks).
    struct Frame {
        Recording recording;
        RenderConfig config;
        ImageProxy target;
        std::map<std::string, BufferProxy> buffers;
        JlBump bump{};     // what the frame's Download(bumpBuf) returned: only a robust recording has one (all zero otherwise)
        int attempts = 1;
    };
    Frame render_to_texture(const Encoding& enc, RenderParams params, void* out_device = nullptr, bool robust = false, bool retain = false);
    void download_target(const Frame& f, void* dst, size_t bytes);
    void release(const Frame& f);  // frees whatever a retain=true frame kept

    Resolver& resolver() { return resolver_; }
    Renderer& renderer() { return renderer_; }

   private:
    void check(int rc, const char* what);
    jh_ctx* ctx_ = nullptr;
    Renderer renderer_;
    Resolver resolver_;
    FullShaders shaders_;
    std::map<ResourceID, std::vector<uint8_t>> downloads_;

   public:
    const std::vector<uint8_t>* get_download(ResourceID id) const {
        auto it = downloads_.find(id);
        return it == downloads_.end() ? nullptr : &it->second;
    }
};

}  // namespace jello
