// renderer.h -- the backend-agnostic recording layer: scene packing (Resolver), render
// configuration (ConfigUniform, workgroup counts, buffer sizes) and the fixed dispatch DAG
// (Renderer::render_full).  Mirrors renderer/{resolve,config,recording,render,ramp_cache}.go.
// The Recording produced here is the drop-in boundary: engine/hip_engine replays it
// over the C ABI of include/jello_hip.h exactly like engine/wgpu_engine.RunRecording replays the
// reference's.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "encoding.h"
#include "jello_formats.h"

namespace jello {

// ---- recording.go -------------------------------------------------------------------------
using ResourceID = uint64_t;
ResourceID next_resource_id();  // recording.go:15-19 (process-wide atomic counter)

struct BufferProxy { uint64_t size = 0; ResourceID id = 0; std::string name; };             // recording.go:133-137
struct ImageProxy { uint32_t width = 0, height = 0; JlImageFormat format = JL_RGBA8; ResourceID id = 0; };  // :149-154
BufferProxy new_buffer_proxy(uint64_t size, const std::string& name);
ImageProxy new_image_proxy(uint32_t width, uint32_t height, JlImageFormat format);

struct ResourceProxy {  // recording.go:31-36
    enum Kind { None = 0, Buffer = 1, Image = 2, ImageArray = 3 } kind = None;
    BufferProxy buffer;
    ImageProxy image;
    std::vector<ImageProxy> image_array;
    static ResourceProxy of(const BufferProxy& b) { ResourceProxy r; r.kind = Buffer; r.buffer = b; return r; }
    static ResourceProxy of(const ImageProxy& i) { ResourceProxy r; r.kind = Image; r.image = i; return r; }
};

using ShaderID = int;
using WorkgroupSize = uint32_t[3];

struct Command {  // recording.go:158-239
    enum Kind { Upload, UploadUniform, UploadImage, WriteImage, Dispatch, DispatchIndirect, Download, Clear, FreeBuffer, FreeImage } kind;
    BufferProxy buffer;              // Upload*/Download/Clear/FreeBuffer; the indirect buffer for DispatchIndirect
    ImageProxy image;                // UploadImage/FreeImage
    std::vector<uint8_t> data;       // Upload*/UploadImage payload (owned copy)
    ShaderID shader = -1;            // Dispatch*
    uint32_t wg_count[3] = {0, 0, 0};
    std::vector<ResourceProxy> bindings;
    uint64_t offset = 0;             // DispatchIndirect / Clear
    int64_t size = -1;               // Clear (-1 = whole buffer)
    uint32_t coords[4] = {0, 0, 0, 0};  // WriteImage: x, y, width, height (recording.go:204-208); texels in `data`
};

class Recording {  // recording.go:38-103
   public:
    std::vector<Command> commands;
    // Not in the reference's Recording: the deepest nesting of clip / blend layers in the encoding this recording renders
    // (0 = unknown or none).  engine/hip_engine passes it to jh_set_clip_depth_hint, which sizes fine's blend-stack scratch
    // (include/jello_hip.h); the Go shim counts it off encoding.DrawTags in RenderToTexture instead (integration/).
    uint32_t max_clip_depth = 0;
    BufferProxy upload(const std::string& name, const void* data, size_t n);
    BufferProxy upload_uniform(const std::string& name, const void* data, size_t n);
    ImageProxy upload_image(uint32_t w, uint32_t h, JlImageFormat format, const void* data, size_t n);
    void write_image(const ImageProxy& img, uint32_t x, uint32_t y, uint32_t w, uint32_t h, const void* data, size_t n);  // recording.go:66-72
    void dispatch(ShaderID shader, const uint32_t wg[3], std::vector<ResourceProxy> resources);
    void dispatch_indirect(ShaderID shader, const BufferProxy& buf, uint64_t offset, std::vector<ResourceProxy> resources);
    void download(const BufferProxy& buf);
    void clear_all(const BufferProxy& buf);
    void free_buffer(const BufferProxy& buf);
    void free_image(const ImageProxy& img);
    void free_resource(const ResourceProxy& r);
};

// ---- render.go:17-43 -------------------------------------------------------------------------
struct FullShaders {
    ShaderID pathtag_reduce = 0, pathtag_reduce2 = 1, pathtag_scan1 = 2, pathtag_scan_small = 3, pathtag_scan_large = 4,
             bbox_clear = 5, flatten = 6, draw_reduce = 7, draw_leaf = 8, clip_reduce = 9, clip_leaf = 10, binning = 11,
             tile_alloc = 12, backdrop_dyn = 13, path_count_setup = 14, path_count = 15, coarse = 16, path_tiling_setup = 17,
             path_tiling = 18, fine_area = 19, fine_msaa8 = 20, fine_msaa16 = 21;
    bool pathtag_is_cpu = false;
};

enum class AaConfig : int { Area = 0, Msaa8 = 1, Msaa16 = 2 };  // render.go:50-56

// Sizes of the bump-allocated buffers, in elements.  The reference hard-codes them
// (config.go:141-151); they are a host policy, so the engine may override them.
struct BumpSizes {
    uint32_t bin_data = 1u << 18, tiles = 1u << 21, lines = 1u << 21, seg_counts = 1u << 21, segments = 1u << 21,
             blend_spill = 1u << 21, ptcl = 1u << 23;
};

struct RenderParams {  // render.go:58-63
    Color base_color;
    uint32_t width = 0, height = 0;
    AaConfig antialiasing_method = AaConfig::Area;
    BumpSizes bump_sizes;  // extension: defaults = the reference's constants
};

// ---- config.go ----------------------------------------------------------------------------------
struct WorkgroupCounts {  // config.go:275-298
    bool use_large_path_scan = false;
    uint32_t path_reduce[3], path_reduce2[3], path_scan1[3], path_scan[3], bbox_clear[3], flatten[3], draw_reduce[3], draw_leaf[3],
        clip_reduce[3], clip_leaf[3], binning[3], tile_alloc[3], path_count_setup[3], backdrop[3], coarse[3], path_tiling_setup[3], fine[3];
};
struct BufferSizes {  // config.go:245-273, all in BYTES here (sizeInBytes applied)
    uint64_t path_reduced, path_reduced2, path_reduced_scan, path_monoids, path_bboxes, draw_reduced, draw_monoids, info, clip_inps,
        clip_els, clip_bics, clip_bboxes, draw_bboxes, bump_alloc, indirect_count, bin_headers, paths, lines, bin_data, tiles, seg_counts,
        segments, blend_spill, ptcl;
};
struct RenderConfig {  // config.go:88-123
    JlConfig gpu;
    WorkgroupCounts workgroup_counts;
    BufferSizes buffer_sizes;
};
WorkgroupCounts new_workgroup_counts(const JlLayout& layout, uint32_t width_in_tiles, uint32_t height_in_tiles, uint32_t num_path_tags);
BufferSizes new_buffer_sizes(const JlLayout& layout, const WorkgroupCounts& wg, const BumpSizes& bump);
RenderConfig new_render_config(const JlLayout& layout, uint32_t width, uint32_t height, const Color& base_color, const BumpSizes& bump);

// ---- ramp_cache.go / resolve.go --------------------------------------------------------------------
struct Ramps { std::vector<uint16_t> data; uint32_t width = 0, height = 0; };  // RGBA16F texels, 4 x u16 each

class Resolver {  // resolve.go:20-33
   public:
    struct Resolved { JlLayout layout; Ramps ramps; std::vector<Image> images; std::vector<uint8_t> packed; };
    Resolved resolve(const Encoding& enc);

   private:
    struct RampEntry { uint32_t id; uint64_t epoch; };
    uint32_t ramp_add(const ColorStop* stops, size_t n);
    void ramp_maintain();
    uint64_t epoch_ = 0;
    std::map<std::string, RampEntry> mapping_;
    std::vector<uint16_t> ramp_data_;
};

// ---- render.go:81-588 ------------------------------------------------------------------------------
class Renderer {
   public:
    struct Result { Recording recording; ResourceProxy out_image; RenderConfig config; };
    Result render_full(const Encoding& enc, Resolver& resolver, const FullShaders& shaders, const RenderParams& params, bool robust = false);
    // Proxies of the last recording, for engines that want to inspect intermediates (tests).
    std::map<std::string, BufferProxy> last_buffers;

   private:
    ResourceProxy mask_buf_;
    AaConfig mask_aa_ = AaConfig::Area;
    ImageProxy empty_;
    std::map<uint64_t, ImageProxy> images_;
};

// mask.go:43-105 -- MSAA sample mask LUTs (used by fine_msaa8/16)
std::vector<uint8_t> make_mask_lut8();
std::vector<uint8_t> make_mask_lut16();

}  // namespace jello
