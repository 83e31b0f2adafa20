/* jello_formats.h -- byte layouts that cross the host <-> device boundary.
 *
 * Every struct here mirrors a `structs.HostLayout` Go type of the reference and the
 * WGSL struct it is bound to.  All little-endian, 4-byte aligned, no implicit padding.
 * Reference: renderer/config.go:25-80 (ConfigUniform, Layout), renderer/path.go:15-111,
 * renderer/draw.go:13-52, renderer/clip.go:9-53, renderer/config.go:301-312,365-372;
 * WGSL: engine/wgpu_engine/shaders/original/shared/{config,pathtag,bbox,segment,tile,
 * drawtag,clip,bump}.wgsl.
 */
#ifndef JELLO_FORMATS_H
#define JELLO_FORMATS_H

#include <stdint.h>

#ifdef __cplusplus
#define JL_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define JL_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* renderer/config.go:57-80 */
typedef struct JlLayout {
    uint32_t n_drawobj;      /* NumDrawObjects */
    uint32_t n_path;         /* NumPaths */
    uint32_t n_clip;         /* NumClips */
    uint32_t bin_data_start; /* BinDataStart: info words precede bin data */
    uint32_t pathtag_base;   /* all *_base are u32-word offsets into the scene */
    uint32_t pathdata_base;
    uint32_t drawtag_base;
    uint32_t drawdata_base;
    uint32_t transform_base;
    uint32_t style_base;
} JlLayout;

/* renderer/config.go:25-55, shared/config.wgsl:5-42 (100 bytes) */
typedef struct JlConfig {
    uint32_t width_in_tiles;
    uint32_t height_in_tiles;
    uint32_t target_width;
    uint32_t target_height;
    float base_color[4];
    JlLayout layout;
    uint32_t lines_size;
    uint32_t binning_size;
    uint32_t tiles_size;
    uint32_t seg_counts_size;
    uint32_t segments_size;
    uint32_t blend_size;
    uint32_t ptcl_size;
} JlConfig;
JL_STATIC_ASSERT(sizeof(JlConfig) == 100, "ConfigUniform is 100 bytes");

/* renderer/path.go:15-28, shared/pathtag.wgsl:4-11 */
typedef struct JlTagMonoid {
    uint32_t trans_ix;
    uint32_t pathseg_ix;
    uint32_t pathseg_offset;
    uint32_t style_ix;
    uint32_t path_ix;
} JlTagMonoid;
JL_STATIC_ASSERT(sizeof(JlTagMonoid) == 20, "TagMonoid");

/* renderer/path.go:56-71, shared/bbox.wgsl:12-19 */
typedef struct JlPathBbox {
    int32_t x0, y0, x1, y1;
    uint32_t draw_flags;
    uint32_t trans_ix;
} JlPathBbox;
JL_STATIC_ASSERT(sizeof(JlPathBbox) == 24, "PathBbox");

/* renderer/path.go:81-88, shared/segment.wgsl:19-25 */
typedef struct JlLineSoup {
    uint32_t path_ix;
    uint32_t pad;
    float p0[2];
    float p1[2];
} JlLineSoup;
JL_STATIC_ASSERT(sizeof(JlLineSoup) == 24, "LineSoup");

/* renderer/path.go:106-111, shared/segment.wgsl:28-35 */
typedef struct JlSegmentCount {
    uint32_t line_ix;
    uint32_t counts; /* seg_within_slice << 16 | seg_within_line */
} JlSegmentCount;
JL_STATIC_ASSERT(sizeof(JlSegmentCount) == 8, "SegmentCount");

/* renderer/path.go:90-97, shared/segment.wgsl:5-10 */
typedef struct JlSegment {
    float p0[2];
    float p1[2];
    float y_edge;
    uint32_t pad;
} JlSegment;
JL_STATIC_ASSERT(sizeof(JlSegment) == 24, "Segment");

/* renderer/path.go:73-79, shared/tile.wgsl:6-11 */
typedef struct JlPath {
    uint32_t bbox[4]; /* x0,y0,x1,y1 in tiles */
    uint32_t tiles;   /* offset into the tiles buffer */
    uint32_t pad[3];
} JlPath;
JL_STATIC_ASSERT(sizeof(JlPath) == 32, "Path");

/* renderer/path.go:99-104, shared/tile.wgsl:13-21 */
typedef struct JlTile {
    int32_t backdrop;
    uint32_t segment_count_or_ix; /* count until coarse, ~seg_ix afterwards */
} JlTile;
JL_STATIC_ASSERT(sizeof(JlTile) == 8, "Tile");

/* renderer/draw.go:13-24, shared/drawtag.wgsl:6-15 */
typedef struct JlDrawMonoid {
    uint32_t path_ix;
    uint32_t clip_ix;
    uint32_t scene_offset;
    uint32_t info_offset;
} JlDrawMonoid;
JL_STATIC_ASSERT(sizeof(JlDrawMonoid) == 16, "DrawMonoid");

/* renderer/clip.go:9-18, shared/clip.wgsl:14-21 */
typedef struct JlClipInp {
    uint32_t ix;
    int32_t path_ix; /* >=0: BeginClip path; <0: ~drawobj_ix of the EndClip */
} JlClipInp;
JL_STATIC_ASSERT(sizeof(JlClipInp) == 8, "ClipInp");

/* renderer/clip.go:26-38, shared/clip.wgsl:4-12 */
typedef struct JlClipBic {
    uint32_t a; /* pops */
    uint32_t b; /* pushes */
} JlClipBic;

/* renderer/clip.go:40-46, shared/clip.wgsl:23-26 */
typedef struct JlClipEl {
    uint32_t parent_ix;
    uint32_t pad[3];
    float bbox[4];
} JlClipEl;
JL_STATIC_ASSERT(sizeof(JlClipEl) == 32, "ClipEl");

/* renderer/clip.go:48-53, binning.wgsl:33-36 */
typedef struct JlBinHeader {
    uint32_t element_count;
    uint32_t chunk_offset;
} JlBinHeader;

/* renderer/config.go:301-312, shared/bump.wgsl:12-22 */
typedef struct JlBump {
    uint32_t failed;
    uint32_t binning;
    uint32_t ptcl;
    uint32_t tile;
    uint32_t seg_counts;
    uint32_t segments;
    uint32_t blend;
    uint32_t lines;
} JlBump;
JL_STATIC_ASSERT(sizeof(JlBump) == 32, "BumpAllocators");

/* renderer/config.go:365-372 */
typedef struct JlIndirectCount {
    uint32_t x, y, z, pad;
} JlIndirectCount;

/* shared/bump.wgsl:5-9 */
enum {
    JL_STAGE_BINNING = 0x1,
    JL_STAGE_TILE_ALLOC = 0x2,
    JL_STAGE_FLATTEN = 0x4,
    JL_STAGE_PATH_COUNT = 0x8,
    JL_STAGE_COARSE = 0x10
};

/* encoding/path.go:130-175 */
enum {
    JL_PATH_TAG_SEG_TYPE = 3,
    JL_PATH_TAG_LINETO = 1,
    JL_PATH_TAG_QUADTO = 2,
    JL_PATH_TAG_CUBICTO = 3,
    JL_PATH_TAG_SUBPATH_END = 4,
    JL_PATH_TAG_F32 = 8,
    JL_PATH_TAG_PATH = 0x10,
    JL_PATH_TAG_TRANSFORM = 0x20,
    JL_PATH_TAG_STYLE = 0x40
};

/* encoding/path.go:38-73, shared/pathtag.wgsl:24-40 */
#define JL_STYLE_FLAGS_STYLE 0x80000000u
#define JL_STYLE_FLAGS_FILL 0x40000000u
#define JL_STYLE_MITER_LIMIT_MASK 0xFFFFu
#define JL_STYLE_FLAGS_START_CAP_MASK 0x0C000000u
#define JL_STYLE_FLAGS_END_CAP_MASK 0x03000000u
#define JL_STYLE_FLAGS_CAP_BUTT 0u
#define JL_STYLE_FLAGS_CAP_SQUARE 0x01000000u
#define JL_STYLE_FLAGS_CAP_ROUND 0x02000000u
#define JL_STYLE_FLAGS_JOIN_MASK 0x30000000u
#define JL_STYLE_FLAGS_JOIN_BEVEL 0u
#define JL_STYLE_FLAGS_JOIN_MITER 0x10000000u
#define JL_STYLE_FLAGS_JOIN_ROUND 0x20000000u

/* encoding/draw.go:16-44, shared/drawtag.wgsl:19-26 */
enum {
    JL_DRAWTAG_NOP = 0,
    JL_DRAWTAG_FILL_COLOR = 0x50,
    JL_DRAWTAG_FILL_LIN_GRADIENT = 0x114,
    JL_DRAWTAG_FILL_RAD_GRADIENT = 0x29c,
    JL_DRAWTAG_FILL_SWEEP_GRADIENT = 0x254,
    JL_DRAWTAG_FILL_IMAGE = 0x248,
    JL_DRAWTAG_BEGIN_CLIP = 0x9,
    JL_DRAWTAG_END_CLIP = 0x21
};

/* shared/ptcl.wgsl:6-26 */
enum {
    JL_PTCL_INITIAL_ALLOC = 64,
    JL_PTCL_INCREMENT = 256,
    JL_PTCL_HEADROOM = 2,
    JL_CMD_END = 0,
    JL_CMD_FILL = 1,
    JL_CMD_STROKE = 2,
    JL_CMD_SOLID = 3,
    JL_CMD_COLOR = 5,
    JL_CMD_LIN_GRAD = 6,
    JL_CMD_RAD_GRAD = 7,
    JL_CMD_SWEEP_GRAD = 8,
    JL_CMD_IMAGE = 9,
    JL_CMD_BEGIN_CLIP = 10,
    JL_CMD_END_CLIP = 11,
    JL_CMD_JUMP = 12
};

/* shared/config.wgsl:46-73 */
enum {
    JL_TILE_WIDTH = 16,
    JL_TILE_HEIGHT = 16,
    JL_N_TILE_X = 16,
    JL_N_TILE_Y = 16,
    JL_N_TILE = 256,
    JL_BLEND_STACK_SPLIT = 4,
    JL_RAD_GRAD_KIND_CIRCULAR = 1,
    JL_RAD_GRAD_KIND_STRIP = 2,
    JL_RAD_GRAD_KIND_FOCAL_ON_CIRCLE = 3,
    JL_RAD_GRAD_KIND_CONE = 4,
    JL_RAD_GRAD_SWAPPED = 1,
    JL_GRADIENT_WIDTH = 512
};

/* renderer/recording.go:140-147 */
enum JlImageFormat { JL_RGBA8 = 0, JL_RGBA8_SRGB = 1, JL_BGRA8 = 2, JL_RGBA16_FLOAT = 3 };

#endif /* JELLO_FORMATS_H */
