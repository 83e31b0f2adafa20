// Package hip_engine replays renderer.Recordings on an AMD MI355X (gfx950) through libjello_hip.so.
//
// It is the drop-in replacement for engine/wgpu_engine on the compute path: Scene, encoding and
// renderer stay as they are; this package walks the Recording exactly like
// wgpu_engine.Engine.RunRecording (engine/wgpu_engine/wgpu.go:322-643) and forwards every command to
// the C ABI of include/jello_hip.h.  Drop this directory into the reference tree as
// engine/hip_engine/ (adjust the two #cgo paths) and apply integration/renderer_bump_sizes.patch.
//
// NOT COMPILED in the repository it comes from (the build image has no Go toolchain); the C++ twin
// jello_amd/host/hip_engine.cpp makes the same calls in the same order and is what the tests run.
package hip_engine

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../jello_amd -ljello_hip
#include <stdlib.h>
#include "jello_hip.h"
*/
import "C"

import (
	"encoding/binary"
	"fmt"
	"image"
	"runtime"
	"unsafe"

	"honnef.co/go/jello/encoding"
	"honnef.co/go/jello/mem"
	"honnef.co/go/jello/profiler"
	"honnef.co/go/jello/renderer"
)

type Engine struct {
	ctx         *C.jh_ctx
	renderer    *renderer.Renderer
	resolver    *renderer.Resolver
	fullShaders *renderer.FullShaders
	downloads   map[renderer.ResourceID][]byte
}

// New mirrors wgpu_engine.New (wgpu.go:157-178).  device is the HIP device ordinal; one Engine per
// GPU, engines on different GPUs are independent.
func New(device int) (*Engine, error) {
	var ctx *C.jh_ctx
	if rc := C.jh_create(&ctx, C.int(device)); rc != C.JH_OK {
		return nil, fmt.Errorf("jh_create: %d", int(rc))
	}
	// ShaderIDs are the jh_stage values, i.e. the FullShaders field order (render.go:17-43).
	fs := &renderer.FullShaders{
		PathtagReduce: C.JH_PATHTAG_REDUCE, PathtagReduce2: C.JH_PATHTAG_REDUCE2, PathtagScan1: C.JH_PATHTAG_SCAN1,
		PathtagScanSmall: C.JH_PATHTAG_SCAN_SMALL, PathtagScanLarge: C.JH_PATHTAG_SCAN_LARGE, BboxClear: C.JH_BBOX_CLEAR,
		Flatten: C.JH_FLATTEN, DrawReduce: C.JH_DRAW_REDUCE, DrawLeaf: C.JH_DRAW_LEAF, ClipReduce: C.JH_CLIP_REDUCE,
		ClipLeaf: C.JH_CLIP_LEAF, Binning: C.JH_BINNING, TileAlloc: C.JH_TILE_ALLOC, BackdropDyn: C.JH_BACKDROP_DYN,
		PathCountSetup: C.JH_PATH_COUNT_SETUP, PathCount: C.JH_PATH_COUNT, Coarse: C.JH_COARSE,
		PathTilingSetup: C.JH_PATH_TILING_SETUP, PathTiling: C.JH_PATH_TILING, FineArea: C.JH_FINE_AREA,
		FineMSAA8: C.JH_FINE_MSAA8, FineMSAA16: C.JH_FINE_MSAA16,
		// PathtagIsCPU stays false: the three-level scan stages exist on the HIP side.
	}
	return &Engine{ctx: ctx, renderer: renderer.New(), resolver: renderer.NewResolver(), fullShaders: fs,
		downloads: map[renderer.ResourceID][]byte{}}, nil
}

func (e *Engine) Close() { C.jh_destroy(e.ctx) }

// The reference panics on every misuse (wgpu.go:77,213,282,544,558,594,955); the C ABI returns
// codes, which become panics here to keep the calling convention of wgpu_engine.
func (e *Engine) check(rc C.int, what string) {
	if rc != C.JH_OK {
		panic(fmt.Sprintf("hip_engine: %s: %s", what, C.GoString(C.jh_last_error(e.ctx))))
	}
}

// ExternalImage hands the engine a caller-owned device allocation for an ImageProxy (wgpu.go:90-93,
// lib.go:257-262): width*height*8 bytes of device memory for the RGBA16F target.
type ExternalImage struct {
	Proxy     renderer.ImageProxy
	DevicePtr unsafe.Pointer
}

// RunRecording mirrors wgpu.go:322-643.  Frees are deferred to the end of the recording
// (wgpu.go:601-616); buffers the recording never frees stay resident (wgpu.go:631-640).
func (e *Engine) RunRecording(rec renderer.Recording, external []ExternalImage, pgroup string) {
	label := C.CString(pgroup) // pgroup = pgroup.Nest("RunRecording"), wgpu.go:330: a no-op unless jh_profile_enable(1)
	defer C.free(unsafe.Pointer(label))
	e.check(C.jh_profile_group_begin(e.ctx, label), "profile_group_begin")
	defer C.jh_profile_group_end(e.ctx)

	for _, x := range external {
		e.check(C.jh_image_import(e.ctx, C.uint64_t(x.Proxy.ID), x.DevicePtr, C.uint32_t(x.Proxy.Width),
			C.uint32_t(x.Proxy.Height), C.int(x.Proxy.Format)), "image_import")
	}
	var freeBufs, freeImages []renderer.ResourceID
	pendingClears := map[renderer.ResourceID]bool{}
	for _, cmd := range rec.Commands {
		switch cmd := cmd.(type) {
		case *renderer.Upload:
			e.upload(cmd.Buffer.ID, cmd.Data)
		case *renderer.UploadUniform:
			e.upload(cmd.Buffer.ID, cmd.Data)
		case *renderer.UploadImage:
			p := cmd.Proxy
			e.check(C.jh_image_upload(e.ctx, C.uint64_t(p.ID), C.uint32_t(p.Width), C.uint32_t(p.Height), C.int(p.Format),
				unsafe.Pointer(unsafe.SliceData(cmd.Data)), C.uint64_t(len(cmd.Data))), "image_upload")
		case *renderer.WriteImage: // wgpu.go:422-452
			p := cmd.Proxy
			if C.jh_image_device_ptr(e.ctx, C.uint64_t(p.ID)) == nil {
				e.check(C.jh_image_create(e.ctx, C.uint64_t(p.ID), C.uint32_t(p.Width), C.uint32_t(p.Height), C.int(p.Format)), "image_create")
			}
			data := imageData(cmd.Image)
			e.check(C.jh_image_write(e.ctx, C.uint64_t(p.ID), C.uint32_t(cmd.Coords[0]), C.uint32_t(cmd.Coords[1]),
				C.uint32_t(cmd.Coords[2]), C.uint32_t(cmd.Coords[3]), unsafe.Pointer(unsafe.SliceData(data)), C.uint64_t(len(data))), "image_write")
		case *renderer.Dispatch:
			b, pin := e.bind(cmd.Bindings, pendingClears)
			e.check(C.jh_dispatch(e.ctx, C.int(cmd.Shader), C.uint32_t(cmd.WorkgroupSize[0]), C.uint32_t(cmd.WorkgroupSize[1]),
				C.uint32_t(cmd.WorkgroupSize[2]), unsafe.SliceData(b), C.int(len(b))), "dispatch")
			pin.Unpin()
		case *renderer.DispatchIndirect:
			b, pin := e.bind(cmd.Bindings, pendingClears)
			e.check(C.jh_dispatch_indirect(e.ctx, C.int(cmd.Shader), C.uint64_t(cmd.Buffer.ID), C.uint64_t(cmd.Offset),
				unsafe.SliceData(b), C.int(len(b))), "dispatch_indirect")
			pin.Unpin()
		case *renderer.Download: // wgpu.go:554-563, 645-657
			dst := make([]byte, cmd.Buffer.Size)
			e.check(C.jh_download(e.ctx, C.uint64_t(cmd.Buffer.ID), unsafe.Pointer(unsafe.SliceData(dst)), 0,
				C.uint64_t(len(dst))), "download")
			e.downloads[cmd.Buffer.ID] = dst
		case *renderer.Clear:
			if C.jh_buffer_device_ptr(e.ctx, C.uint64_t(cmd.Buffer.ID)) != nil {
				e.check(C.jh_clear(e.ctx, C.uint64_t(cmd.Buffer.ID), C.uint64_t(cmd.Offset), C.int64_t(cmd.Size)), "clear")
			} else {
				pendingClears[cmd.Buffer.ID] = true // wgpu.go:583-585: cleared when the buffer is first bound
			}
		case *renderer.FreeBuffer:
			freeBufs = append(freeBufs, cmd.Buffer.ID)
		case *renderer.FreeImage:
			freeImages = append(freeImages, cmd.Image.ID)
		default:
			panic(fmt.Sprintf("unhandled command %T", cmd))
		}
	}
	for _, id := range freeBufs {
		C.jh_free(e.ctx, C.uint64_t(id))
	}
	for _, id := range freeImages {
		C.jh_image_free(e.ctx, C.uint64_t(id))
	}
}

// The slice is only read during the call: jh_upload copies it into pinned staging memory
// (queue.WriteBuffer semantics, wgpu.go:360).
func (e *Engine) upload(id renderer.ResourceID, data []byte) {
	e.check(C.jh_upload(e.ctx, C.uint64_t(id), unsafe.Pointer(unsafe.SliceData(data)), C.uint64_t(len(data))), "upload")
}

// bind converts []ResourceProxy into []C.jh_binding; transient buffers and images are materialised
// on first use (wgpu.go:877-925).  A runtime.Pinner keeps the image-array id slices alive for the call.
func (e *Engine) bind(res []renderer.ResourceProxy, pendingClears map[renderer.ResourceID]bool) ([]C.jh_binding, *runtime.Pinner) {
	pin := new(runtime.Pinner)
	out := make([]C.jh_binding, len(res))
	for i, r := range res {
		switch r.Kind {
		case renderer.ResourceProxyKindBuffer:
			id := C.uint64_t(r.BufferProxy.ID)
			if C.jh_buffer_device_ptr(e.ctx, id) == nil {
				e.check(C.jh_buffer_create(e.ctx, id, C.uint64_t(r.BufferProxy.Size)), "buffer_create")
				if pendingClears[r.BufferProxy.ID] {
					e.check(C.jh_clear(e.ctx, id, 0, -1), "clear")
					delete(pendingClears, r.BufferProxy.ID)
				}
			}
			out[i] = C.jh_binding{kind: C.JH_BIND_BUFFER, id: id}
		case renderer.ResourceProxyKindImage:
			p := r.ImageProxy
			if C.jh_image_device_ptr(e.ctx, C.uint64_t(p.ID)) == nil {
				e.check(C.jh_image_create(e.ctx, C.uint64_t(p.ID), C.uint32_t(p.Width), C.uint32_t(p.Height), C.int(p.Format)), "image_create")
			}
			out[i] = C.jh_binding{kind: C.JH_BIND_IMAGE, id: C.uint64_t(p.ID)}
		case renderer.ResourceProxyKindImageArray:
			ids := make([]C.uint64_t, len(r.ImageArray))
			for k, p := range r.ImageArray {
				if C.jh_image_device_ptr(e.ctx, C.uint64_t(p.ID)) == nil {
					e.check(C.jh_image_create(e.ctx, C.uint64_t(p.ID), C.uint32_t(p.Width), C.uint32_t(p.Height), C.int(p.Format)), "image_create")
				}
				ids[k] = C.uint64_t(p.ID)
			}
			if len(ids) > 0 {
				pin.Pin(unsafe.SliceData(ids))
			}
			out[i] = C.jh_binding{kind: C.JH_BIND_IMAGE_ARRAY, count: C.uint32_t(len(ids)), ids: unsafe.SliceData(ids)}
		}
	}
	return out, pin
}

// imageData as in wgpu.go:297-320.
func imageData(img image.Image) []byte {
	switch img := img.(type) {
	case *image.NRGBA:
		if img.Stride != 4*img.Rect.Dx() {
			panic("subimages are not supported")
		}
		return img.Pix
	case *image.RGBA:
		if img.Stride != 4*img.Rect.Dx() {
			panic("subimages are not supported")
		}
		return img.Pix
	default:
		panic(fmt.Sprintf("unsupported image type %T", img))
	}
}

// grow is the policy of jello_amd/host/hip_engine.cpp: the reported need plus a quarter.
func grow(have, need uint32) uint32 {
	if need <= have {
		return have
	}
	g := uint64(need) + uint64(need)/4 + 1024
	if g > 0xffffffff {
		g = 0xffffffff
	}
	return uint32(g)
}

// maxClipDepth is the deepest nesting of BeginClip ... EndClip in the encoding's draw tag stream.
func maxClipDepth(enc *encoding.Encoding) uint32 {
	var depth, deepest uint32
	for _, tag := range enc.DrawTags {
		switch tag {
		case encoding.DrawTagBeginClip:
			depth++
			if depth > deepest {
				deepest = depth
			}
		case encoding.DrawTagEndClip:
			if depth > 0 {
				depth--
			}
		}
	}
	return deepest
}

// RenderToTexture mirrors lib.go:244-264 and adds what the fixed sizes of renderer/config.go:141-151
// need for scenes beyond the Vello test scenes: the recording is made with robust = true (it then
// downloads BumpAllocators, render.go:458-460); if bump.Failed is set, the bump-allocated buffers are
// grown to the reported need and the frame is rendered again (Vello's regrow loop).  params.BumpSizes
// (integration/renderer_bump_sizes.patch) carries the sizes; start it from Scene.bumpEstimate
// (scene.go:36-43) -- see DESIGN.md section 1 for the two defects of renderer/estimate.go that make its
// result unusable as it stands -- or from the reference's constants.
// target is device memory for Width*Height RGBA16F pixels.  Returns the attempts it took.
func (e *Engine) RenderToTexture(arena *mem.Arena, enc *encoding.Encoding, target unsafe.Pointer,
	params *renderer.RenderParams, pgroup profiler.ProfilerGroup) int {
	for attempt := 1; ; attempt++ {
		var render renderer.Render
		recording := e.renderer.RenderEncodingCoarse(arena, &render, enc, e.resolver, e.fullShaders, params, true, pgroup)
		out := render.OutImage()
		bumpID := render.BumpBuf().ID // (accessor added by the patch; the proxy of the "bumpBuf" buffer)
		recording = e.renderer.RecordFine(arena, &render, e.fullShaders, recording, pgroup)
		// fine's blend-stack scratch is sized from the nesting depth of the clip layers (include/jello_hip.h,
		// jh_set_clip_depth_hint); the hint is taken back after the run so that it never outlives its scene
		// (taken back by a deferred call: a panic inside RunRecording must not leave a stale hint on the context -- a hint
		// that is too small for a later scene loses colours without an error; jh_debug_clip_hint_overflows would tell)
		func() {
			e.check(C.jh_set_clip_depth_hint(e.ctx, C.uint32_t(maxClipDepth(enc))), "set_clip_depth_hint")
			defer C.jh_set_clip_depth_hint(e.ctx, 0)
			e.RunRecording(recording, []ExternalImage{{Proxy: out, DevicePtr: target}}, "RunRecording")
		}()
		raw := e.downloads[bumpID]
		if len(raw) < 32 || attempt >= 6 {
			e.check(C.jh_sync(e.ctx), "sync")
			return attempt
		}
		var b renderer.BumpAllocators // Failed Binning Ptcl Tile SegCounts Segments Blend Lines (config.go:301-312)
		b.Failed = binary.LittleEndian.Uint32(raw[0:])
		b.Binning = binary.LittleEndian.Uint32(raw[4:])
		b.Ptcl = binary.LittleEndian.Uint32(raw[8:])
		b.Tile = binary.LittleEndian.Uint32(raw[12:])
		b.SegCounts = binary.LittleEndian.Uint32(raw[16:])
		b.Segments = binary.LittleEndian.Uint32(raw[20:])
		b.Blend = binary.LittleEndian.Uint32(raw[24:])
		b.Lines = binary.LittleEndian.Uint32(raw[28:])
		if b.Failed == 0 {
			e.check(C.jh_sync(e.ctx), "sync")
			return attempt
		}
		s := params.BumpSizes
		before := *s
		widthInTiles, heightInTiles := (params.Width+15)/16, (params.Height+15)/16
		s.Lines = grow(s.Lines, b.Lines)
		s.BinData = grow(s.BinData, b.Binning+render.Layout().BinDataStart)
		s.Tiles = grow(s.Tiles, b.Tile)
		s.SegCounts = grow(s.SegCounts, b.SegCounts)
		s.Segments = grow(s.Segments, max(b.Segments, b.SegCounts))
		s.BlendSpill = grow(s.BlendSpill, b.Blend)
		s.Ptcl = grow(s.Ptcl, b.Ptcl+widthInTiles*heightInTiles*64)
		if before == *s {
			return attempt // nothing left to grow
		}
	}
}

// TrimScratch gives the context's internal scratch arrays back (the count / offset arrays of the deterministic allocators,
// flatten's temporary: they grow on demand and are kept).  For the frame after one that was much larger -- or after a first
// frame that ran with far more generous BumpSizes than the scenes need.  Waits for the stream.
func (e *Engine) TrimScratch() {
	e.check(C.jh_scratch_trim(e.ctx), "scratch_trim")
}

// SetBand: ONE target over several GPUs -- this engine then writes the PTCL and rasterises only the
// 256-pixel bin rows [row0, row1) of the target, with every allocation offset identical to the
// unsharded run (record the same Recording on every GPU; nothing else changes).
func (e *Engine) SetBand(row0, row1 uint32) {
	e.check(C.jh_set_band(e.ctx, C.uint32_t(row0), C.uint32_t(row1)), "set_band")
}

// SelfTest runs the library's toolchain checks on the device: the allocation patterns the kernels rely on (jh_selftest_atomics,
// forms 0 plain per-lane atomic / 1 hand-aggregated / 2 wave-private LDS) against a serial execution.  A deployment that rebuilds
// libjello_hip.so with another ROCm can call it once after New; it returns an error naming the form that disagrees.
func (e *Engine) SelfTest() error {
	for form := 0; form < 3; form++ {
		if rc := C.jh_selftest_atomics(e.ctx, C.int(form), C.uint32_t(1+form), 1024); rc != 0 {
			return fmt.Errorf("hip_engine: jh_selftest_atomics form %d: %d", form, int(rc))
		}
	}
	return nil
}
