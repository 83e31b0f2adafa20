"""Locating, building and loading the native libraries (ctypes).

The product path fails loudly if the HIP extension is missing: there is no CPU fallback anywhere in
this package.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


class HostLibraryMissing(RuntimeError):
    pass


def lib_paths():
    # JELLO_HIP_LIB: an experiment library built with `make -C jello_amd/csrc VARIANT=...` (tools/ only; the tests and
    # bench.py's default run never set it)
    return {
        "hip": os.environ.get("JELLO_HIP_LIB") or os.path.join(_HERE, "libjello_hip.so"),
        # JELLO_HOST_LIB: the sanitizer build of tools/sanitize_cpu.sh (CPU suite under ASan / UBSan)
        "host": os.environ.get("JELLO_HOST_LIB") or os.path.join(_HERE, "libjello_host.so"),
    }


def build(verbose=False):
    """Compile every HIP extension for gfx950 (hipcc cross-compiles without a GPU) and the host lib."""
    for sub in ("csrc", "host"):
        cmd = ["make", "-C", os.path.join(_HERE, sub), "-j8"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if verbose or r.returncode != 0:
            print(r.stdout)
        if r.returncode != 0:
            raise RuntimeError("build failed in jello_amd/%s" % sub)
    return lib_paths()


_host = None


def load_host():
    """Load libjello_host.so (which pulls in libjello_hip.so through its rpath)."""
    global _host
    if _host is not None:
        return _host
    p = lib_paths()
    for k in ("hip", "host"):
        if not os.path.exists(p[k]):
            raise HostLibraryMissing(
                "%s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the render path)" % p[k])
    ctypes.CDLL(p["hip"], mode=ctypes.RTLD_GLOBAL)
    _host = ctypes.CDLL(p["host"])
    _declare(_host)
    return _host


# ---- ctypes mirrors of the structs in host/capi.cpp ----
class PathEl(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("pad", ctypes.c_int32), ("pts", ctypes.c_double * 6)]


class CColorStop(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_float), ("pad", ctypes.c_float), ("rgba", ctypes.c_double * 4)]


class CBrush(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32), ("extend", ctypes.c_int32), ("color", ctypes.c_double * 4),
        ("p0", ctypes.c_double * 2), ("p1", ctypes.c_double * 2),
        ("r0", ctypes.c_float), ("r1", ctypes.c_float), ("t0", ctypes.c_float), ("t1", ctypes.c_float),
        ("stops", ctypes.POINTER(CColorStop)), ("n_stops", ctypes.c_int32),
        ("image_width", ctypes.c_uint32), ("image_height", ctypes.c_uint32),
        ("image_pixels", ctypes.c_void_p), ("image_key", ctypes.c_uint64),
    ]


class CStroke(ctypes.Structure):
    _fields_ = [("width", ctypes.c_double), ("join", ctypes.c_int32), ("start_cap", ctypes.c_int32),
                ("end_cap", ctypes.c_int32), ("pad", ctypes.c_int32), ("miter_limit", ctypes.c_double)]


class CBumpSizes(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("bin_data", "tiles", "lines", "seg_counts", "segments", "blend_spill", "ptcl")]


class CRenderParams(ctypes.Structure):
    _fields_ = [("base_color", ctypes.c_double * 4), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("aa", ctypes.c_int32), ("pad", ctypes.c_uint32), ("bump", CBumpSizes)]


class CBinding(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("count", ctypes.c_uint32), ("id", ctypes.c_uint64), ("size", ctypes.c_uint64),
                ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("format", ctypes.c_int32), ("pad", ctypes.c_int32),
                ("ids", ctypes.POINTER(ctypes.c_uint64)), ("dims", ctypes.POINTER(ctypes.c_uint32))]


class CCommand(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("shader", ctypes.c_int32), ("wg", ctypes.c_uint32 * 3), ("pad", ctypes.c_uint32),
                ("buf_id", ctypes.c_uint64), ("buf_size", ctypes.c_uint64), ("buf_name", ctypes.c_char_p),
                ("img_id", ctypes.c_uint64), ("img_w", ctypes.c_uint32), ("img_h", ctypes.c_uint32),
                ("img_format", ctypes.c_int32), ("n_bindings", ctypes.c_int32),
                ("data", ctypes.POINTER(ctypes.c_uint8)), ("data_len", ctypes.c_uint64),
                ("offset", ctypes.c_uint64), ("size", ctypes.c_int64), ("bindings", ctypes.POINTER(CBinding)),
                ("coords", ctypes.c_uint32 * 4)]


class CConfig(ctypes.Structure):
    """include/jello_formats.h JlConfig (renderer/config.go ConfigUniform)."""
    _fields_ = [("width_in_tiles", ctypes.c_uint32), ("height_in_tiles", ctypes.c_uint32), ("target_width", ctypes.c_uint32),
                ("target_height", ctypes.c_uint32), ("base_color", ctypes.c_float * 4),
                ("n_drawobj", ctypes.c_uint32), ("n_path", ctypes.c_uint32), ("n_clip", ctypes.c_uint32),
                ("bin_data_start", ctypes.c_uint32), ("pathtag_base", ctypes.c_uint32), ("pathdata_base", ctypes.c_uint32),
                ("drawtag_base", ctypes.c_uint32), ("drawdata_base", ctypes.c_uint32), ("transform_base", ctypes.c_uint32),
                ("style_base", ctypes.c_uint32), ("lines_size", ctypes.c_uint32), ("binning_size", ctypes.c_uint32),
                ("tiles_size", ctypes.c_uint32), ("seg_counts_size", ctypes.c_uint32), ("segments_size", ctypes.c_uint32),
                ("blend_size", ctypes.c_uint32), ("ptcl_size", ctypes.c_uint32)]


def _declare(L):
    vp, ci, cu = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint
    dp = ctypes.POINTER(ctypes.c_double)
    L.jl_last_error.restype = ctypes.c_char_p
    L.jl_scene_new.restype = vp
    L.jl_scene_free.argtypes = [vp]
    L.jl_scene_reset.argtypes = [vp]
    L.jl_scene_fill.argtypes = [vp, ci, dp, ctypes.POINTER(CBrush), dp, ctypes.POINTER(PathEl), ci]
    L.jl_scene_stroke.argtypes = [vp, ctypes.POINTER(CStroke), dp, ctypes.POINTER(CBrush), dp, ctypes.POINTER(PathEl), ci]
    L.jl_scene_push_layer.argtypes = [vp, ci, ci, ctypes.c_float, dp, ctypes.POINTER(PathEl), ci]
    L.jl_scene_pop_layer.argtypes = [vp]
    L.jl_scene_append.argtypes = [vp, vp, dp]
    L.jl_scene_apply_transform.argtypes = [vp, dp]
    L.jl_scene_stream.restype = ctypes.c_uint64
    L.jl_scene_stream.argtypes = [vp, ci, ctypes.POINTER(vp)]
    L.jl_scene_counts.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    L.jl_scene_bump_sizes.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp]
    L.jl_scene_bump_sizes_clamped.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32]
    L.jl_scene_bump_sizes_clamped.restype = ctypes.c_uint32
    L.jl_scene_bump_estimate.argtypes = [vp, dp, ctypes.POINTER(ctypes.c_uint32)]
    L.jl_scene_fill_stroke_cubics.argtypes = [vp, ci, dp, dp, dp, dp, ci, ci, ci]
    L.jl_ptcl_stats.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
    L.jl_host_new.restype = vp
    L.jl_host_free.argtypes = [vp]
    L.jl_record.restype = vp
    L.jl_record.argtypes = [vp, vp, ctypes.POINTER(CRenderParams), ci]
    L.jl_recording_free.argtypes = [vp]
    L.jl_recording_len.argtypes = [vp]
    L.jl_recording_commands.restype = ctypes.POINTER(CCommand)
    L.jl_recording_commands.argtypes = [vp]
    L.jl_recording_config.restype = ctypes.POINTER(CConfig)
    L.jl_recording_config.argtypes = [vp]
    L.jl_recording_target.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.jl_recording_buffer.restype = ctypes.c_uint64
    L.jl_recording_buffer.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
    L.jl_recording_wg_counts.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ci]
    L.jl_engine_new.restype = vp
    L.jl_engine_new.argtypes = [ci]
    L.jl_engine_free.argtypes = [vp]
    L.jl_engine_ctx.restype = vp
    L.jl_engine_ctx.argtypes = [vp]
    L.jl_engine_run.argtypes = [vp, vp, cu, ctypes.c_uint64, vp]
    L.jl_engine_release.argtypes = [vp, vp]
    L.jl_engine_render.restype = vp
    L.jl_engine_render.argtypes = [vp, vp, ctypes.POINTER(CRenderParams), vp, ci, ci, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ci)]
    # C ABI of libjello_hip.so (include/jello_hip.h), reachable through the same process image
    hip = ctypes.CDLL(lib_paths()["hip"])
    L.hip = hip
    hip.jh_last_error.restype = ctypes.c_char_p
    hip.jh_last_error.argtypes = [vp]
    hip.jh_stage_name.restype = ctypes.c_char_p
    hip.jh_download.argtypes = [vp, ctypes.c_uint64, vp, ctypes.c_uint64, ctypes.c_uint64]
    hip.jh_upload.argtypes = [vp, ctypes.c_uint64, vp, ctypes.c_uint64]
    hip.jh_buffer_create.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64]
    hip.jh_buffer_size.restype = ctypes.c_uint64
    hip.jh_buffer_size.argtypes = [vp, ctypes.c_uint64]
    hip.jh_buffer_device_ptr.restype = vp
    hip.jh_buffer_device_ptr.argtypes = [vp, ctypes.c_uint64]
    hip.jh_image_download.argtypes = [vp, ctypes.c_uint64, vp, ctypes.c_uint64]
    hip.jh_image_device_ptr.restype = vp
    hip.jh_image_device_ptr.argtypes = [vp, ctypes.c_uint64]
    hip.jh_sync.argtypes = [vp]
    hip.jh_set_stream.argtypes = [vp, vp]
    hip.jh_set_band.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32]
    hip.jh_profile_enable.argtypes = [vp, ci]
    hip.jh_profile_collect.argtypes = [vp, vp, ci]
    hip.jh_profile_group_begin.argtypes = [vp, ctypes.c_char_p]
    hip.jh_profile_group_end.argtypes = [vp]
    hip.jh_profile_collect_tree.argtypes = [vp, vp, ci]
    hip.jh_image_create.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ci]
    hip.jh_image_write.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp, ctypes.c_uint64]
    hip.jh_buffer_import.argtypes = [vp, ctypes.c_uint64, vp, ctypes.c_uint64]
    hip.jh_graph_begin.argtypes = [vp]
    hip.jh_graph_end.argtypes = [vp, ctypes.POINTER(vp)]
    hip.jh_graph_launch.argtypes = [vp, vp]
    hip.jh_graph_node_counts.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    hip.jh_graph_destroy.argtypes = [vp, vp]
    hip.jh_free.argtypes = [vp, ctypes.c_uint64]
    hip.jh_clear.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64]
    hip.jh_pool_bytes.restype = ctypes.c_uint64
    hip.jh_pool_bytes.argtypes = [vp]
    hip.jh_device_info.argtypes = [vp, ctypes.c_char_p, ci, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_uint64)]
    hip.jh_debug_poison_scratch.argtypes = [vp, ci]
    hip.jh_debug_scratch_bytes.restype = ctypes.c_uint64
    hip.jh_debug_scratch_bytes.argtypes = [vp, ci]
    if hasattr(hip, "jh_debug_flatten_regions"):
        hip.jh_debug_flatten_regions.restype = ci
        hip.jh_debug_flatten_regions.argtypes = [vp, ctypes.c_uint32]
    if hasattr(hip, "jh_scratch_trim"):
        hip.jh_scratch_trim.restype = ci
        hip.jh_scratch_trim.argtypes = [vp]
    hip.jh_set_clip_depth_hint.argtypes = [vp, ctypes.c_uint32]
    if hasattr(hip, "jh_debug_clip_hint_overflows"):  # (an experiment library built from older sources, JELLO_HIP_LIB, may lack it)
        hip.jh_debug_clip_hint_overflows.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
    hip.jh_debug_graph_self_cleans.restype = ctypes.c_uint64
    hip.jh_debug_graph_self_cleans.argtypes = [vp]
