"""Recording and Engine wrappers.

``Host.record`` = renderer.RenderFull (render.go:572-588): produces the Recording (no GPU needed).
``Engine``      = engine/hip_engine: replays a Recording on the MI355X through the C ABI.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import CConfig

STAGE_NAMES = ["pathtag_reduce", "pathtag_reduce2", "pathtag_scan1", "pathtag_scan_small", "pathtag_scan_large", "bbox_clear",
               "flatten", "draw_reduce", "draw_leaf", "clip_reduce", "clip_leaf", "binning", "tile_alloc", "backdrop_dyn",
               "path_count_setup", "path_count", "coarse", "path_tiling_setup", "path_tiling", "fine_area", "fine_msaa8", "fine_msaa16"]


class CMD:
    UPLOAD, UPLOAD_UNIFORM, UPLOAD_IMAGE, WRITE_IMAGE, DISPATCH, DISPATCH_INDIRECT, DOWNLOAD, CLEAR, FREE_BUFFER, FREE_IMAGE = range(10)


RUN_UPLOADS, RUN_DISPATCHES, RUN_FREES, RUN_ALL = 1, 2, 4, 7
RUN_SKIP_FINE, RUN_ONLY_FINE = 8, 16  # with RUN_DISPATCHES: everything but the fine stage / the fine stage alone


class Recording:
    """A renderer.Recording plus the RenderConfig it was built from."""

    def __init__(self, L, handle):
        self._L, self._h = L, handle

    def __del__(self):
        try:
            self._L.jl_recording_free(self._h)
        except Exception:
            pass

    def __len__(self):
        return self._L.jl_recording_len(self._h)

    def commands(self):
        n = len(self)
        arr = self._L.jl_recording_commands(self._h)
        out = []
        for i in range(n):
            c = arr[i]
            binds = []
            for j in range(c.n_bindings):
                b = c.bindings[j]
                d = {"kind": b.kind, "id": b.id, "size": b.size, "width": b.width, "height": b.height, "format": b.format}
                if b.kind == 3:
                    d["ids"] = [b.ids[k] for k in range(b.count)]
                    d["dims"] = [(b.dims[3 * k], b.dims[3 * k + 1], b.dims[3 * k + 2]) for k in range(b.count)]
                binds.append(d)
            data = ctypes.string_at(c.data, c.data_len) if c.data_len else b""
            out.append({"kind": c.kind, "shader": c.shader, "wg": tuple(c.wg), "buf_id": c.buf_id, "buf_size": c.buf_size,
                        "buf_name": (c.buf_name or b"").decode(), "img_id": c.img_id, "img_w": c.img_w, "img_h": c.img_h,
                        "img_format": c.img_format, "data": data, "offset": c.offset, "size": c.size, "bindings": binds,
                        "coords": tuple(c.coords)})
        return out

    @property
    def config(self):
        c = self._L.jl_recording_config(self._h).contents
        return {f: (list(getattr(c, f)) if f == "base_color" else getattr(c, f)) for f, _ in CConfig._fields_}

    def config_bytes(self):
        return ctypes.string_at(self._L.jl_recording_config(self._h), ctypes.sizeof(CConfig))

    @property
    def target(self):
        i, w, h = ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_uint32()
        self._L.jl_recording_target(self._h, ctypes.byref(i), ctypes.byref(w), ctypes.byref(h))
        return {"id": i.value, "width": w.value, "height": h.value}

    def buffer(self, name):
        sz = ctypes.c_uint64()
        i = self._L.jl_recording_buffer(self._h, name.encode(), ctypes.byref(sz))
        if i == 0:
            raise KeyError(name)
        return i, sz.value

    def workgroup_counts(self):
        names = ["path_reduce", "path_reduce2", "path_scan1", "path_scan", "bbox_clear", "flatten", "draw_reduce", "draw_leaf",
                 "clip_reduce", "clip_leaf", "binning", "tile_alloc", "path_count_setup", "backdrop", "coarse", "path_tiling_setup", "fine"]
        out = (ctypes.c_uint32 * (3 * len(names) + 1))()
        self._L.jl_recording_wg_counts(self._h, out, len(out))
        d = {n: tuple(out[3 * i:3 * i + 3]) for i, n in enumerate(names)}
        d["use_large_path_scan"] = bool(out[3 * len(names)])
        return d


class Host:
    """renderer.Renderer + renderer.Resolver + FullShaders: the recording side (CPU only)."""

    def __init__(self):
        self._L = _lib.load_host()
        self._h = self._L.jl_host_new()

    def __del__(self):
        try:
            self._L.jl_host_free(self._h)
        except Exception:
            pass

    def record(self, scene, params, robust=False):
        p = params._c()
        h = self._L.jl_record(self._h, scene._h, ctypes.byref(p), 1 if robust else 0)
        if not h:
            raise RuntimeError(self._L.jl_last_error().decode())
        return Recording(self._L, h)


class Engine:
    """engine/hip_engine: one context = one GPU + one stream.  Raises if no MI355X/HIP is available."""

    def __init__(self, device=0):
        self._L = _lib.load_host()
        self._h = self._L.jl_engine_new(device)
        if not self._h:
            raise RuntimeError("hip_engine: " + self._L.jl_last_error().decode())
        self.ctx = self._L.jl_engine_ctx(self._h)
        self.hip = self._L.hip

    def close(self):
        if self._h:
            self._L.jl_engine_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s | %s" % (what, rc, self._L.jl_last_error().decode(), self.hip.jh_last_error(self.ctx).decode()))

    def run(self, recording, flags=RUN_ALL, out_device_ptr=None):
        self._check(self._L.jl_engine_run(self._h, recording._h, flags, 0, out_device_ptr), "run_recording")

    def release(self, recording):
        self._check(self._L.jl_engine_release(self._h, recording._h), "release")

    def render(self, scene, params, out_device_ptr=None, robust=True, retain=False):
        """RenderToTexture (+ regrow loop).  Returns (Recording, bump dict, attempts)."""
        p = params._c()
        bump = (ctypes.c_uint32 * 8)()
        attempts = ctypes.c_int()
        h = self._L.jl_engine_render(self._h, scene._h, ctypes.byref(p), out_device_ptr, 1 if robust else 0, 1 if retain else 0, bump, ctypes.byref(attempts))
        if not h:
            raise RuntimeError("render_to_texture: " + self._L.jl_last_error().decode())
        names = ["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"]
        return Recording(self._L, h), dict(zip(names, bump)), attempts.value

    def capture(self, recording, out_device_ptr=None):
        """Capture one dispatch-only replay of `recording` into a hipGraph; returns an opaque handle for replay().
        The recording must have been run once (buffers + scratch exist)."""
        self._check(self.hip.jh_graph_begin(self.ctx), "graph_begin")
        try:
            self.run(recording, RUN_DISPATCHES, out_device_ptr)
        finally:
            g = ctypes.c_void_p()
            rc = self.hip.jh_graph_end(self.ctx, ctypes.byref(g))
        self._check(rc, "graph_end")
        return g

    def replay(self, graph):
        self._check(self.hip.jh_graph_launch(self.ctx, graph), "graph_launch")

    def graph_node_counts(self, graph):
        """(kernel launches, other nodes) of one replay of a captured frame."""
        k, o = ctypes.c_uint32(0), ctypes.c_uint32(0)
        self._check(self.hip.jh_graph_node_counts(self.ctx, graph, ctypes.byref(k), ctypes.byref(o)), "graph_node_counts")
        return k.value, o.value

    def graph_destroy(self, graph):
        self.hip.jh_graph_destroy(self.ctx, graph)

    def sync(self):
        self._check(self.hip.jh_sync(self.ctx), "sync")

    def scratch_bytes(self, slot=-1):
        """Capacity of the context's internal scratch arrays (all of them, or one slot)."""
        return int(self.hip.jh_debug_scratch_bytes(self.ctx, slot))

    def trim_scratch(self):
        """Frees the internal scratch arrays (they only grow): after a frame much larger than the ones to come."""
        if hasattr(self.hip, "jh_scratch_trim"):  # (an older library under JELLO_HIP_LIB, tools/ab_kernels.sh: nothing to give back)
            self._check(self.hip.jh_scratch_trim(self.ctx), "scratch_trim")

    def set_stream(self, stream_ptr):
        self._check(self.hip.jh_set_stream(self.ctx, stream_ptr), "set_stream")

    def clear(self, buf_id, offset=0, size=-1):
        self._check(self.hip.jh_clear(self.ctx, buf_id, offset, size), "clear")

    def set_band(self, bin_row0=0, bin_row1=0xffffffff):
        """Band mode (sharding.band_for_rank): write the PTCL and rasterise only the bin rows [bin_row0, bin_row1);
        no arguments = the whole target."""
        self._check(self.hip.jh_set_band(self.ctx, int(bin_row0), int(bin_row1)), "set_band")

    def download(self, buf_id, nbytes=None, offset=0, dtype=np.uint8):
        size = self.hip.jh_buffer_size(self.ctx, buf_id)
        if nbytes is None:
            nbytes = size - offset
        out = np.empty(nbytes, dtype=np.uint8)
        self._check(self.hip.jh_download(self.ctx, buf_id, out.ctypes.data, offset, nbytes), "download")
        return out.view(dtype)

    def download_image(self, img_id, width, height):
        out = np.empty((height, width, 4), dtype=np.uint16)
        self._check(self.hip.jh_image_download(self.ctx, img_id, out.ctypes.data, out.nbytes), "image_download")
        return out

    def profile(self, on=True):
        self.hip.jh_profile_enable(self.ctx, 1 if on else 0)

    def profile_collect(self, max_records=4096):
        class Rec(ctypes.Structure):
            _fields_ = [("stage", ctypes.c_int32), ("pad", ctypes.c_uint32), ("ms", ctypes.c_float)]
        arr = (Rec * max_records)()
        n = self.hip.jh_profile_collect(self.ctx, arr, max_records)
        if n < 0:
            self._check(n, "profile_collect")
        return [(STAGE_NAMES[arr[i].stage], arr[i].ms) for i in range(n)]

    def profile_collect_tree(self, max_nodes=1 << 16):
        """Profiler.Collect (profiler.go:337-385): list of dicts {kind, parent, stage, label, cpu_start_ms, cpu_end_ms,
        gpu_start_ms, gpu_end_ms}; a node's parent precedes it."""
        class Node(ctypes.Structure):
            _fields_ = [("kind", ctypes.c_int32), ("parent", ctypes.c_int32), ("stage", ctypes.c_int32), ("pad", ctypes.c_uint32),
                        ("label", ctypes.c_char * 48), ("cpu_start_ms", ctypes.c_double), ("cpu_end_ms", ctypes.c_double),
                        ("gpu_start_ms", ctypes.c_float), ("gpu_end_ms", ctypes.c_float)]
        arr = (Node * max_nodes)()
        n = self.hip.jh_profile_collect_tree(self.ctx, arr, max_nodes)
        if n < 0:
            self._check(n, "profile_collect_tree")
        return [{"kind": "group" if arr[i].kind == 0 else "query", "parent": arr[i].parent, "stage": arr[i].stage,
                 "label": arr[i].label.decode(), "cpu_start_ms": arr[i].cpu_start_ms, "cpu_end_ms": arr[i].cpu_end_ms,
                 "gpu_start_ms": arr[i].gpu_start_ms, "gpu_end_ms": arr[i].gpu_end_ms} for i in range(n)]

    def profile_group(self, label):
        """Context manager: ProfilerGroup.Nest(label) ... End()."""
        eng = self

        class _G:
            def __enter__(self_inner):
                eng._check(eng.hip.jh_profile_group_begin(eng.ctx, label.encode()), "profile_group_begin")

            def __exit__(self_inner, *a):
                eng._check(eng.hip.jh_profile_group_end(eng.ctx), "profile_group_end")
        return _G()

    def debug_poison_scratch(self, byte=0xA5):
        """jh_debug_poison_scratch: scratch memory as a fresh, non-zero allocation would be (tests only)."""
        rc = self.hip.jh_debug_poison_scratch(self.ctx, int(byte))
        if rc != 0:
            raise RuntimeError("jh_debug_poison_scratch failed: %d" % rc)

    def graph_self_cleans(self):
        """Replays that found one of the internal counters dirty (failed frame, poisoned scratch) and zeroed it first."""
        return int(self.hip.jh_debug_graph_self_cleans(self.ctx))

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        cus, mem = ctypes.c_int(), ctypes.c_uint64()
        self.hip.jh_device_info(self.ctx, name, 256, ctypes.byref(cus), ctypes.byref(mem))
        return {"name": name.value.decode(), "compute_units": cus.value, "total_mem": mem.value}
