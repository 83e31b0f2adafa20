"""CPU: the C++ host mirror of encoding/ + renderer/ (jello_amd/host) behaves like the reference's Go
code: tag/transform/style de-duplication, fill auto-close, stroke cap markers, clip bookkeeping,
Layout, WorkgroupCounts, BufferSizes and the dispatch DAG of RenderFull."""
import numpy as np
import pytest

import jello_amd
from jello_amd import (Brush, Cap, Compose, Fill, Host, Join, Mix, Path, RenderParams, Scene, Stroke, STAGE_NAMES, CMD)

RED = Brush.solid((1, 0, 0, 1))


def tags(s):
    return list(s.stream("path_tags"))


def test_transform_and_style_dedup(built):
    s = Scene()
    tri = Path().move_to(0, 0).line_to(10, 0).line_to(0, 10).close()
    s.fill(Fill.NonZero, None, RED, None, tri)
    s.fill(Fill.NonZero, None, RED, None, tri)       # same transform + style: no new markers (encoding.go:124-149)
    s.fill(Fill.EvenOdd, None, RED, None, tri)       # new style only
    s.fill(Fill.EvenOdd, (2, 0, 0, 2, 5, 5), RED, None, tri)  # new transform only
    t = tags(s)
    assert t.count(0x20) == 2 and t.count(0x40) == 2 and t.count(0x10) == 4
    assert s.counts()["num_paths"] == 4
    styles = np.frombuffer(s.stream("styles"), np.uint32).reshape(-1, 2)
    assert styles[0, 0] == 0 and styles[1, 0] == 0x40000000


def test_fill_is_auto_closed_and_zero_length_segments_dropped(built):
    s = Scene()
    s.fill(Fill.NonZero, None, RED, None, Path().move_to(0, 0).line_to(10, 0).line_to(10, 0).line_to(10, 10))  # no explicit close
    t = tags(s)
    # transform, style, line, line, auto-close line | SUBPATH_END, path (path.go:377-405, 234-254)
    assert t == [0x20, 0x40, 0x09, 0x09, 0x0d, 0x10]
    pd = np.frombuffer(s.stream("path_data"), np.float32)
    assert pd.tolist() == [0, 0, 10, 0, 10, 10, 0, 0]


def test_open_and_closed_stroke_cap_markers(built):
    s = Scene()
    st = Stroke(2.0, Join.Bevel, 4.0, Cap.Square, Cap.Round)
    s.stroke(st, None, RED, None, Path().move_to(0, 0).line_to(10, 0).line_to(10, 10))
    t = tags(s)
    assert t == [0x20, 0x40, 0x09, 0x09, 0x0e, 0x10]       # open: quad marker | SUBPATH_END (path.go:459-482)
    pd = np.frombuffer(s.stream("path_data"), np.float32)
    assert pd[-4:].tolist() == [0, 0, 10, 0]                # marker = (first point, start tangent end)
    flags = np.frombuffer(s.stream("styles"), np.uint32)[0]
    assert flags == (0x80000000 | 0x00000000 | (0x01000000 << 2) | 0x02000000 | 0x4400)
    s2 = Scene()
    s2.stroke(st, None, RED, None, Path().move_to(0, 0).line_to(10, 0).line_to(10, 10).close())
    assert tags(s2) == [0x20, 0x40, 0x09, 0x09, 0x09, 0x0d, 0x10]  # closed: close line + line marker | SUBPATH_END


def test_empty_path_is_not_encoded_and_layers_balance(built):
    s = Scene()
    s.fill(Fill.NonZero, None, RED, None, Path())            # nothing
    assert s.counts()["num_paths"] == 0 and s.stream("draw_tags") == b""
    s.push_layer(Mix.Multiply, Compose.SrcOver, 0.5, None, Path.rect(0, 0, 50, 50))
    s.fill(Fill.NonZero, None, RED, None, Path.rect(10, 10, 20, 20))
    s.pop_layer()
    s.pop_layer()                                            # unbalanced pop is ignored (encoding.go:368-371)
    c = s.counts()
    assert c == {"num_paths": 3, "num_path_segments": 8, "num_clips": 2, "num_open_clips": 0}
    dt = np.frombuffer(s.stream("draw_tags"), np.uint32).tolist()
    assert dt == [0x9, 0x50, 0x21]
    dd = np.frombuffer(s.stream("draw_data"), np.uint32)
    assert dd[0] == (1 << 8) | 0 and dd[1:2].view(np.float32)[0] == 0.5


def test_invalid_clip_shape_becomes_empty_path(built):
    s = Scene()
    s.push_layer(Mix.Clip, Compose.SrcOver, 1.0, None, Path())   # scene.go:54-66
    s.pop_layer()
    assert s.counts()["num_paths"] == 2
    assert tags(s)[-3:] == [0x0d, 0x10, 0x10]                     # empty-shape line | SUBPATH_END, its path, EndClip's dummy path


def test_open_clips_are_closed_by_the_resolver(built):
    s = Scene()
    s.push_layer(Mix.Clip, Compose.SrcOver, 1.0, None, Path.rect(0, 0, 50, 50))
    s.fill(Fill.NonZero, None, RED, None, Path.rect(10, 10, 20, 20))
    rec = Host().record(s, RenderParams(64, 64))
    cfg = rec.config
    assert cfg["n_path"] == 2 and cfg["n_drawobj"] == 2 and cfg["n_clip"] == 1  # resolve.go:86-88,117-119
    scene = np.frombuffer(rec.commands()[1]["data"], np.uint32)
    assert scene[cfg["drawtag_base"]:cfg["drawtag_base"] + 3].tolist() == [0x9, 0x50, 0x21]


def test_gradient_patches_and_ramps(built):
    from jello_amd import ColorStop
    s = Scene()
    stops = [ColorStop(0.0, (1, 0, 0, 1)), ColorStop(1.0, (0, 0, 1, 1))]
    s.fill(Fill.NonZero, None, Brush.linear((0, 0), (100, 0), stops), None, Path.rect(0, 0, 100, 100))
    s.fill(Fill.NonZero, None, Brush.radial((50, 50), 5.0, (50, 50), 40.0, stops), None, Path.rect(0, 0, 100, 100))
    s.fill(Fill.NonZero, None, Brush.linear((0, 0), (100, 0), stops), None, Path.rect(0, 0, 100, 100))   # same stops: same ramp id
    rec = Host().record(s, RenderParams(128, 128))
    cfg = rec.config
    scene = np.frombuffer(rec.commands()[2]["data"], np.uint32) if rec.commands()[1]["kind"] != CMD.UPLOAD else None
    up = [c for c in rec.commands() if c["kind"] == CMD.UPLOAD and c["buf_name"] == "scene"][0]
    scene = np.frombuffer(up["data"], np.uint32)
    assert scene[cfg["drawtag_base"]:cfg["drawtag_base"] + 3].tolist() == [0x114, 0x29c, 0x114]
    assert cfg["bin_data_start"] == 4 + 10 + 4
    dd = cfg["drawdata_base"]
    assert scene[dd] == (0 << 2) | 0 and scene[dd + 5] == (0 << 2) | 0 and scene[dd + 12] == (0 << 2) | 0   # ramp id 0, Pad
    img = [c for c in rec.commands() if c["kind"] == CMD.UPLOAD_IMAGE and c["img_format"] == 3][0]
    assert (img["img_w"], img["img_h"]) == (512, 1)
    ramp = np.frombuffer(img["data"], np.float16).reshape(512, 4).astype(np.float32)
    assert tuple(ramp[0]) == (1, 0, 0, 1) and tuple(ramp[-1]) == (0, 0, 1, 1) and np.all(ramp[:, 3] == 1)


def test_workgroup_counts_large_scan_and_buffer_sizes(built):
    from jello_amd import scenes
    s, p = scenes.scene_c3(70_000, 1024)      # 140k draw objects -> 560k tag bytes -> 547 tag workgroups > 256
    rec = Host().record(s, p)
    wg = rec.workgroup_counts()
    cfg = rec.config
    n_tag_bytes = (cfg["pathdata_base"] - cfg["pathtag_base"]) * 4
    assert n_tag_bytes % 1024 == 0
    assert wg["use_large_path_scan"] and wg["path_reduce"][0] == n_tag_bytes // 1024 and wg["path_reduce2"][0] == 256
    assert wg["path_scan1"][0] == (wg["path_reduce"][0] + 255) // 256
    assert wg["flatten"][0] == n_tag_bytes // 256
    assert wg["draw_reduce"][0] == 256 and wg["binning"][0] == (140_000 + 255) // 256
    names = [STAGE_NAMES[c["shader"]] for c in rec.commands() if c["kind"] in (CMD.DISPATCH, CMD.DISPATCH_INDIRECT)]
    assert names == ["pathtag_reduce", "pathtag_reduce2", "pathtag_scan1", "pathtag_scan_large", "bbox_clear", "flatten", "draw_reduce",
                     "draw_leaf", "binning", "tile_alloc", "path_count_setup", "path_count", "backdrop_dyn", "coarse", "path_tiling_setup",
                     "path_tiling", "fine_area"]
    # reference defaults for the bump buffers (config.go:144-151)
    assert cfg["lines_size"] == 1 << 21 and cfg["ptcl_size"] == 1 << 23 and cfg["binning_size"] == (1 << 18) - cfg["bin_data_start"]


def test_binding_contract_matches_render_go(built):
    """Binding counts per stage = SURVEY Appendix C."""
    from jello_amd import scenes
    s, p = scenes.scene_c4(60, 256)
    rec = Host().record(s, p)
    want = {"pathtag_reduce": 3, "pathtag_scan_small": 4, "bbox_clear": 2, "flatten": 6, "draw_reduce": 3, "draw_leaf": 7, "clip_reduce": 4,
            "clip_leaf": 7, "binning": 8, "tile_alloc": 6, "path_count_setup": 2, "path_count": 6, "backdrop_dyn": 4, "coarse": 9,
            "path_tiling_setup": 3, "path_tiling": 6, "fine_area": 8}
    seen = {}
    for c in rec.commands():
        if c["kind"] in (CMD.DISPATCH, CMD.DISPATCH_INDIRECT):
            seen[STAGE_NAMES[c["shader"]]] = len(c["bindings"])
    for k, v in want.items():
        if k in seen:
            assert seen[k] == v, k
    assert "clip_leaf" in seen and "fine_area" in seen
    clears = [c for c in rec.commands() if c["kind"] == CMD.CLEAR]
    assert len(clears) == 1 and clears[0]["buf_name"] == "bumpBuf" and clears[0]["size"] == -1   # render.go:237


def test_more_than_256_bins_is_refused(built):
    """binning / coarse index bins in a 256-entry table (binning.wgsl:52,131): a larger target renders a wrong frame in the
    reference; the recording is refused here.  Any shape of at most 256 bins is fine."""
    from jello_amd import scenes
    s, _ = scenes.scene_c1()
    host = jello_amd.Host()
    for w, h in ((4096, 4096), (8192, 2048), (2048, 8192), (16, 65536)):
        host.record(s, jello_amd.RenderParams(w, h))
    for w, h in ((4097, 4096), (4096, 4097), (8192, 8192)):
        with pytest.raises(RuntimeError, match="bins"):
            host.record(s, jello_amd.RenderParams(w, h))
