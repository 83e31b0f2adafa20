"""Shared parity helpers: run one scene through the HIP engine and the CPU oracle and compare every
buffer of the pipeline bit for bit (integer/index data and f32 bit patterns alike)."""
import numpy as np

import jello_amd
from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS
from oracle.oracle_engine import OracleEngine

BUMP = ["failed", "binning", "ptcl", "tile", "seg_counts", "segments", "blend", "lines"]


def ptcl_walk(ptcl, cfg):
    """Follow every tile's command stream; returns the set of word indices that are live (reachable)."""
    live = np.zeros(ptcl.shape[0], dtype=bool)
    sizes = {1: 4, 3: 1, 5: 5, 6: 3, 7: 3, 8: 3, 9: 2, 10: 1, 11: 3}
    ntiles = cfg["width_in_tiles"] * cfg["height_in_tiles"]
    for t in range(ntiles):
        ix = t * 64
        live[ix] = True
        ix += 1
        for _ in range(1 << 22):
            tag = int(ptcl[ix])
            if tag == 0:
                live[ix] = True
                break
            if tag == 12:
                live[ix:ix + 2] = True
                ix = int(ptcl[ix + 1])
                continue
            n = sizes[tag]
            live[ix:ix + n] = True
            ix += n
    return live


def compare(engine, scene, params, check_image=True, names=None):
    """Returns a dict of results; raises AssertionError with a precise message on the first mismatch."""
    host = jello_amd.Host()
    rec = host.record(scene, params)
    cfg = rec.config
    # --- GPU (retain everything: run without the deferred frees) ---
    engine.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    engine.sync()
    # --- oracle ---
    orc = OracleEngine()
    orc.run(rec)
    out = {}
    try:
        gb = engine.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8]
        ob = orc.get(rec, "bumpBuf", np.uint32)[:8]
        out["bump"] = dict(zip(BUMP, [int(x) for x in gb]))
        assert list(gb) == list(ob), "bump allocators differ: gpu %s oracle %s" % (dict(zip(BUMP, gb)), dict(zip(BUMP, ob)))
        assert gb[0] == 0, "bump.failed = %d (buffers too small for this scene)" % gb[0]
        n_lines, n_segc, n_seg, n_tile = int(gb[7]), int(gb[4]), int(gb[5]), int(gb[3])
        nd = cfg["n_drawobj"]
        n_tagw = (cfg["pathdata_base"] - cfg["pathtag_base"])

        def both(name, dtype, count=None, cols=1):
            g = engine.download(rec.buffer(name)[0], dtype=dtype)
            o = orc.get(rec, name, dtype)
            if count is not None:
                g, o = g[:count * cols], o[:count * cols]
            return g, o

        def check(name, dtype, count, cols=1, mask=None):
            g, o = both(name, dtype, count, cols)
            if mask is not None:
                g, o = g[mask], o[mask]
            if not np.array_equal(g, o):
                bad = np.flatnonzero(g != o)
                i = int(bad[0])
                raise AssertionError("%s differs at %d word(s); first at word %d (row %d): gpu 0x%x oracle 0x%x" %
                                     (name, bad.size, i, i // cols, int(g[i]), int(o[i])))
            out[name] = int(g.size)

        check("tagmonoidBuf", np.uint32, n_tagw, 5)
        check("pathBboxBuf", np.uint32, cfg["n_path"], 6)
        check("linesBuf", np.uint32, n_lines, 6)
        check("drawMonoidBuf", np.uint32, nd, 4)
        check("infoBinDataBuf", np.uint32, cfg["bin_data_start"] + int(gb[1]))
        if cfg["n_clip"]:
            check("clipBboxBuf", np.uint32, cfg["n_clip"], 4)
        check("drawBboxBuf", np.uint32, nd, 4)
        check("binHeaderBuf", np.uint32, ((nd + 255) // 256) * 256, 2)
        check("pathBuf", np.uint32, nd * 8 // 8 * 8 // 8, 8, mask=np.tile(np.array([1, 1, 1, 1, 1, 0, 0, 0], bool), nd))
        check("segCountsBuf", np.uint32, n_segc, 2)
        if n_segc:  # SegmentCount.counts = seg_within_slice << 16 | seg_within_line
            out["max_tile_segments"] = int(both("segCountsBuf", np.uint32, n_segc, 2)[0][1::2].max() >> 16) + 1
        check("tileBuf", np.uint32, n_tile, 2)
        # segments: 5 live words of 6 per record
        check("segmentsBuf", np.uint32, n_seg, 6, mask=np.tile(np.array([1, 1, 1, 1, 1, 0], bool), n_seg))
        # PTCL: compare every reachable word (the rest of a 256-word chunk / 64-word head is stale pool memory)
        gp, op = both("ptclBuf", np.uint32)
        live_o = ptcl_walk(op, cfg)
        live_g = ptcl_walk(gp, cfg)
        assert np.array_equal(live_o, live_g), "PTCL reachability differs"
        if not np.array_equal(gp[live_o], op[live_o]):
            bad = np.flatnonzero(live_o & (gp != op))
            raise AssertionError("ptclBuf differs at %d live word(s); first at %d: gpu 0x%x oracle 0x%x" % (bad.size, bad[0], gp[bad[0]], op[bad[0]]))
        out["ptcl_live_words"] = int(live_o.sum())
        if check_image:
            t = rec.target
            gi = engine.download_image(t["id"], t["width"], t["height"])
            oi = orc.target(rec).copy()

            def canon_nan(a):  # 0/0 is -qNaN on x86 and +qNaN on gfx950; any NaN compares equal to any NaN
                a = a.copy()
                a[(a & 0x7fff) > 0x7c00] = 0x7e00
                return a
            gi, oi = canon_nan(gi), canon_nan(oi)
            if not np.array_equal(gi, oi):
                bad = np.argwhere(gi != oi)
                y, x, c = bad[0]
                # f16 ULP distance
                def ordered(a):
                    a = a.astype(np.int32)
                    return np.where(a & 0x8000, 0x8000 - (a & 0x7fff), a)
                ulp = np.abs(ordered(gi) - ordered(oi)).max()
                raise AssertionError("image differs in %d channel values, max %d ULP(f16); first at (x=%d,y=%d,c=%d): gpu 0x%04x oracle 0x%04x" %
                                     (len(bad), ulp, x, y, c, gi[y, x, c], oi[y, x, c]))
            out["image"] = gi
        out["stage_seconds_oracle"] = dict(orc.stage_seconds)
    finally:
        engine.release(rec)
    return out
