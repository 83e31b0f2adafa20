"""Run-to-run determinism at full size: replay the C3 frame N times and compare the bytes of every pipeline buffer and of
the image with the first run (any race in the count -> scan -> write scheme would show up as a differing buffer)."""
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import jello_amd
from jello_amd import BumpSizes, scenes, Aa
from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = jello_amd.Engine(0)
host = jello_amd.Host()
names = ["bumpBuf", "tagmonoidBuf", "pathBboxBuf", "linesBuf", "drawMonoidBuf", "infoBinDataBuf", "binHeaderBuf", "pathBuf", "tileBuf",
         "segCountsBuf", "segmentsBuf", "ptclBuf"]
for label, mk, aa in [("C3 100k @4096 area", lambda: scenes.scene_c3(100000, 4096), Aa.Area),
                      ("C4 30k @2048 msaa8", lambda: scenes.scene_c4(30000, 2048), Aa.Msaa8)]:
    s, p = mk()
    p.aa = aa
    p.bump = BumpSizes(lines=1 << 23, seg_counts=1 << 24, segments=1 << 24, tiles=1 << 23, ptcl=1 << 27, bin_data=1 << 22, blend_spill=1 << 26)
    rec = host.record(s, p)
    t = rec.target
    ref = None
    eng.run(rec, RUN_UPLOADS | RUN_DISPATCHES)  # allocates every buffer (possibly recycled pool memory with old contents)
    eng.sync()
    for it in range(n):
        for nm in names:  # stale bytes must neither mask a missing write nor differ between runs
            eng.clear(rec.buffer(nm)[0])
        eng.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
        eng.sync()
        h = {}
        for nm in names:
            h[nm] = hashlib.sha1(eng.download(rec.buffer(nm)[0], dtype=np.uint8).tobytes()).hexdigest()
        h["image"] = hashlib.sha1(eng.download_image(t["id"], t["width"], t["height"]).tobytes()).hexdigest()
        if ref is None:
            ref = h
        else:
            diff = [k for k in h if h[k] != ref[k]]
            if diff:
                print("%s run %d differs in %s" % (label, it, diff), flush=True)
                sys.exit(1)
    print("%s: %d runs identical (%d buffers + image)" % (label, n, len(names)), flush=True)
    eng.release(rec)

# ---- graph soak: the C3 frame captured, poisoned scratch between captures, many replays each ----
# usage: determinism.py <n eager> <n graph replays>   (second argument; default 0 = skip)
n_replay = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if n_replay:
    s, p = scenes.scene_c3(100000, 4096)
    p.bump = s.bump_sizes(4096, 4096)
    rec = host.record(s, p)
    t = rec.target
    eng.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
    eng.sync()

    def digest():
        h = hashlib.sha1()
        h.update(eng.download(rec.buffer("bumpBuf")[0], dtype=np.uint8)[:32].tobytes())
        h.update(eng.download(rec.buffer("ptclBuf")[0], dtype=np.uint8).tobytes())
        h.update(eng.download_image(t["id"], t["width"], t["height"]).tobytes())
        return h.hexdigest()
    ref = digest()
    captures, per = 10, max(1, n_replay // 10)
    done = 0
    for c in range(captures):
        if c & 1:
            eng.debug_poison_scratch(0xA5 if c & 2 else 0x5A)  # the graph is captured on dirty counters: it carries its own fills
        g = eng.capture(rec)
        for i in range(per):
            if i % 97 == 13:
                eng.debug_poison_scratch(0xC3)                  # replay on dirty counters: jh_graph_launch cleans first
            eng.replay(g)
            if i % 50 == 49 or i == per - 1:                     # (hashing a 128 MiB frame every replay would measure the PCIe link)
                eng.sync()
                d = digest()
                if d != ref:
                    print("graph soak: capture %d replay %d differs" % (c, i), flush=True)
                    sys.exit(1)
            done += 1
        eng.sync()
        eng.graph_destroy(g)
    print("graph soak: %d replays over %d captures (odd ones captured on poisoned scratch, a poisoned replay every 97): identical; "
          "self-cleaning replays: %d" % (done, captures, eng.graph_self_cleans()), flush=True)
    eng.release(rec)
