"""Command statistics of a rendered frame's PTCL (run on the GPU box): commands per tile by kind, live words,
segments -- the inputs of the lower bound DESIGN.md section 4 derives for the fine stage.
   python tools/ptcl_stats.py [c3|c4|c4n] [paths] [size]"""
import collections, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import jello_amd
from jello_amd import scenes
from jello_amd.engine import RUN_DISPATCHES, RUN_UPLOADS

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (100000 if which == "c3" else 30000)
size = int(sys.argv[3]) if len(sys.argv) > 3 else (4096 if which == "c3" else 2048)
s, p = {"c3": scenes.scene_c3, "c4": scenes.scene_c4, "c4n": scenes.scene_c4_nested}[which](n, size)
p.bump = s.bump_sizes(size, size)
eng = jello_amd.Engine()
rec = jello_amd.Host().record(s, p)
eng.run(rec, RUN_UPLOADS | RUN_DISPATCHES)
eng.sync()
cfg = rec.config
ptcl = eng.download(rec.buffer("ptclBuf")[0], dtype=np.uint32)
bump = eng.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8]
names = {1: "FILL", 3: "SOLID", 5: "COLOR", 6: "LIN_GRAD", 7: "RAD_GRAD", 8: "SWEEP_GRAD", 9: "IMAGE", 10: "BEGIN_CLIP", 11: "END_CLIP", 12: "JUMP"}
sizes = {1: 4, 3: 1, 5: 5, 6: 3, 7: 3, 8: 3, 9: 2, 10: 1, 11: 3}
cnt = collections.Counter()
words = 0
segs = 0
ntiles = cfg["width_in_tiles"] * cfg["height_in_tiles"]
for t in range(ntiles):
    ix = t * 64 + 1
    words += 1
    while True:
        tag = int(ptcl[ix])
        if tag == 0:
            words += 1
            break
        cnt[names[tag]] += 1
        if tag == 12:
            words += 2
            ix = int(ptcl[ix + 1])
            continue
        if tag == 1:
            segs += int(ptcl[ix + 1]) >> 1
        words += sizes[tag]
        ix += sizes[tag]
out = {"scene": which, "paths": n, "size": size, "tiles": ntiles, "live_ptcl_words": words, "fill_segments": segs,
       "commands": dict(cnt), "per_tile": {k: round(v / ntiles, 2) for k, v in cnt.items()},
       "live_words_per_tile": round(words / ntiles, 1), "segments_per_tile": round(segs / ntiles, 1), "bump_segments": int(bump[5])}
print(json.dumps(out))
