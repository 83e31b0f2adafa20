import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import jello_amd
from jello_amd import scenes, BumpSizes
from jello_amd.engine import RUN_UPLOADS, RUN_DISPATCHES
eng = jello_amd.Engine(0)
s,p = scenes.scene_c4(3000, 512)
p.bump = BumpSizes(lines=1<<18, seg_counts=1<<19, segments=1<<19, tiles=1<<19, ptcl=1<<21, bin_data=1<<18, blend_spill=1<<16)
rec = jello_amd.Host().record(s,p)
cfg = rec.config
eng.run(rec, RUN_UPLOADS|RUN_DISPATCHES); eng.sync()
pt = eng.download(rec.buffer("ptclBuf")[0], dtype=np.uint32).copy()
bump = eng.download(rec.buffer("bumpBuf")[0], dtype=np.uint32)[:8].copy()
print(sys.argv[1], "bump", bump, "ptcl words", pt.size)
np.save("/root/repo/gpurun_out/dbg_ptcl_%s.npy" % sys.argv[1], pt)
