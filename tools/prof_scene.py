"""Replay one of tools/shape_scenes.py's cases 20 times, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_scene.py <name>`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import jello_amd
from shape_scenes import select, big_buffers

eng = jello_amd.Engine(0)
name, mk = select(sys.argv[1:2])[0]
s, p = mk()
p.bump = big_buffers()
rec, bump, attempts = eng.render(s, p, robust=True, retain=True)
torch.cuda.synchronize()
for _ in range(20):
    eng.run(rec, jello_amd.engine.RUN_DISPATCHES)
torch.cuda.synchronize()
