"""Workload statistics of the flatten stage (checker-side instrumentation; not part of the product path)."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import jello_amd
from jello_amd import scenes
from jello_amd.scene import RenderParams
from oracle import oracle_engine

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    which = sys.argv[3] if len(sys.argv) > 3 else "c3"
    if which == "c3":
        scene, params = scenes.scene_c3(n, size)
    elif which == "c4":
        scene, params = scenes.scene_c4(n, size)
    elif which == "c2":
        scene, params = scenes.scene_c2(n, size)
    elif which == "large":
        scene, params = scenes.scene_large_shapes()
    else:
        scene, params = scenes.scene_fuzz(int(which), extreme=True)
    host = jello_amd.Host()
    rec = host.record(scene, params)
    lib = oracle_engine.lib()
    out = (ctypes.c_uint64 * 32)()
    lib.oracle_flatten_stats(out, 1)
    lib.oracle_flatten_depth((ctypes.c_uint64 * 20)(), 1)
    eng = oracle_engine.OracleEngine()
    eng.run(rec)
    lib.oracle_flatten_stats(out, 1)
    v = list(out)
    print("jobs", v[0], "attempts", v[1], "lines", v[2], "pieces", v[3])
    print("attempts/job %.2f lines/piece %.2f pieces/job %.2f" % (v[1] / v[0], v[2] / v[3], v[3] / v[0]))
    print("hist attempts/job:", v[4:32])
    d = (ctypes.c_uint64 * 20)()
    lib.oracle_flatten_depth(d, 1)
    print("hist deepest piece per job (dt = 2^-depth):", list(d))

if __name__ == "__main__":
    main()
