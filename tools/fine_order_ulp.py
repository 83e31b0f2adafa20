"""What would north_star's tolerance ("fine RGBA within 1 ULP per channel") leave of the image if the f32 order of
fill_path's sum (fine.wgsl:832-864) were given up?  CPU experiment on the oracle (oracle_set_fine_order): the same scenes
rendered with the WGSL's association, with a per-fill segmented sum (backdrop added last) and with the segments reversed;
reports per scene set the number of channel values that differ and the largest f16 ULP distance, split by the pixel's
alpha (a difference in a pixel whose alpha is 0 in f16 is invisible but counts for a per-channel bound all the same).

    python tools/fine_order_ulp.py [n_fuzz_scenes]        ->  one JSON document on stdout
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jello_amd  # noqa: E402
from jello_amd import scenes  # noqa: E402
from oracle import oracle_engine  # noqa: E402


def ordered(a):
    a = a.astype(np.int32)
    return np.where(a & 0x8000, 0x8000 - (a & 0x7fff), a)


def render(scene, params, order):
    L = oracle_engine.lib()
    L.oracle_set_fine_order(order)
    try:
        rec = jello_amd.Host().record(scene, params)
        o = oracle_engine.OracleEngine()
        o.run(rec)
        return o.target(rec).copy()
    finally:
        L.oracle_set_fine_order(0)


def compare(ref, img):
    nan = lambda a: (a & 0x7fff) > 0x7c00  # noqa: E731
    ok = ~(nan(ref) | nan(img))
    d = np.abs(ordered(ref) - ordered(img)) * ok
    alpha0 = (ref[..., 3:4] & 0x7fff) == 0
    vis = d * ~alpha0
    return {"differing": int((d != 0).sum()), "max_ulp": int(d.max()), "differing_visible": int((vis != 0).sum()), "max_ulp_visible": int(vis.max()),
            "over_1_ulp": int((d > 1).sum()), "over_1_ulp_visible": int((vis > 1).sum())}


def main():
    n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    L = oracle_engine.lib()
    L.oracle_set_threads(os.cpu_count() or 1)
    sets = {"c3_density_1024": [scenes.scene_c3(6250, 1024)], "c1": [scenes.scene_c1()], "c2_substitute": [scenes.scene_c2()],
            "fuzz": [scenes.scene_fuzz(1000 + i)[:2] for i in range(n_fuzz)]}
    out = {"what": __doc__.split("\n\n")[0], "orders": {"1": "terms of a fill summed on their own, backdrop added last", "2": "segments of a fill reversed"}, "sets": {}}
    for name, items in sets.items():
        agg = {}
        for s, p in items:
            ref = render(s, p, 0)
            for order in (1, 2):
                c = compare(ref, render(s, p, order))
                a = agg.setdefault(str(order), {"scenes": 0, "channel_values": 0})
                a["scenes"] += 1
                a["channel_values"] += int(ref.size)
                for k, v in c.items():
                    a[k] = max(a.get(k, 0), v) if k.startswith("max") else a.get(k, 0) + v
        out["sets"][name] = agg
        print(name, json.dumps(agg), file=sys.stderr)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
