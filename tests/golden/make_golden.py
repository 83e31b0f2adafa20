#!/usr/bin/env python3
"""Generates tests/golden/c1_oracle.json: hashes of every pipeline buffer the CPU oracle produces for
config C1 (+ a 32x32 crop of the image).  These are ORACLE-GENERATED fixtures (regression pins for the
oracle and, on the GPU box, for the HIP path); they are not outputs of the reference, which cannot be
executed in this environment.  Rounding policy for transcendentals: binary64 evaluation
rounded once to binary32 (oracle/omath.h).  Run from the repo root: python tests/golden/make_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import jello_amd  # noqa: E402
from jello_amd import scenes  # noqa: E402
from oracle.oracle_engine import OracleEngine  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fixture():
    s, p = scenes.scene_c1()
    rec = jello_amd.Host().record(s, p)
    o = OracleEngine()
    o.run(rec)
    cfg = rec.config
    bump = o.get(rec, "bumpBuf", np.uint32)[:8]
    n_lines, n_segc, n_seg, n_tile = int(bump[7]), int(bump[4]), int(bump[5]), int(bump[3])
    nd = cfg["n_drawobj"]
    out = {"bump": [int(x) for x in bump]}
    out["lines"] = digest(o.get(rec, "linesBuf", np.uint32).reshape(-1, 6)[:n_lines])
    out["lines_head"] = o.get(rec, "linesBuf", np.uint32).reshape(-1, 6)[:8].tolist()
    out["tag_monoids"] = digest(o.get(rec, "tagmonoidBuf", np.uint32)[:256 * 5])
    out["path_bboxes"] = o.get(rec, "pathBboxBuf", np.int32).reshape(-1, 6)[:cfg["n_path"]].tolist()
    out["draw_monoids"] = o.get(rec, "drawMonoidBuf", np.uint32).reshape(-1, 4)[:nd].tolist()
    out["paths"] = o.get(rec, "pathBuf", np.uint32).reshape(-1, 8)[:nd, :5].tolist()
    out["seg_counts"] = digest(o.get(rec, "segCountsBuf", np.uint32).reshape(-1, 2)[:n_segc])
    out["tiles"] = digest(o.get(rec, "tileBuf", np.uint32).reshape(-1, 2)[:n_tile])
    out["segments"] = digest(o.get(rec, "segmentsBuf", np.uint32).reshape(-1, 6)[:n_seg, :5])
    ptcl = o.get(rec, "ptclBuf", np.uint32)
    # tile (3,2) of the 32x32 grid: inside the rect -> [blend_ix=0][SOLID][COLOR 1 0 0 1][END]
    t = 2 * 32 + 3
    out["ptcl_tile_3_2"] = ptcl[t * 64:t * 64 + 8].tolist()
    out["ptcl_tile_0_0"] = ptcl[0:12].tolist()
    img = o.target(rec)
    out["image"] = digest(img)
    out["image_crop_y296_x40_32x32"] = img[296:328, 40:72].reshape(-1).tolist()
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c1_oracle.json")
    json.dump(fixture(), open(path, "w"))
    print("wrote", path)
