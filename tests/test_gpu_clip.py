"""-m gpu: k_clip_reduce / k_clip_leaf through jh_dispatch on clip streams built directly as buffers (tests/clip_streams.py),
against the definition and the oracle: streams that span blocks, stacks deeper than a block (beyond the 256 entries the WGSL
can see), more than 256 blocks, EndClips with nothing to close.  `reduced` and the defined words of `clip_els` -- the buffers
between the two dispatches -- are compared as well."""
import ctypes

import numpy as np
import pytest

import clip_streams as C

pytestmark = pytest.mark.gpu
STREAMS = C.streams()
ID0 = 0x7e57c11b0000


class Binding(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("count", ctypes.c_uint32), ("id", ctypes.c_uint64), ("ids", ctypes.POINTER(ctypes.c_uint64))]


def run_hip(engine, stream, poison=0xA7):
    hip, ctx = engine.hip, engine.ctx
    hip.jh_dispatch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(Binding), ctypes.c_int]
    hip.jh_free.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    hip.jh_upload.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    cfg, clip_inp, path_bboxes, dm = C.pack(stream)
    n = len(stream)
    n_red = (n - 1) // C.BLOCK
    reduced = np.full((max(n_red, 1), 2), poison * 0x01010101, np.uint32)
    els = np.full((n, 8), poison * 0x01010101, np.uint32)
    out = np.full((n, 4), poison * 0x01010101, np.uint32)
    names = ["cfg", "inp", "pb", "red", "els", "dm", "out"]
    arrs = dict(zip(names, [cfg, clip_inp, path_bboxes, reduced, els, dm, out]))
    ids = {k: ID0 + i for i, k in enumerate(names)}
    try:
        for k in names:
            assert hip.jh_upload(ctx, ids[k], arrs[k].ctypes.data, arrs[k].nbytes) == 0

        def bind(*ks):
            return (Binding * len(ks))(*[Binding(1, 0, ids[k], None) for k in ks])
        if n_red:
            assert hip.jh_dispatch(ctx, C.ST_CLIP_REDUCE, n_red, 1, 1, bind("inp", "pb", "red", "els"), 4) == 0, hip.jh_last_error(ctx)
        assert hip.jh_dispatch(ctx, C.ST_CLIP_LEAF, (n + C.BLOCK - 1) // C.BLOCK, 1, 1, bind("cfg", "inp", "pb", "red", "els", "dm", "out"), 7) == 0, \
            hip.jh_last_error(ctx)
        engine.sync()
        got = {k: engine.download(ids[k], dtype=np.uint32).copy() for k in ("red", "els", "dm", "out")}
    finally:
        for k in names:
            hip.jh_free(ctx, ids[k])
    return (got["out"][:4 * n].reshape(n, 4), got["dm"][:4 * n].reshape(n, 4), got["red"][:2 * max(n_red, 1)].reshape(-1, 2)[:n_red],
            got["els"][:8 * n].reshape(n, 8))


@pytest.mark.parametrize("name,stream", STREAMS, ids=[s[0] for s in STREAMS])
def test_clip_stages_on_streams(engine, name, stream):
    want_boxes, want_dm = C.by_definition(stream)
    boxes, dm, reduced, els = run_hip(engine, stream)
    bad = np.flatnonzero((boxes != want_boxes.view(np.uint32)).any(axis=1))
    assert bad.size == 0, "clip_bboxes differ at %d records, first %d: gpu %s want %s" % (
        bad.size, bad[0], boxes[bad[0]].view(np.float32), want_boxes[bad[0]])
    bad = np.flatnonzero((dm != want_dm).any(axis=1))
    assert bad.size == 0, "draw_monoids differ at %d records, first %d: gpu %s want %s" % (bad.size, bad[0], dm[bad[0]], want_dm[bad[0]])
    o_boxes, o_dm, o_reduced, o_els = C.run_oracle(stream)
    assert np.array_equal(boxes, o_boxes.view(np.uint32)) and np.array_equal(dm, o_dm)
    assert np.array_equal(reduced, o_reduced), "reduced (the nesting summaries between the two dispatches)"
    for b in range(reduced.shape[0]):  # clip_els: the first `opens` records of a block's 256 are defined: parent_ix and bbox
        k = int(reduced[b, 1])
        sl = slice(b * C.BLOCK, b * C.BLOCK + k)
        assert np.array_equal(els[sl][:, [0, 4, 5, 6, 7]], o_els[sl][:, [0, 4, 5, 6, 7]]), "clip_els of block %d" % b


def test_staircase_closed_form_on_the_gpu(engine):
    """The hand-derived answer (tests/clip_streams.staircase_answer) held against the HIP buffers directly."""
    for depth in (5, 130, 300, 1000):
        want, partner = C.staircase_answer(depth)
        boxes, dm, _, _ = run_hip(engine, C.staircase(depth))
        assert np.array_equal(boxes, want.view(np.uint32)), depth
        assert [int(v) for v in dm[depth:, 0]] == partner[depth:]


def test_leaf_alone_on_the_oracles_intermediates(engine):
    """clip_leaf fed with the ORACLE's `reduced` / `clip_els` (the per-stage swap the boundary allows): it may depend on nothing
    but its bindings."""
    hip, ctx = engine.hip, engine.ctx
    hip.jh_dispatch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(Binding), ctypes.c_int]
    hip.jh_upload.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    hip.jh_free.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    stream = dict(STREAMS)["sawtooth_70_60"]
    o_boxes, o_dm, o_reduced, o_els = C.run_oracle(stream)
    cfg, clip_inp, path_bboxes, dm = C.pack(stream)
    n = len(stream)
    out = np.zeros((n, 4), np.uint32)
    arrs = [cfg, clip_inp, path_bboxes, np.ascontiguousarray(o_reduced), np.ascontiguousarray(o_els), dm, out]
    ids = [ID0 + 0x100 + i for i in range(7)]
    try:
        for i, a in zip(ids, arrs):
            assert hip.jh_upload(ctx, i, a.ctypes.data, a.nbytes) == 0
        b = (Binding * 7)(*[Binding(1, 0, i, None) for i in ids])
        assert hip.jh_dispatch(ctx, C.ST_CLIP_LEAF, (n + C.BLOCK - 1) // C.BLOCK, 1, 1, b, 7) == 0
        engine.sync()
        got = engine.download(ids[6], dtype=np.uint32)[:4 * n].reshape(n, 4)
        got_dm = engine.download(ids[5], dtype=np.uint32)[:4 * n].reshape(n, 4)
    finally:
        for i in ids:
            hip.jh_free(ctx, i)
    assert np.array_equal(got, o_boxes.view(np.uint32)) and np.array_equal(got_dm, o_dm)
