"""clip_reduce / clip_leaf at stage level on the CPU: the oracle's two stages against the definition (a Python list as the
stack, tests/clip_streams.py) and against a closed-form answer, on streams that span blocks, nest deeper than a block and
close layers that were never opened.  The same streams go through the HIP kernels in tests/test_gpu_clip.py."""
import numpy as np
import pytest

import clip_streams as C

STREAMS = C.streams()


def test_staircase_closed_form():
    """300 nested layers = 600 records in three blocks, a stack deeper than one block (hand-derived: clip_streams.staircase)."""
    for depth in (1, 5, 130, 300):
        want, partner = C.staircase_answer(depth)
        got, dm = C.by_definition(C.staircase(depth))
        assert np.array_equal(got, want)
        for j in range(depth, 2 * depth):
            assert dm[j, 0] == partner[j] and dm[j, 2] == 5 * partner[j] + 3
    # spot values, written out: layer 7 = x narrowed by layer 6, y by layer 7
    want, _ = C.staircase_answer(300)
    assert list(want[7]) == [6.0, 7.0, 3994.0, 3993.0]
    assert list(want[299]) == [298.0, 299.0, 3702.0, 3701.0]
    assert list(want[300]) == [298.0, 297.0, 3702.0, 3703.0]   # the first EndClip: what stays open is layer 298
    assert list(want[599]) == [-1e9, -1e9, 1e9, 1e9]           # the last one closes the outermost layer


@pytest.mark.parametrize("name,stream", STREAMS, ids=[s[0] for s in STREAMS])
def test_oracle_follows_the_definition(built, name, stream):
    want_boxes, want_dm = C.by_definition(stream)
    boxes, dm, reduced, els = C.run_oracle(stream)
    assert np.array_equal(boxes.view(np.uint32), want_boxes.view(np.uint32))
    assert np.array_equal(dm, want_dm)
    for b, ((closes, opens), stack) in enumerate(C.block_summaries(stream)):
        assert (int(reduced[b, 0]), int(reduced[b, 1])) == (closes, opens), "block %d" % b
        for place, (rec, box) in enumerate(stack):
            el = els[b * C.BLOCK + place]
            assert int(el[0]) == rec and list(el[4:8].view(np.float32)) == list(box), "block %d place %d" % (b, place)
