"""CPU: the oracle reproduces the committed C1 fixtures (tests/golden/c1_oracle.json, made by
tests/golden/make_golden.py) -- a regression pin; on the GPU box test_gpu_parity compares the HIP path
with the same oracle."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_c1_oracle_fixture(built):
    import make_golden
    want = json.load(open(os.path.join(HERE, "golden", "c1_oracle.json")))
    got = make_golden.fixture()
    for k in want:
        assert got[k] == want[k], k
    # hand-checkable pieces of the fixture
    assert want["ptcl_tile_3_2"][:7] == [0, 3, 5, 0x3f800000, 0, 0, 0x3f800000]      # blend_ix, SOLID, COLOR(1,0,0,1)
    assert want["ptcl_tile_3_2"][7] == 0                                               # END
    assert want["bump"][0] == 0 and want["bump"][7] == 70
