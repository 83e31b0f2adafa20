"""The transcendental-free DECISION of flatten's subdivision test (jello_amd/csrc/flatten_fast.h) against the oracle's pinned
sequence, on the CPU with the very IEEE binary32 operations the device executes (tests/native/flatten_fast_check.cpp): at
every node of the subdivision trees of five families of cubics, and on adversarial operands, |v~ - v| <= delta and no
decision contradicts `err * scale <= tol` (flatten.wgsl:401).  The kernel's use of it is covered by the GPU parity suite
(the oracle evaluates the pinned sequence for every node) and by tools/soak_flatten_fast.py."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_decision_never_contradicts_the_pinned_sequence():
    src = os.path.join(ROOT, "tests", "native", "flatten_fast_check.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "ffcheck")
        fma = ["-mfma"] if "fma" in open("/proc/cpuinfo").read().split() else []
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-strict-aliasing", "-fopenmp",
                               "-w", "-o", exe, src] + fma)
        env = dict(os.environ, OMP_NUM_THREADS="4")
        r = subprocess.run([exe, "60000", "1500000", "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout
    assert r.stdout.strip().endswith("ok")
    # every family ran, nothing contradicted, and the C3-like family falls back to the pinned sequence for < 1 % of its nodes
    rows = re.findall(r"^(.+?)\s+nodes\s+(\d+).*?unsure\s+([\d.]+)%.*?contradictions (\d+)\s+bound violations (\d+)", r.stdout, re.M)
    assert len(rows) == 6
    for name, nodes, unsure, contra, viol in rows:
        assert int(nodes) > 10000 and int(contra) == 0 and int(viol) == 0, name
    assert float(rows[0][2]) < 1.0
    m = re.search(r"vs the pinned atan2_ = ([\d.e+-]+) \(FF_ET = ([\d.e+-]+)\)", r.stdout)
    assert m and float(m.group(1)) <= float(m.group(2))
