"""CPU: libjello_hip.so loads without a GPU and exports every symbol include/jello_hip.h declares;
without a device the engine fails loudly (no fallback)."""
import ctypes
import os
import re

import pytest

import jello_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "jello_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(jh_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported(built):
    lib = ctypes.CDLL(jello_amd.lib_paths()["hip"])
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s


def test_stage_enum_matches_fullshaders_order(built):
    lib = ctypes.CDLL(jello_amd.lib_paths()["hip"])
    lib.jh_stage_name.restype = ctypes.c_char_p
    names = [lib.jh_stage_name(i).decode() for i in range(22)]
    assert names == jello_amd.STAGE_NAMES  # renderer/render.go:17-43


def test_formats_header_sizes():
    """include/jello_formats.h carries static_asserts; compile it as C and C++."""
    import subprocess
    import tempfile
    for comp, ext in (("gcc", "c"), ("g++", "cpp")):
        with tempfile.NamedTemporaryFile("w", suffix="." + ext, delete=False) as f:
            f.write('#include "jello_formats.h"\n#include "jello_hip.h"\nint main(void){return sizeof(JlConfig)==100?0:1;}\n')
        out = f.name + ".bin"
        subprocess.check_call([comp, "-I", os.path.join(ROOT, "include"), f.name, "-o", out])
        assert subprocess.call([out]) == 0
        os.unlink(f.name)
        os.unlink(out)


def test_no_gpu_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        jello_amd.Engine(0)
