"""oracle_engine -- TEST INFRASTRUCTURE: replays a renderer Recording on the CPU oracle
(oracle/liboracle.so), the way the reference's engine runs its Go "CPU shaders" when UseCPU is set
(engine/wgpu_engine/wgpu.go:454-471, lib.go:56-101).  Buffers are numpy arrays keyed by ResourceID.

Used only as the checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OBuf(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("n", ctypes.c_uint64)]


def build():
    r = subprocess.run(["make", "-C", _HERE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout)
        raise RuntimeError("oracle build failed")
    return os.path.join(_HERE, "liboracle.so")


_lib = None


def lib():
    global _lib
    if _lib is None:
        # JELLO_ORACLE_LIB: the sanitizer build of tools/sanitize_cpu.sh
        p = os.environ.get("JELLO_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(p):
            build()
        _lib = ctypes.CDLL(p)
        _lib.oracle_dispatch.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(OBuf), ctypes.c_int]
    return _lib


class OracleEngine:
    """Replay of a Recording with the canonical (sequential) allocation order."""

    def __init__(self, poison=0xCD):
        self.bufs = {}     # id -> np.uint8 array
        self.images = {}   # id -> (np.uint8 array, w, h, format)
        self.poison = poison
        self.stage_seconds = {}
        self.L = lib()

    def _buf(self, bid, size):
        b = self.bufs.get(bid)
        if b is None or b.nbytes < size:
            # Buffers with no prior writer are NOT zeroed in the reference (pool reuse);
            # poison them so any reliance on zero-init shows up.
            b = np.full(max(size, 4), self.poison, dtype=np.uint8)
            self.bufs[bid] = b
        return b

    def run(self, recording, stop_after=None, only=None):
        """only = a stage name: nothing but the dispatches of that stage, on the buffers an earlier run() left (tests that change a
        buffer between two stages)."""
        from jello_amd.engine import CMD, STAGE_NAMES as _NAMES
        pending_clear = set()
        for c in recording.commands():
            k = c["kind"]
            if only is not None and not (k in (CMD.DISPATCH, CMD.DISPATCH_INDIRECT) and _NAMES[c["shader"]] == only):
                continue
            if k in (CMD.UPLOAD, CMD.UPLOAD_UNIFORM):
                self.bufs[c["buf_id"]] = np.frombuffer(c["data"], dtype=np.uint8).copy()
            elif k == CMD.UPLOAD_IMAGE:
                self.images[c["img_id"]] = (np.frombuffer(c["data"], dtype=np.uint8).copy(), c["img_w"], c["img_h"], c["img_format"])
            elif k == CMD.CLEAR:
                if c["buf_id"] in self.bufs:
                    b = self.bufs[c["buf_id"]]
                    end = b.nbytes if c["size"] < 0 else c["offset"] + c["size"]
                    b[c["offset"]:end] = 0
                else:
                    pending_clear.add(c["buf_id"])
            elif k in (CMD.DISPATCH, CMD.DISPATCH_INDIRECT):
                arr = []
                keep = []
                for b in c["bindings"]:
                    if b["kind"] == 1:
                        fresh = b["id"] not in self.bufs
                        nb = self._buf(b["id"], b["size"])
                        if fresh and b["id"] in pending_clear:
                            nb[:] = 0
                            pending_clear.discard(b["id"])
                        arr.append((nb.ctypes.data, nb.nbytes))
                    elif b["kind"] == 2:
                        if b["id"] not in self.images:
                            bpp = 8 if b["format"] == 3 else 4
                            self.images[b["id"]] = (np.zeros(max(1, b["width"] * b["height"] * bpp), dtype=np.uint8), b["width"], b["height"], b["format"])
                        im = self.images[b["id"]][0]
                        arr.append((im.ctypes.data, im.nbytes))
                    else:
                        # image array -> (descriptor table, texel blob) as two oracle bindings
                        table = np.zeros((max(1, len(b["ids"])), 2), dtype=np.uint64)
                        blobs = []
                        off = 0
                        for i, iid in enumerate(b["ids"]):
                            im, w, h, fmt = self.images.get(iid, (np.zeros(4, np.uint8), 1, 1, 0))
                            table[i, 0] = off
                            hh = int(h) | ((1 << 31) if fmt == 1 else 0)  # JL_RGBA8_SRGB: texels decode to linear
                            table[i, 1] = np.uint64(w) | (np.uint64(hh) << np.uint64(32))
                            blobs.append(im)
                            off += im.nbytes // 4
                        blob = np.concatenate(blobs) if blobs else np.zeros(4, np.uint8)
                        keep += [table, blob]
                        arr.append((table.ctypes.data, table.nbytes))
                        arr.append((blob.ctypes.data, blob.nbytes))
                obufs = (OBuf * len(arr))(*[OBuf(p, n) for p, n in arr])
                if k == CMD.DISPATCH_INDIRECT:
                    ind = self.bufs[c["buf_id"]].view(np.uint32)
                    gx, gy, gz = int(ind[c["offset"] // 4]), 1, 1
                else:
                    gx, gy, gz = c["wg"]
                t0 = time.perf_counter()
                rc = self.L.oracle_dispatch(c["shader"], gx, gy, gz, obufs, len(arr))
                dt = time.perf_counter() - t0
                if rc != 0:
                    raise RuntimeError("oracle: stage %d not implemented" % c["shader"])
                from jello_amd.engine import STAGE_NAMES
                name = STAGE_NAMES[c["shader"]]
                self.stage_seconds[name] = self.stage_seconds.get(name, 0.0) + dt
                if stop_after is not None and name == stop_after:
                    return
            # DOWNLOAD / FREE_*: buffers stay available for inspection

    def get(self, recording, name, dtype=np.uint8):
        bid, _ = recording.buffer(name)
        return self.bufs[bid].view(dtype)

    def target(self, recording):
        t = recording.target
        im = self.images[t["id"]][0]
        return im.view(np.uint16).reshape(t["height"], t["width"], 4)
