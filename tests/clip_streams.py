"""Clip streams for the stage-level tests of clip_reduce / clip_leaf (orig/clip_reduce.wgsl:24-67, orig/clip_leaf.wgsl:80-207):
generators, a by-definition reference (a Python list as the stack; shares no code with oracle/ or the kernels) and the helpers
that run the two stages on the oracle.

A stream is a list of records ("begin", box) / ("end",): BeginClip of a path with that integer box, EndClip.  `pack` lays it
out as the buffers the two dispatches bind: clip_inp {ix, path_ix}, path_bboxes, draw_monoids.
"""
import ctypes

import numpy as np

EVERYTHING = (-1e9, -1e9, 1e9, 1e9)
BLOCK = 256
ST_CLIP_REDUCE, ST_CLIP_LEAF = 9, 10  # jh_stage = FullShaders field order (render.go:17-43)


def meet(a, b):  # shared/bbox.wgsl:21-23
    return (max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3]))


def pack(stream):
    """Draw object i = record i (every record is a draw object of its own: BeginClip i draws path i)."""
    n = len(stream)
    clip_inp = np.zeros((n, 2), np.int32)
    path_bboxes = np.zeros((n, 6), np.int32)
    draw_monoids = np.zeros((n, 4), np.uint32)
    for i, r in enumerate(stream):
        clip_inp[i, 0] = i
        draw_monoids[i] = (0x70000000 + i, i + 1, 5 * i + 3, 0x600 + i)  # path_ix (EndClip: to be replaced), clip_ix, scene_offset, info_offset
        if r[0] == "begin":
            clip_inp[i, 1] = i
            path_bboxes[i, :4] = r[1]
            draw_monoids[i, 0] = i
        else:
            clip_inp[i, 1] = ~i
    cfg = np.zeros(25, np.uint32)
    cfg[8], cfg[9], cfg[10] = n, n, n  # layout.n_drawobj, n_path, n_clip
    return cfg, clip_inp, path_bboxes, draw_monoids


def by_definition(stream):
    """(clip_bboxes f32 [n,4], draw_monoids after the fix-up) from the definition: a stack of open layers."""
    cfg, clip_inp, path_bboxes, dm = pack(stream)
    out = np.zeros((len(stream), 4), np.float32)
    stack = []  # (record, box of the layer met with all layers under it)
    for i, r in enumerate(stream):
        if r[0] == "begin":
            box = meet(stack[-1][1] if stack else EVERYTHING, tuple(float(v) for v in r[1]))
            stack.append((i, box))
            out[i] = box
        else:
            if not stack:  # nothing to close: the record is left alone (the WGSL would index clip_inp[-1])
                out[i] = EVERYTHING
                continue
            opener, _ = stack.pop()
            dm[i, 0] = clip_inp[opener, 1]
            dm[i, 2] = dm[clip_inp[opener, 0], 2]
            out[i] = stack[-1][1] if stack else EVERYTHING
    return out, dm


def block_summaries(stream):
    """What clip_reduce leaves per full block of 256 records: (closes, opens) and the open BeginClips, bottom first, as
    (record, box) -- from the definition."""
    n_blocks = (len(stream) - 1) // BLOCK if stream else 0
    out = []
    for b in range(n_blocks):
        closes, stack = 0, []
        for i in range(b * BLOCK, (b + 1) * BLOCK):
            r = stream[i]
            if r[0] == "begin":
                stack.append((i, tuple(float(v) for v in r[1])))
            elif stack:
                stack.pop()
            else:
                closes += 1
        out.append(((closes, len(stack)), stack))
    return out


# ---- generators ---------------------------------------------------------------------------------------------------------
def staircase(depth):
    """`depth` nested layers, then `depth` EndClips.  Layer i narrows x when i is even and y when i is odd, so the box of
    layer i in closed form is (xe, yo, 4000 - xe, 4000 - yo) with xe = the largest even number <= i, yo = the largest odd one
    (0 while i = 0).  EndClip number j (closing layer depth - 1 - j) gets the box of layer depth - 2 - j."""
    s = []
    for i in range(depth):
        s.append(("begin", (i, 0, 4000 - i, 4000) if i % 2 == 0 else (0, i, 4000, 4000 - i)))
    s += [("end",)] * depth
    return s


def staircase_answer(depth):
    def layer(i):
        if i < 0:
            return EVERYTHING
        xe = i - (i % 2)
        yo = i if i % 2 else i - 1
        return (float(xe), float(yo), 4000.0 - xe, 4000.0 - yo) if yo > 0 else (float(xe), 0.0, 4000.0 - xe, 4000.0)
    boxes = [layer(i) for i in range(depth)] + [layer(depth - 2 - j) for j in range(depth)]
    partner = [None] * depth + [depth - 1 - j for j in range(depth)]
    return np.array(boxes, np.float32), partner


def random_stream(rng, n, p_begin=0.5, extra_ends=0, max_depth=None):
    """Random walk; `extra_ends` EndClips with nothing to close are sprinkled in front, open layers stay open at the end."""
    s, depth = [], 0
    for _ in range(extra_ends):
        s.append(("end",))
    while len(s) < n:
        go_up = rng.random() < p_begin
        if max_depth is not None and depth >= max_depth:
            go_up = False
        if go_up or depth == 0:
            x0, y0 = int(rng.integers(-50, 3000)), int(rng.integers(-50, 3000))
            s.append(("begin", (x0, y0, x0 + int(rng.integers(0, 2500)), y0 + int(rng.integers(0, 2500)))))
            depth += 1
        else:
            s.append(("end",))
            depth -= 1
    return s[:n]


def sawtooth(rng, n, up, down):
    """`up` BeginClips, `down` EndClips, repeated: the stack grows by up - down per tooth (deep stacks, every block reaches
    into several earlier ones)."""
    s = []
    while len(s) < n:
        for _ in range(up):
            x0, y0 = int(rng.integers(0, 100)), int(rng.integers(0, 100))
            s.append(("begin", (x0, y0, 4000 - int(rng.integers(0, 100)), 4000 - int(rng.integers(0, 100)))))
        s += [("end",)] * down
    return s[:n]


def streams():
    """(name, stream) pairs: every shape the kernels distinguish."""
    rng = np.random.default_rng(0x636c6970)
    out = [("one_layer", staircase(1)), ("staircase_5", staircase(5)), ("staircase_64", staircase(64)),
           ("staircase_130", staircase(130)),         # 260 records: the EndClips of block 1 close layers of block 0
           ("staircase_300", staircase(300)),         # a stack deeper than a block (the WGSL's 256-entry window would not do)
           ("staircase_1000", staircase(1000))]
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1000, 4099):
        out.append(("random_%d" % n, random_stream(rng, n)))
    out.append(("random_shallow_20000", random_stream(rng, 20000, 0.5, max_depth=3)))
    out.append(("random_rising_9000", random_stream(rng, 9000, 0.6)))
    out.append(("random_falling_9000", staircase(700)[:700] + random_stream(rng, 8300, 0.42)))
    out.append(("unbalanced_front", random_stream(rng, 3000, 0.5, extra_ends=300)))
    out.append(("sawtooth_70_60", sawtooth(rng, 6000, 70, 60)))
    out.append(("sawtooth_300_299", sawtooth(rng, 8000, 300, 299)))
    out.append(("all_begin_3000", [("begin", (i % 97, i % 89, 4000 - i % 83, 4000 - i % 79)) for i in range(3000)]))
    out.append(("all_end_700", [("end",)] * 700))
    out.append(("many_blocks_70000", random_stream(rng, 70000, 0.5, max_depth=40)))  # more than 256 blocks
    return out


# ---- the oracle's two stages on a packed stream ---------------------------------------------------------------------------
def run_oracle(stream):
    """Returns (clip_bboxes f32 [n,4], draw_monoids u32 [n,4], reduced u32 [blocks,2], clip_els raw u32 [n,8])."""
    from oracle.oracle_engine import OBuf, lib
    L = lib()
    cfg, clip_inp, path_bboxes, dm = pack(stream)
    n = len(stream)
    n_red = (n - 1) // BLOCK if n else 0
    reduced = np.full((max(n_red, 1), 2), 0xCDCDCDCD, np.uint32)
    els = np.full((max(n, 1), 8), 0xCDCDCDCD, np.uint32)
    out = np.full((max(n, 1), 4), 0xCDCDCDCD, np.uint32)

    def bufs(*arrs):
        return (OBuf * len(arrs))(*[OBuf(a.ctypes.data, a.nbytes) for a in arrs])
    if n_red:
        assert L.oracle_dispatch(ST_CLIP_REDUCE, n_red, 1, 1, bufs(clip_inp, path_bboxes, reduced, els), 4) == 0
    n_leaf = (n + BLOCK - 1) // BLOCK
    if n_leaf:
        assert L.oracle_dispatch(ST_CLIP_LEAF, n_leaf, 1, 1, bufs(cfg, clip_inp, path_bboxes, reduced, els, dm, out), 7) == 0
    return out[:n].view(np.float32), dm, reduced[:n_red], els
