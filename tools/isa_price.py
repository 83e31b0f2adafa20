"""Prices the vector instructions of one kernel of an ISA listing (hipcc -S) with the issue costs MEASURED on gfx950 by
tools/ubench/valu3.hip / valu4.hip (profiles/r03_ubench_issue_rates*.txt): cycles of its SIMD per wave64 instruction with two or
more waves resident.  Static mix only (every instruction counted once, whatever its trip count).
   python3 tools/isa_price.py /tmp/asm/fine.s 'k_fine_areaILi0ELb0ELb0E'  ->  one JSON line
Classes: 2.2 cycles: v_add/sub/mul_f32, v_fmac_f32, v_and/or/xor_b32, v_lshrrev_b32, v_add/sub_u32, v_mov_b32 -- with VGPR, inline-constant
or literal sources only (an SGPR source makes it 4.1); 8.1: v_rcp/rsq/sqrt/exp/log/sin/cos_f32, v_readlane_b32 with an SGPR lane select,
v_permlane32_swap; 16: v_rcp/rsq/sqrt_f64; 4.2: everything else (min/max/med3, fma with three sources, compares, selects, conversions,
shifts left, bit counts, DPP, packed f32 -- two results per instruction --, f64 add/mul/fma, lane reads with a constant lane)."""
import collections
import json
import re
import sys

FAST = {"v_add_f32", "v_mul_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_add_u32",
        "v_sub_u32", "v_subrev_u32", "v_mov_b32"}
SLOW8 = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32", "v_permlane32_swap_b32"}
SLOW16 = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}


def price(line):
    toks = line.strip().split(None, 1)
    op, args = toks[0], (toks[1] if len(toks) > 1 else "")
    if not op.startswith("v_"):
        return None
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in SLOW16:
        return 16.2
    if base in SLOW8:
        return 8.1
    if base == "v_readlane_b32":
        return 8.1 if re.search(r",\s*s\d+\s*$", args) else 4.2
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return 4.2
    if base in FAST:
        srcs = args.split(",")[1:]
        if any(re.match(r"\s*-?\|?(s\d+|s\[|vcc|exec|m0)", s) for s in srcs):
            return 4.1
        return 2.2
    return 4.2


def main():
    lines = open(sys.argv[1]).read().splitlines()
    pat = re.compile(sys.argv[2])
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat.search(l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    n = collections.Counter()
    cyc = 0.0
    for l in lines[start:end]:
        s = l.strip()
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        c = price(s)
        op = s.split()[0]
        if c is not None:
            n["valu"] += 1
            n["valu_fast"] += 1 if c < 3 else 0
            cyc += c
        elif op.startswith("s_"):
            n["salu"] += 1
        elif op.startswith("ds_"):
            n["lds"] += 1
    print(json.dumps({"kernel": sys.argv[2], "static_valu": n["valu"], "static_valu_fast_class": n["valu_fast"], "static_salu": n["salu"],
                      "static_lds": n["lds"], "valu_cycles_per_inst_static_mix": round(cyc / max(n["valu"], 1), 2)}))


if __name__ == "__main__":
    main()
