#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc ISA listing (`make -C vulkan-compute-tests_amd asm`).

  python tools/isa_blocks.py <file.s> <kernel-symbol-substring> [--loop] [--weights regions.json] [--dump out.s]

Splits the kernel into basic blocks (labels, and the fall-through blocks the compiler comments as `%bb.N`), classifies every
instruction by issue class —
    fp2   v_add/sub/mul/fmac/fmaak/fmamk_f32            (~2.3 cycles per wave64 instruction, profiles/r01_valu_microbench.txt)
    fma3  v_fma_f32 (three-address)                      (~3.7)
    int2  two-operand integer / logic / shift / v_mov    (~2.3)
    sel4  v_cndmask, v_cmp*, v_min/max/med3, v_cvt, VOP3 integer (and_or, bfi, lshl_add, alignbit, mul_lo, mbcnt ...)  (~4.2)
    trans v_rcp/rsq/sqrt/sin/cos/exp/log                 (~8.2)
    salu, lds, vmem, s_nop, branch
— and prints one row per block plus totals.  --loop restricts the census to the blocks inside the innermost `Depth=2` loop
(the bounce loop of the path tracer).  The per-class cycle prices give a static issue estimate of a block; weighted by how
often a block runs (tools/pt_region_stats.py gives executions per sample round) that is the table under profiles/."""
import argparse
import collections
import re
import sys

PRICE = {"fp2": 2.3, "fma3": 3.7, "int2": 2.3, "sel4": 4.2, "trans": 8.2, "s_nop": 4.0}
FP2 = re.compile(r"^v_(add|sub|subrev|mul|fmac|fmaak|fmamk)_f32")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|sin|cos|exp|log)_f32")
INT2 = re.compile(r"^v_(mov_b32|and_b32|or_b32|xor_b32|not_b32|lshlrev_b32|lshrrev_b32|ashrrev_i32|add_u32|sub_u32|subrev_u32|add_co_u32|addc_co_u32|sub_co_u32|accvgpr)")


def classify(op):
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if FP2.match(op):
        return "fp2"
    if op.startswith("v_fma_f32") or op.startswith("v_mad_f32"):
        return "fma3"
    if TRANS.match(op):
        return "trans"
    if INT2.match(op):
        return "int2"
    if op.startswith("v_"):
        return "sel4"
    return "other"


def kernel_lines(path, sym):
    lines = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*:", l) and sym in l:
            start = i
        elif start is not None and l.startswith("\t.end_amdhsa_kernel") or (start is not None and l.startswith(".Lfunc_end")):
            end = i
            break
    if start is None:
        sys.exit(f"kernel containing {sym!r} not found")
    return lines[start:end]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("symbol")
    ap.add_argument("--loop", action="store_true", help="only the blocks of the innermost Depth=2 loop")
    ap.add_argument("--dump", default=None, help="write the kernel's listing here")
    ap.add_argument("--blocks", action="store_true", help="print every block (default: totals only)")
    a = ap.parse_args()
    lines = kernel_lines(a.file, a.symbol)
    if a.dump:
        open(a.dump, "w").write("\n".join(lines) + "\n")
    blocks = []   # (name, in_loop, Counter)
    cur = ["entry", False, collections.Counter()]
    blocks.append(cur)
    depth2_header = None
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        m2 = re.match(r"^; %bb\.(\d+):", l)
        if m or m2:
            name = m.group(1) if m else "bb." + m2.group(1)
            cur = [name, "Depth=2" in l, collections.Counter()]
            blocks.append(cur)
            continue
        if "Depth=2" in l and l.strip().startswith(";"):   # the loop-header comment sits on its own line after the label
            cur[1] = True
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        cur[2][classify(op)] += 1
        cur[2]["op:" + re.sub(r"_e(32|64)$", "", op)] += 1
    sel = [b for b in blocks if (b[1] or not a.loop)]
    classes = ["fp2", "fma3", "int2", "sel4", "trans", "s_nop", "salu", "branch", "lds", "vmem", "other"]
    if a.blocks:
        print(f"{'block':14s} " + " ".join(f"{c:>6s}" for c in classes) + "   est.cycles")
        for name, _, c in sel:
            if sum(c[k] for k in classes):
                est = sum(c[k] * PRICE.get(k, 0.0) for k in classes)
                print(f"{name:14s} " + " ".join(f"{c[k]:6d}" for k in classes) + f"   {est:9.1f}")
    tot = collections.Counter()
    for _, _, c in sel:
        tot.update(c)
    valu = sum(tot[k] for k in ("fp2", "fma3", "int2", "sel4", "trans"))
    print(f"blocks {len(sel)}  VALU {valu}  " + "  ".join(f"{k} {tot[k]}" for k in classes if tot[k]))
    print(f"static issue estimate {sum(tot[k] * PRICE.get(k, 0.0) for k in classes):.0f} cycles (each block once)")
    detail = ["v_cmp", "v_cndmask", "v_mov", "v_min", "v_max", "v_med3", "v_cvt", "v_mul_lo", "v_and_or", "v_bfi", "s_and_saveexec", "s_or_saveexec",
              "s_cbranch_execz", "s_cbranch_execnz", "s_nop"]
    out = []
    for d in detail:
        n = sum(v for k, v in tot.items() if k.startswith("op:" + d))
        if n:
            out.append(f"{d}* {n}")
    print("  ".join(out))


if __name__ == "__main__":
    main()
