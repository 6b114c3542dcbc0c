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


def operands(text):
    body = text.split(None, 1)[1] if " " in text else ""
    return [o.strip() for o in body.split(",")]


def select_census(blocks):
    """Compare / select / move census per block, loop blocks only, and the compare+select pairs that are min / max in disguise:
    v_cmp_{lt,gt,le,ge}_f32 m, A, B   followed in the same block, before m is rewritten, by   v_cndmask d, X, Y, m   with {X, Y} = {A, B}
    (operand modifiers such as |x| disqualify a pair: v_min / v_max take them, but then the select does not return A or B itself)."""
    print(f"{'block':12s} {'VALU':>5s} {'v_cmp':>6s} {'v_cndmask':>9s} {'v_mov':>6s} {'min/max/med3':>12s} {'cmp+sel = min/max':>18s}")
    tot = collections.Counter()
    for name, _, c, ins in blocks:
        if not c["in_loop"]:
            continue
        valu = sum(c[k] for k in ("fp2", "fma3", "int2", "sel4", "trans"))
        ncmp = sum(v for k, v in c.items() if k.startswith("op:v_cmp"))
        nsel = sum(v for k, v in c.items() if k.startswith("op:v_cndmask"))
        nmov = sum(v for k, v in c.items() if k.startswith("op:v_mov"))
        nmm = sum(v for k, v in c.items() if k.startswith(("op:v_min", "op:v_max", "op:v_med3")))
        pairs = 0
        live = {}   # mask register -> (A, B) of the float compare that wrote it
        for t in ins:
            op = t.split()[0]
            ops = operands(t)
            if re.match(r"v_cmp_(lt|gt|le|ge|nlt|ngt|nle|nge)_f32", op):
                if op.endswith("_e64") or len(ops) == 3:
                    live[ops[0]] = (ops[1], ops[2])
                else:
                    live["vcc"] = (ops[-2], ops[-1])
            elif op.startswith("v_cndmask"):
                mask = ops[3] if len(ops) > 3 else "vcc"
                if mask in live and set(ops[1:3]) == set(live[mask]) and not any("|" in o or o.startswith("-") for o in ops[1:3] + list(live[mask])):
                    pairs += 1
            elif op.startswith(("v_cmp", "s_and_saveexec", "s_or_b64", "s_and_b64", "s_xor_b64", "s_andn2_b64")):
                live.pop(ops[0] if ops else "", None)
                if not op.endswith("_e64") and op.startswith("v_cmp"):
                    live.pop("vcc", None)
        tot.update(valu=valu, cmp=ncmp, sel=nsel, mov=nmov, mm=nmm, pairs=pairs)
        if ncmp or nsel or nmov:
            print(f"{name:12s} {valu:5d} {ncmp:6d} {nsel:9d} {nmov:6d} {nmm:12d} {pairs:18d}")
    print(f"{'loop total':12s} {tot['valu']:5d} {tot['cmp']:6d} {tot['sel']:9d} {tot['mov']:6d} {tot['mm']:12d} {tot['pairs']:18d}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("symbol")
    ap.add_argument("--loop", action="store_true", help="only the blocks of the innermost Depth=2 loop")
    ap.add_argument("--dump", default=None, help="write the kernel's listing here")
    ap.add_argument("--blocks", action="store_true", help="print every block (default: totals only)")
    ap.add_argument("--select-census", action="store_true",
                    help="round 6: per block the compare / select / move / min-max instructions, and how many compare+select pairs are a "
                         "min or max of the compared operands themselves (what a v_min / v_max / v_min3 rewrite could absorb)")
    a = ap.parse_args()
    lines = kernel_lines(a.file, a.symbol)
    if a.dump:
        open(a.dump, "w").write("\n".join(lines) + "\n")
    blocks = []   # (name, in_loop, Counter, instruction texts)
    cur = ["entry", False, collections.Counter(), []]
    blocks.append(cur)
    depth2_header = None
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        m2 = re.match(r"^; %bb\.(\d+):", l)
        if m or m2:
            name = m.group(1) if m else "bb." + m2.group(1)
            cur = [name, "Depth=2" in l, collections.Counter(), []]
            if "Depth=" in l:
                cur[2]["in_loop"] = 1
            blocks.append(cur)
            continue
        if "Depth=2" in l and l.strip().startswith(";"):   # the loop-header comment sits on its own line after the label
            cur[1] = True
            cur[2]["in_loop"] = 1
            continue
        if "Depth=" in l and l.strip().startswith(";"):
            cur[2]["in_loop"] = 1
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        cur[2][classify(op)] += 1
        cur[2]["op:" + re.sub(r"_e(32|64)$", "", op)] += 1
        cur[3].append(t.split(";")[0].strip())
    if a.select_census:
        return select_census(blocks)
    sel = [b for b in blocks if (b[1] or not a.loop)]
    classes = ["fp2", "fma3", "int2", "sel4", "trans", "s_nop", "salu", "branch", "lds", "vmem", "other"]
    if a.blocks:
        print(f"{'block':14s} " + " ".join(f"{c:>6s}" for c in classes) + "   est.cycles")
        for name, _, c, _ in sel:
            if sum(c[k] for k in classes):
                est = sum(c[k] * PRICE.get(k, 0.0) for k in classes)
                print(f"{name:14s} " + " ".join(f"{c[k]:6d}" for k in classes) + f"   {est:9.1f}")
    tot = collections.Counter()
    for _, _, c, _ in sel:
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
