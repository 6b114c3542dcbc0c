#!/usr/bin/env python3
"""VALU instructions of one kernel by SOURCE LINE (an ISA listing compiled with -gline-tables-only).

  hipcc <HIPFLAGS> -gline-tables-only -S --cuda-device-only -o /tmp/k.s csrc/pathtrace_fast.hip
  python tools/isa_lines.py /tmp/k.s pathtrace_pool_kernelILb1 [--bucket 10]

Prints, per source file, the lines that own VALU instructions (count, of which transcendental) — the static picture of where a
kernel's issue slots go, to be read next to the source.  Line attribution is the compiler's (`.loc`): an instruction hoisted or
sunk keeps the line it came from."""
import collections
import re
import sys

from isa_blocks import classify, kernel_lines  # noqa: E402  (same directory)


def main():
    path, sym = sys.argv[1], sys.argv[2]
    files = {}
    for l in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = m.group(2)
    cur = (0, 0)
    cnt = collections.Counter()
    trans = collections.Counter()
    for l in kernel_lines(path, sym):
        s = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c = classify(op)
        if c in ("fp2", "fma3", "int2", "sel4", "trans"):
            cnt[cur] += 1
            if c == "trans":
                trans[cur] += 1
    total = sum(cnt.values())
    print(f"VALU {total}  trans {sum(trans.values())}")
    for f in sorted({k[0] for k in cnt}):
        sub = sum(v for k, v in cnt.items() if k[0] == f)
        print(f"== {files.get(f, f)}: {sub}")
        for k in sorted(k for k in cnt if k[0] == f):
            print(f"  {k[1]:5d}  {cnt[k]:4d}" + (f"  trans {trans[k]}" if trans[k] else ""))


if __name__ == "__main__":
    main()
