#!/usr/bin/env python3
"""Instruction census of the hot loop of a kernel in a gfx950 assembly listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --cuda-device-only -S -o eng.s <package>/csrc/hw_engine.hip
    python tools/loop_census.py eng.s zbc_superposed_kernelIfLi1ELi1E
Finds the kernel whose mangled name contains the pattern, takes its innermost loops with at least 10 wide multiplies and counts
opcodes (the numbers quoted in DESIGN.md section 5 and in bench.py's VALU_MODEL come from here)."""
import re
import sys
from collections import Counter


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2]
    m = re.search(r"^(_ZN3hwk[^\n:]*%s[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel" % re.escape(pat), s, re.S | re.M)
    if not m:
        sys.exit("kernel not found")
    body = m.group(2)
    lines = body.split("\n")
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", l.strip())
        if mm:
            labels[mm.group(1)] = i
    spans = []
    for i, l in enumerate(lines):
        mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] <= i:
            spans.append((labels[mm.group(1)], i))
    inner = [a for a in spans if not any(b != a and a[0] <= b[0] and b[1] <= a[1] for b in spans)]
    hot = [sp for sp in inner if sum("v_mad_u64_u32" in x for x in lines[sp[0]:sp[1] + 1]) >= 10]
    print(m.group(1))
    for best in hot:                       # a kernel may hold several copies of the loop (e.g. the uniform-high-word one)
        loop = [l.strip() for l in lines[best[0]:best[1] + 1] if l.strip() and not l.strip().startswith((";", "."))]
        c = Counter(l.split()[0] for l in loop)
        print(len(loop), "instructions in hot loop", lines[best[0]].strip().split(":")[0])
        for k, v in sorted(c.items(), key=lambda kv: (-kv[1], kv[0])):
            if not k.startswith("s_") or len(sys.argv) > 3:
                print(f"  {k:24s} {v}")
    tail = s[m.end():m.end() + 6000]
    for k in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy"):
        mm = re.search(r"; %s: (\d+)" % k, tail)
        if mm:
            print(k, mm.group(1))
    if len(sys.argv) > 3:
        print("\n".join(loop))


if __name__ == "__main__":
    main()
