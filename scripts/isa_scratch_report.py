#!/usr/bin/env python3
"""Where do a pair kernel's scratch (private-memory) accesses sit?  Reads the gfx950 assembly of pair_kslice.hip
(hipcc -S --cuda-device-only) and, per kernel, lists the basic blocks that hold the tile walk's VALU work
(v_bitop3 count) with the scratch loads / stores inside them: spills INSIDE a hot block cost every chunk,
spills outside cost once per k-mer length or per workgroup.

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I../../include -S --cuda-device-only pair_kslice.hip -o /tmp/kslice.s
    python scripts/isa_scratch_report.py /tmp/kslice.s
"""
import re
import sys


def main(path):
    text = open(path).read()
    funcs = re.split(r"\n(?=_ZN3skl\w*pair_kernel\w*:)", text)
    for f in funcs:
        m = re.match(r"(_ZN3skl\w*pair_kernel\w*):", f)
        if not m:
            continue
        body = f.split(".Lfunc_end")[0]
        blocks, cur, label = [], [], "entry"
        for line in body.split("\n"):
            if re.match(r"^\.LBB\d+_\d+:", line):
                blocks.append((label, cur))
                cur, label = [], line.split(":")[0]
            else:
                cur.append(line)
        blocks.append((label, cur))
        total = sum("scratch_" in l for l in body.split("\n"))
        hot = [(lab, sum("v_bitop3" in l for l in b), sum("scratch_load" in l for l in b), sum("scratch_store" in l for l in b))
               for lab, b in blocks if sum("v_bitop3" in l for l in b) >= 100]
        in_hot = sum(h[2] + h[3] for h in hot)
        print(f"{m.group(1)}: {total} scratch instructions, {in_hot} of them inside the {len(hot)} blocks that hold the walk")
        for lab, nb, nl, ns in hot:
            if nl + ns:
                print(f"    {lab}: {nb} v_bitop3, {nl} scratch loads, {ns} scratch stores")


if __name__ == "__main__":
    main(sys.argv[1])
