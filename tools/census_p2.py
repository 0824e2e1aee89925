#!/usr/bin/env python3
"""Dynamic VALU instruction count of one Poseidon2 permutation from the gfx950 ISA.

Compiles tools/ubench_p2.hip (which inlines p2::permute once inside a loop) to assembly, splits
the kernel at its two rolled loops (four full rounds each, `#pragma unroll 1` in
poseidon2_core.hpp) and weights their bodies by the trip count.  The result is the constant
raiko_amd/segment.py:P2_VALU_PER_PERMUTATION used by bench.py's `roofline.alu`.

  python tools/census_p2.py          (needs hipcc; no GPU)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "p2.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "raiko_amd", "csrc"),
                               os.path.join(ROOT, "tools", "ubench_p2.hip"), "-S", "--cuda-device-only", "-o", asm],
                              stderr=subprocess.DEVNULL)
        text = open(asm).read()
    start = text.index("_Z6k_perm")
    lines = text[start:].split("\n")
    end = next(i for i, l in enumerate(lines) if "s_endpgm" in l)
    lines = lines[:end]
    # inner loops: a label that is the target of a backward branch, nested in the harness loop
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    loops.sort()
    outer = max(loops, key=lambda ab: ab[1] - ab[0])
    inner = [ab for ab in loops if ab != outer and outer[0] <= ab[0] and ab[1] <= outer[1]]
    assert len(inner) == 2, loops

    def count(a, b):
        n = fast = 0
        for l in lines[a:b + 1]:
            t = l.strip().split(" ")[0] if l.strip() else ""
            if t.startswith("v_"):
                n += 1
                fast += bool(re.match(r"v_(add|sub|subrev)_u32", t))
        return n, fast

    total = list(count(outer[0], outer[1]))
    for a, b in inner:  # bodies run 4 times (rounds 0..3 and 4..7), counted once above
        n, f = count(a, b)
        total[0] += 3 * n
        total[1] += 3 * f
    print("VALU instructions per permutation: %d (plain add/sub: %d); loop bodies: %s" %
          (total[0], total[1], [count(a, b)[0] for a, b in inner]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
