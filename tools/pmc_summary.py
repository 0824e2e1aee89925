#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter collection (SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU
SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU): usage pmc_summary.py <counter_collection.csv> <out.csv> [name filter ...]"""
import collections, csv, re, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^()]*>)?)", n)
    return m.group(1) if m else n[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    keep = sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in rows:
        k = short(r["Kernel_Name"])
        if keep and not any(t in k for t in keep):
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Launches", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
                    "conflict_over_lds_active", "valu_busy = ACTIVE_INST_VALU*4/(BUSY_CYCLES*32)"])
        for k, v in sorted(agg.items()):
            gg = lambda c: v[c][1] / max(v[c][0], 1)
            lds, conf, busy = gg("SQ_LDS_IDX_ACTIVE"), gg("SQ_LDS_BANK_CONFLICT"), gg("SQ_BUSY_CYCLES")
            w.writerow([k, v["SQ_BUSY_CYCLES"][0], round(gg("SQ_INSTS_VALU")), round(gg("SQ_ACTIVE_INST_VALU")), round(busy), round(lds), round(conf),
                        round(conf / lds, 4) if lds else 0, round(gg("SQ_ACTIVE_INST_VALU") * 4 / (busy * 32), 3) if busy else 0])
    print(open(sys.argv[2]).read())


if __name__ == "__main__":
    main()
