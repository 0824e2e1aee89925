#!/usr/bin/env python3
"""Copies the round's judged evidence from gpurun_out/ (scratch) into profiles/ (tracked):
kernel-stats summaries, the raw FETCH_SIZE / WRITE_SIZE counter collections + hbm_traffic.json,
a per-kernel summary of the SQ LDS / VALU counters, and the bench lines.
usage: collect_profiles.py <round tag, e.g. r02>   (expects gpurun_out/<tag>_final_{if1,if3,pmc_fetch,pmc_write,pmc_sq}/)"""
import collections
import csv
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^()]*>)?)", n)
    return m.group(1) if m else n[:60]


def main():
    tag = sys.argv[1]
    g, p = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    cp = lambda a, b: shutil.copyfile(os.path.join(g, a), os.path.join(p, b))
    cp(f"{tag}_final_if1/{tag}_kernel_stats.csv", f"{tag}_final_kernel_stats_inflight1.csv")
    cp(f"{tag}_final_if3/{tag}_kernel_stats.csv", f"{tag}_final_kernel_stats_inflight3.csv")
    cp(f"{tag}_final_pmc_fetch/{tag}_counter_collection.csv", f"{tag}_final_pmc_fetch_size.csv")
    cp(f"{tag}_final_pmc_write/{tag}_counter_collection.csv", f"{tag}_final_pmc_write_size.csv")
    cp(f"{tag}_final_bench.json", f"{tag}_final_bench.json")
    for extra in ("other_shapes.jsonl", "bench_sp1_p3_lookups.json", "bench_p2_chip.jsonl", "bench_compress_hashes.json"):
        if os.path.exists(os.path.join(g, f"{tag}_{extra}")):
            cp(f"{tag}_{extra}", f"{tag}_{extra}")
    for k in ("if1", "if3"):
        lines = [l for l in open(os.path.join(g, f"{tag}_final_{k}.log")) if l.startswith('{"metric"')]
        open(os.path.join(p, f"{tag}_final_bench_inflight{k[-1]}_under_rocprof.json"), "w").write(lines[-1])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"),
                           os.path.join(p, f"{tag}_final_pmc_fetch_size.csv"), os.path.join(p, f"{tag}_final_pmc_write_size.csv"),
                           "hash_rows_kernel", os.path.join(p, "hbm_traffic.json")], stdout=subprocess.DEVNULL)
    d = json.load(open(os.path.join(p, "hbm_traffic.json")))
    bench = json.load(open(os.path.join(p, f"{tag}_final_bench.json")))
    algo = bench["roofline"]["algorithmic_bytes_per_launch"]
    d.update(source=f"profiles/{tag}_final_pmc_fetch_size.csv + {tag}_final_pmc_write_size.csv",
             method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --no-cpu --no-h2d "
                    "--no-verify --inflight 1 --steps 1 --warmup 0`; FETCH_SIZE doubled (gfx950 streaming-read correction), KiB -> bytes",
             algorithmic_bytes_per_launch=algo, ratio=round(d["bytes_per_launch"] / algo, 4))
    json.dump(d, open(os.path.join(p, "hbm_traffic.json"), "w"), indent=1)
    rows = list(csv.DictReader(open(os.path.join(g, f"{tag}_final_pmc_sq/{tag}_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in rows:
        k = short(r["Kernel_Name"])
        if not any(t in k for t in ("nf_", "hash_", "ntt_pass")):
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    out = os.path.join(p, f"{tag}_final_pmc_ntt_hash_lds_valu.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Launches", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE",
                    "SQ_LDS_BANK_CONFLICT", "conflict_over_lds_active", "valu_busy = ACTIVE_INST_VALU*4/(BUSY_CYCLES*32)"])
        tc = ta = 0
        for k, v in sorted(agg.items()):
            gg = lambda c: v[c][1] / max(v[c][0], 1)
            n = v["SQ_BUSY_CYCLES"][0]
            lds, conf = gg("SQ_LDS_IDX_ACTIVE"), gg("SQ_LDS_BANK_CONFLICT")
            if k.startswith("nf_"):
                tc += conf * n
                ta += lds * n
            w.writerow([k, n, round(gg("SQ_INSTS_VALU")), round(gg("SQ_ACTIVE_INST_VALU")), round(gg("SQ_BUSY_CYCLES")), round(lds),
                        round(conf), round(conf / lds, 4) if lds else 0, round(gg("SQ_ACTIVE_INST_VALU") * 4 / (gg("SQ_BUSY_CYCLES") * 32), 3)])
        w.writerow(["ALL nf_* kernels (weighted by launches)", "", "", "", "", round(ta), round(tc), round(tc / ta, 4), ""])
    print(open(out).read())
    print(json.dumps(d))
    print("bench:", bench["value"], bench["ms_per_step"], bench.get("value_with_h2d"), bench["cpu_baseline"]["value"], bench["cpu_baseline"]["value_1_thread"])
    st = list(csv.DictReader(open(os.path.join(p, f"{tag}_final_kernel_stats_inflight1.csv"))))
    # proofs in the trace: hash_rows_kernel runs 7 times per segment proof (3 groups + check + 3 FRI rounds at 2^20 cycles)
    n_proofs = max(1, round(sum(int(r["Calls"]) for r in st if short(r["Name"]).startswith("hash_rows_kernel")) / 7))
    print("proofs in the serial trace:", n_proofs)
    for r in st[:9]:
        print(short(r["Name"]).ljust(44), r["Calls"].rjust(5), "avg %.1f us" % (float(r["AverageNs"]) / 1e3),
              "%.3f ms/segment" % (float(r["TotalDurationNs"]) / 1e6 / n_proofs))


if __name__ == "__main__":
    main()
