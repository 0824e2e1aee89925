#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, TCC slots do not fit
both) into profiles/hbm_traffic.json for bench.py's roofline.traffic.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of
a coalesced streaming read, so it is doubled; WRITE_SIZE is exact.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substring> <out.json>
"""
import csv
import json
import sys


def total(path, counter, needle):
    n, s = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
            n += 1
            s += float(r["Counter_Value"])
    return n, s


def main():
    fetch_csv, write_csv, needle, out = sys.argv[1:5]
    nf, f = total(fetch_csv, "FETCH_SIZE", needle)
    nw, w = total(write_csv, "WRITE_SIZE", needle)
    assert nf == nw and nf > 0, (nf, nw)
    fetch_bytes = 2.0 * f * 1024.0
    write_bytes = w * 1024.0
    doc = {
        "kernel": needle,
        "launches": nf,
        "fetch_bytes_total": fetch_bytes,
        "write_bytes_total": write_bytes,
        "bytes_per_launch": round((fetch_bytes + write_bytes) / nf),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                  "`bench.py --steps 1 --warmup 0 --inflight 1`; FETCH_SIZE doubled (gfx950 streaming-read correction), KiB -> bytes",
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
