"""Regenerates tests/golden/seal_digests.json from the CPU oracle (regression pins of the
restated algorithm; NOT vectors of the risc0 binary, which cannot run here).
`--large` also (re)computes the BASELINE-size entries of seal_digests_large.json: config 2
(S20: 2^20 cycles, 16/16/224 columns, seed 20240807) and the script's po2 = 18
(script/prove-block.sh:71) -- minutes of CPU time on 8 cores."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as o
from raiko_amd.segment import synthetic_segment
from test_oracle_prover import CASES, LARGE_CASES, ROUND2_CASES, digest, round2_case

def run(cases, name):
    out = {}
    for key, (po2, widths, seed) in sorted(cases.items()):
        seal = o.oracle_prove(synthetic_segment(po2, widths, seed=seed))
        out[key] = {"words": int(seal.size), "sha256": digest(seal)}
        print(key, out[key], flush=True)
    json.dump(out, open(os.path.join(HERE, name), "w"), indent=1, sort_keys=True)

run(CASES, "seal_digests.json")
# round 2: other parameter sets / protocol shapes / circuits given as constraint lists
out = {}
for key in ROUND2_CASES:
    kw, seg = round2_case(key)
    kw = dict(kw)
    o.oracle_set_params(kw.pop("preset"), **kw)
    seal = o.oracle_prove(seg)
    o.oracle_set_params()
    out[key] = {"words": int(seal.size), "sha256": digest(seal)}
    print(key, out[key], flush=True)
json.dump(out, open(os.path.join(HERE, "seal_digests_round2.json"), "w"), indent=1, sort_keys=True)
if "--large" in sys.argv:
    run(LARGE_CASES, "seal_digests_large.json")
