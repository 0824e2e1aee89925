"""Regenerates tests/golden/seal_digests.json from the CPU oracle (regression pins of the
restated algorithm; NOT vectors of the risc0 binary, which cannot run here)."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as o
from raiko_amd.segment import synthetic_segment
from test_oracle_prover import CASES, digest
out = {}
for name, (po2, widths, seed) in sorted(CASES.items()):
    seal = o.oracle_prove(synthetic_segment(po2, widths, seed=seed))
    out[name] = {"words": int(seal.size), "sha256": digest(seal)}
json.dump(out, open(os.path.join(HERE, "seal_digests.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
