"""Regenerates tests/golden/pcs_digests.json from the CPU oracle (oracle/or_pcs.c): regression pins of the restated
Plonky3 PCS steps on seeded inputs -- NOT vectors of Plonky3 itself (its crates are outside the reference tree and
there is no Rust toolchain here).  The oracle's functions are pinned separately by big-integer algebra (tests/test_pcs.py)."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from pcs_cases import PCS_CASES, oracle_outputs

out = {key: oracle_outputs(key) for key in sorted(PCS_CASES)}
for k, v in out.items():
    print(k, v)
json.dump(out, open(os.path.join(HERE, "pcs_digests.json"), "w"), indent=1, sort_keys=True)
