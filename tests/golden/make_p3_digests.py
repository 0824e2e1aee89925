"""Regenerates tests/golden/p3_digests.json: sha256 of the oracle's proof (oracle/or_p3.c) for every case of
tests/p3_cases.py.  Self-generated regression pins -- nothing reference-held exists for this path."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as o  # noqa: E402
from p3_cases import P3_CASES, init_of, sha, tables_of  # noqa: E402

out = {}
for case, (preset, over, _, _) in P3_CASES.items():
    o.oracle_set_params(preset, **over)
    pf = o.oracle_p3_prove(tables_of(case), init_of(case))
    out[case] = {"words": int(pf.size), "sha256": sha(pf)}
o.oracle_set_params()
with open(os.path.join(HERE, "p3_digests.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
