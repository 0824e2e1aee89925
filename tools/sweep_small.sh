#!/bin/bash
# in-flight / verify sweep at small segment sizes (prove-block.sh uses execution_po2 = 18)
for p in 16 18; do for k in 3 6; do for v in "" "--no-verify"; do
  timeout -k 10 120 python bench.py --po2 $p --inflight $k --steps 96 --no-cpu --no-h2d $v 2>/dev/null > /tmp/sweep.json || exit 1
  python - "$p" "$k" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep.json"))
print(json.dumps({"po2": int(sys.argv[1]), "inflight": int(sys.argv[2]), "verify": sys.argv[3] == "", "cycles_per_s": d["value"],
                  "ms_per_segment": d["ms_per_step"], "serial_ms": d["pipeline"]["serial_stage_ms"]["total"]}))
PY
done; done; done
