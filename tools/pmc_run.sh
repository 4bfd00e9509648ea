#!/bin/bash
# usage: tools/pmc_run.sh "<COUNTERS...>" <python script + args>   (GPU box; PMC-only run: no tracing flags)
# prints the per-kernel mean of every counter for kernels of this library
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
CNT="$1"; shift
OUT=$(mktemp -d /tmp/pmc.XXXXXX)
export TMPDIR=/tmp
cd /tmp; export PYTHONPATH="$R:$PYTHONPATH"
rocprofv3 --pmc $CNT --output-format csv -d "$OUT" -o pmc -- "$@" > "$OUT/log.txt" 2>&1 || { tail -20 "$OUT/log.txt"; exit 1; }
python "$R/tools/summarize_prof.py" "$OUT/summary.json" "$OUT" > /dev/null
python - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "avl" in k:
        print(k[:48].ljust(48), "  ".join(f"{c}={x['mean']:.4g}" for c, x in v.items()))
PY
