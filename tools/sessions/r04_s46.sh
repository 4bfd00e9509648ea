#!/bin/bash
# does the compact (3 B / element) walk re-fetch the half-shared cache lines from L2?  TCC request counters, raw vs compact
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s46
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i "TCC_REQ\|TCC_HIT\|TCC_MISS\|TCC_READ\|TCP_TCC_READ\|TCP_TOTAL_CACHE\|TCC_EA_RDREQ\|TCP_TCC" | head -40 > $O/counters.txt
for form in raw compact; do
  extra=""; [ $form = compact ] && extra="--resident compact"
  for cnt in "TCC_REQ_sum" "TCC_HIT_sum" "TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum"; do
    timeout 300 rocprofv3 --pmc $cnt --output-format csv -d /tmp/t_${form}_$cnt -o x -- python $R/bench.py --steps 5 --warmup 2 --settle-steps 0 --profile-run $extra > /dev/null 2>&1
    f=$(find /tmp/t_${form}_$cnt -name "*counter_collection.csv" | head -1)
    python - "$f" "$form" "$cnt" >> $O/tcc.txt <<'PY'
import csv, sys
f, form, cnt = sys.argv[1:4]
vals = {}
try:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name") or row.get("Kernel Name") or ""
        if "sim_split" in k and "prepare" not in k:
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for c, v in vals.items():
        print(form, c, sum(v) / len(v), len(v))
except Exception as e:
    print(form, cnt, "failed", e)
PY
  done
done
