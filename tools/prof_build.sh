#!/bin/bash
# GPU box: per-kernel averages of the builder for a given frames-per-launch.  usage: tools/prof_build.sh B [steps]
B=${1:-1}; STEPS=${2:-1000}
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pb_$B
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$B -o b -- python $R/bench.py --workload build --steps $STEPS --warmup 32 --build-batch $B --no-cpu > /dev/null 2>&1
python - <<PY
import csv
rows = list(csv.reader(open("/tmp/pb_$B/b_kernel_stats.csv")))
for r in rows[1:]:
    if "avl::" in r[0]:
        print(f"B=$B  {r[0].split('(')[0][:48]:48s} calls {r[1]:>6s}  avg {float(r[3])/1e3:9.2f} us  per-frame {float(r[3])/1e3/$B:7.2f} us")
PY
