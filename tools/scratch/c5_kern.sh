#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for cfg in "--queries 128" "--queries 128 --dense" "--queries 64 --dense" "--queries 64"; do
 for m in 0 1; do
 rm -rf /tmp/prof
 AVL_SIM_NO_ONEPASS=$m timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $R/bench.py --feat-dim 1536 $cfg --no-cpu --steps 30 --warmup 10 > /tmp/o.txt 2>&1
 python - "$cfg" "$m" <<PY
import csv,glob,sys
f=glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)
out=[]
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if 'sim_' in n and float(r['AverageNs'])>50000:
        import re
        m=re.search(r'(sim_[a-z0-9_]+kernel)(ILi\d+ELi\d+ELb\d)?', n)
        out.append(f"{m.group(0) if m else n[:40]}:{float(r['AverageNs'])/1e3:.1f}us x{r['Calls']}")
print(sys.argv[1], 'no_onepass='+sys.argv[2], ' '.join(out))
PY
 done
done
