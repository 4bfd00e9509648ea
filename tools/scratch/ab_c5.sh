#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for Q in 64 128 96; do
for rep in 1 2; do
for m in 0 1; do
 AVL_SIM_NO_ONEPASS=$m timeout -s KILL 200 python bench.py --feat-dim 1536 --queries $Q --no-cpu --steps 30 --warmup 10 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('Q=$Q no_onepass=$m', round(r['ms_per_step'],4), 'ms', round(r['roofline']['frac'],4))"
done; done; done
