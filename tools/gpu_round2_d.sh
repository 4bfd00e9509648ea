#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/r02d; mkdir -p $O
timeout 900 python tools/ab_sim.py --reps 3 --shapes 2000000x1024x64,2000000x1536x64,2000000x768x32,2000000x1536x128,2000000x512x128,2000000x512x64 stock tb > $O/ab_tb.log 2>&1; tail -14 $O/ab_tb.log
for v in stock tb stock tb; do
  if [ "$v" = stock ]; then unset AVLMAPS_HIP_LIB; else export AVLMAPS_HIP_LIB=$R/variants/libavlmaps_hip_$v.so; fi
  timeout 300 python bench.py --feat-dim 1536 --queries 128 --steps 100 --no-build-extra --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v config5 blocks', d['ms_per_step'], d['roofline']['frac'])"
done
