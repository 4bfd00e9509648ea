#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -o c5 -- python $R/bench.py --feat-dim 1536 --queries 128 --steps 100 --profile-run > $O/config5_prof.log 2>&1
cp /tmp/p_c5/c5_kernel_stats.csv $O/config5_kernel_stats.csv 2>/dev/null
grep -E "avl" $O/config5_kernel_stats.csv | cut -c1-200 | head
cd $R
timeout 600 python tools/ab_sim.py --reps 2 --shapes 2000000x1024x64,2000000x1536x64,2000000x512x64 stock > $O/ab_d1024.log 2>&1; tail -5 $O/ab_d1024.log
timeout 900 python -m pytest tests/test_sim_gpu.py tests/test_builder_gpu.py -q -m gpu > $O/pytest_sim.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sim.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/pytest_sim.log | head -20; grep -n "^E  " $O/pytest_sim.log | head -20
