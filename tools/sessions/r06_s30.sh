#!/bin/bash
# round 6, the committed tree as the driver will run it: GPU suite, smoke, default bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s30; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/t_gpu.log 2>&1; tail -2 $O/t_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; tail -c 2500 $O/bench_default.log
