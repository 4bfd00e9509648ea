#!/bin/bash
# round 5: final check of the tree -- smoke(), the whole GPU suite, the default line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s30; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log | cut -c1-330; tail -4 $O/bench_default.log | grep real
