#!/bin/bash
# round 5, first call of the re-entered session: GPU tests on the current tree, the default line, config 5 raw / compact, compact config 2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.log 2>&1; tail -c 600 $O/bench_default.log | head -c 300
C5="--feat-dim 1536 --queries 128"
timeout 400 python bench.py $C5 --steps 200 --no-pmc > $O/config5_line.log 2>&1
timeout 400 python bench.py $C5 --steps 200 --resident compact --no-pmc > $O/config5_compact_line.log 2>&1
timeout 400 python bench.py --resident compact --no-build-extra --no-pmc > $O/index_compact_line.log 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -o c5 -- python $GRAFT_REPO_ROOT/bench.py $C5 --steps 100 --profile-run > $GRAFT_REPO_ROOT/$O/c5_trace.log 2>&1
cp /tmp/p_c5/c5_kernel_stats.csv $GRAFT_REPO_ROOT/$O/ 2>/dev/null
