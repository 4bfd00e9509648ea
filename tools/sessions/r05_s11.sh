#!/bin/bash
# round 5: full GPU suite on the K3 change, default bench line, builder at D = 1536 (six-chunk variant: register count went up)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s11; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python tools/probe_build_width.py 1536 1 2000 > $O/width1536.txt 2>&1
timeout 300 python tools/probe_build_width.py 1536 0 2000 >> $O/width1536.txt 2>&1
timeout 300 python tools/probe_build_width.py 1536 16 2000 >> $O/width1536.txt 2>&1
timeout 300 python tools/probe_build_width.py 1024 1 2000 >> $O/width1536.txt 2>&1
timeout 300 python tools/probe_build_width.py 768 1 2000 >> $O/width1536.txt 2>&1
cat $O/width1536.txt | grep -v amdgpu.ids
timeout 600 python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log | cut -c1-400
