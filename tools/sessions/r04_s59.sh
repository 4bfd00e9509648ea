#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s59
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/s59/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/s59/smoke.txt 2>&1
timeout 900 python bench.py > gpurun_out/s59/bench.json 2> gpurun_out/s59/bench.err
