#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02e; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/pytest_gpu.log | head -20; grep -n "^E  " $O/pytest_gpu.log | head -30
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
