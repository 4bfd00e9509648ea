#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02c; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/pytest_gpu.log | head -40
grep -n "^E " $O/pytest_gpu.log | head -40
