#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s36
timeout 900 python -m pytest tests/test_builder_gpu.py -q -m gpu -k "heat" -x 2>&1 | tail -5 > gpurun_out/s36/tests.txt
timeout 300 python -m pytest tests/test_api_gpu.py -q -m gpu -k "index" -x 2>&1 | tail -3 >> gpurun_out/s36/tests.txt
for k in uniform clustered; do timeout 200 python tools/probe_heat.py $k 30 >> gpurun_out/s36/probe.txt 2>&1; done
HEAT_GEOMETRY=cube timeout 200 python tools/probe_heat.py uniform 30 >> gpurun_out/s36/probe.txt 2>&1
timeout 300 python tools/fuzz_parity.py 120 77 > gpurun_out/s36/fuzz.txt 2>&1
