#!/bin/bash
# round 6, final tree: soak -- the GPU suite three times over, 20 min of parity fuzzing, 15 min of merge fuzzing (half of it with a chunked exchange)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s26; mkdir -p $O
for k in 1 2 3; do
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/t_gpu_$k.log 2>&1; tail -1 $O/t_gpu_$k.log
done
timeout 1400 python tools/fuzz_parity.py 1200 51 > $O/fuzz_parity.log 2>&1; tail -1 $O/fuzz_parity.log
timeout 1100 python tools/fuzz_merge2.py 900 61 > $O/fuzz_merge2.log 2>&1; tail -1 $O/fuzz_merge2.log
