#!/bin/bash
# round 5, final tree: randomised parity sweep incl. the C frame loop (a frame's launch prepares the next frame's K1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_s47
timeout 1000 python tools/fuzz_parity.py 420 550078 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r05_s47/fuzz.txt
