#!/bin/bash
# round 5: closing sweeps on the tree with the specialised K3: randomised parity sweep (builder modes vs the sequential oracle, similarity shapes), deferred-fuse soak
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s21; mkdir -p $O
timeout 1000 python tools/fuzz_parity.py 800 550001 > $O/fuzz.txt 2>&1; tail -6 $O/fuzz.txt
timeout 600 bash tools/soak_deferred.sh > $O/soak_deferred.txt 2>&1; tail -4 $O/soak_deferred.txt
