#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_s31; mkdir -p $O
PROBE_CPROFILE_RANK=3 timeout 600 python tools/probe_merge2.py 8 10000 6 > $O/cprofile.log 2>&1
grep -A45 "function calls" $O/cprofile.log | cut -c1-170
