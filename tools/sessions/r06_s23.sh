#!/bin/bash
# round 6, final tree: the whole profile collection (tools/collect_profiles.sh r06), the merge's kernel table (eight ranks as threads of one process), the fuzzers
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
O=gpurun_out/prof_r06
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_m2 -o m2 -- python $GRAFT_REPO_ROOT/tools/probe_merge2.py 8 10000 4 > $GRAFT_REPO_ROOT/$O/merge2_one_process.log 2>&1; cp /tmp/p_m2/m2_kernel_stats.csv $GRAFT_REPO_ROOT/$O/merge2_8ranks_one_process_kernel_stats.csv )
timeout 700 python tools/fuzz_parity.py 420 31 > $O/fuzz_parity_final.log 2>&1; tail -2 $O/fuzz_parity_final.log
timeout 500 python tools/fuzz_merge2.py 300 41 > $O/fuzz_merge2_final.log 2>&1; tail -2 $O/fuzz_merge2_final.log
ls $O | wc -l
