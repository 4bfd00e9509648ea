#!/bin/bash
# GPU box: collect the rocprofv3 evidence for bench.py into gpurun_out/prof_<tag>/ (copy the summaries to profiles/ afterwards).
# usage: tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r01}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
run() { timeout 300 "$@"; }
run rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_index -o index -- python $R/bench.py --profile-run > $O/index_bench.log 2>&1
cp /tmp/p_index/index_kernel_stats.csv $O/ 2>/dev/null
run rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_build -o build -- python $R/bench.py --workload build --steps 500 --warmup 20 --no-cpu > $O/build_bench.log 2>&1
cp /tmp/p_build/build_kernel_stats.csv $O/ 2>/dev/null
# PMC passes: counters only, no tracing flags
run rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o fetch -- python $R/bench.py --steps 5 --warmup 2 --settle-steps 0 --profile-run > $O/pmc_fetch.log 2>&1
run rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o write -- python $R/bench.py --steps 5 --warmup 2 --settle-steps 0 --profile-run > $O/pmc_write.log 2>&1
run rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetchb -o fetchb -- python $R/bench.py --workload build --steps 100 --warmup 5 --no-cpu > $O/pmc_fetch_build.log 2>&1
run rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_writeb -o writeb -- python $R/bench.py --workload build --steps 100 --warmup 5 --no-cpu > $O/pmc_write_build.log 2>&1
run rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_sq -o sq -- python $R/bench.py --steps 5 --warmup 2 --settle-steps 0 --profile-run > $O/pmc_sq.log 2>&1
python $R/tools/summarize_prof.py $O/pmc_index.json /tmp/p_fetch /tmp/p_write /tmp/p_sq > /dev/null
python $R/tools/summarize_prof.py $O/pmc_build.json /tmp/p_fetchb /tmp/p_writeb > /dev/null
run rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -o c5 -- python $R/bench.py --feat-dim 1536 --queries 128 --steps 100 --profile-run > $O/config5_bench.log 2>&1
cp /tmp/p_c5/c5_kernel_stats.csv $O/config5_kernel_stats.csv 2>/dev/null
run rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b64 -o b64 -- python $R/bench.py --workload build --steps 4000 --warmup 64 --build-batch 64 --no-cpu > $O/build_b64_bench.log 2>&1
cp /tmp/p_b64/b64_kernel_stats.csv $O/build_b64_kernel_stats.csv 2>/dev/null
cd $R
(timeout 600 python bench.py) > $O/bench_default.log 2>&1
(timeout 600 python bench.py --workload build --steps 5000 --warmup 20) > $O/build_config3.log 2>&1
timeout 300 python tools/power_probe.py 3 2>&1 | grep -v '^/sys/class/drm\|amdgpu.ids' > $O/power_probe.txt
ls -la $O
