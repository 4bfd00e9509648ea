#!/bin/bash
# GPU box: collect the rocprofv3 evidence for bench.py into gpurun_out/prof_<tag>/ (tools/publish_profiles.py then copies the
# judged summaries to profiles/<tag>_*).   usage: tools/collect_profiles.sh <tag>
# Kernel traces (--kernel-trace --stats) and PMC passes (--pmc only) are separate runs: never combined with tracing flags.
set -u
TAG=${1:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
run() { timeout 420 "$@"; }
trace() {   # trace <name> <bench args...>
  local name=$1; shift
  run rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o $name -- python $R/bench.py "$@" > $O/${name}_bench.log 2>&1
  cp /tmp/p_$name/${name}_kernel_stats.csv $O/ 2>/dev/null
}
pmc() {     # pmc <name> "<counters>" <bench args...>
  local name=$1; local cnt=$2; shift 2
  run rocprofv3 --pmc $cnt --output-format csv -d /tmp/q_$name -o $name -- python $R/bench.py "$@" > $O/pmc_${name}.log 2>&1
}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
# --- config 2 (headline): 2M x 512, 64 queries
trace index --profile-run
pmc index_fetch FETCH_SIZE --steps 5 --warmup 2 --settle-steps 0 --profile-run
pmc index_write WRITE_SIZE --steps 5 --warmup 2 --settle-steps 0 --profile-run
pmc index_sq "$SQ" --steps 5 --warmup 2 --settle-steps 0 --profile-run
python $R/tools/summarize_prof.py $O/pmc_index.json /tmp/q_index_fetch /tmp/q_index_write /tmp/q_index_sq > /dev/null
# --- the same shape on VLMap's compact resident copy (3 B per element), and the reference's 65-column query ("64 categories + other")
trace index_compact --profile-run --resident compact
pmc ic_fetch FETCH_SIZE --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
pmc ic_write WRITE_SIZE --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
pmc ic_sq "$SQ" --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
python $R/tools/summarize_prof.py $O/pmc_index_compact.json /tmp/q_ic_fetch /tmp/q_ic_write /tmp/q_ic_sq > /dev/null
trace index_q65 --profile-run --queries 65
# --- config 5: 2M x 1536, 128 block-structured queries (column-block launches) and the dense single pass
C5="--feat-dim 1536 --queries 128"
trace config5 $C5 --steps 100 --profile-run
trace config5_dense $C5 --steps 100 --profile-run --dense
pmc c5_fetch FETCH_SIZE $C5 --steps 5 --warmup 2 --settle-steps 0 --profile-run
pmc c5_write WRITE_SIZE $C5 --steps 5 --warmup 2 --settle-steps 0 --profile-run
pmc c5_sq "$SQ" $C5 --steps 5 --warmup 2 --settle-steps 0 --profile-run
python $R/tools/summarize_prof.py $O/pmc_config5.json /tmp/q_c5_fetch /tmp/q_c5_write /tmp/q_c5_sq > /dev/null
trace config5_compact $C5 --steps 100 --profile-run --resident compact
pmc c5c_fetch FETCH_SIZE $C5 --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
pmc c5c_write WRITE_SIZE $C5 --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
pmc c5c_sq "$SQ" $C5 --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
python $R/tools/summarize_prof.py $O/pmc_config5_compact.json /tmp/q_c5c_fetch /tmp/q_c5c_write /tmp/q_c5c_sq > /dev/null
# --- build: per-frame launches and 16 frames per launch, 10k frames, finalize inside the timed region
trace build --workload build --steps 10000 --warmup 20 --no-cpu
trace build_b16 --workload build --steps 10000 --warmup 20 --no-cpu --build-batch 16
trace build_b64 --workload build --steps 10000 --warmup 64 --no-cpu --build-batch 64
trace build_deferred --workload build --steps 10000 --warmup 20 --no-cpu --deferred-fuse
pmc build_fetch FETCH_SIZE --workload build --steps 200 --warmup 5 --no-cpu
pmc build_write WRITE_SIZE --workload build --steps 200 --warmup 5 --no-cpu
python $R/tools/summarize_prof.py $O/pmc_build.json /tmp/q_build_fetch /tmp/q_build_write > /dev/null
pmc build16_fetch FETCH_SIZE --workload build --steps 320 --warmup 16 --no-cpu --build-batch 16
pmc build16_write WRITE_SIZE --workload build --steps 320 --warmup 16 --no-cpu --build-batch 16
python $R/tools/summarize_prof.py $O/pmc_build_b16.json /tmp/q_build16_fetch /tmp/q_build16_write > /dev/null
cd $R
# --- plain runs (no profiler): the lines the judge reads
(timeout 900 python bench.py) > $O/bench_default.log 2>&1
(timeout 600 python bench.py $C5 --steps 200 --no-pmc) > $O/config5_line.log 2>&1
(timeout 600 python bench.py $C5 --steps 200 --resident compact --no-pmc) > $O/config5_compact_line.log 2>&1
(timeout 600 python bench.py --resident compact --no-build-extra) > $O/index_compact_line.log 2>&1
(timeout 600 python bench.py --queries 65 --no-build-extra --no-pmc) > $O/index_q65_line.log 2>&1
(timeout 600 python bench.py --workload build --steps 10000) > $O/build_10k.log 2>&1
(timeout 600 python bench.py --workload build --steps 5000) > $O/build_config3.log 2>&1
(timeout 600 python bench.py --workload build --steps 10000 --deferred-fuse --no-cpu) > $O/build_10k_deferred.log 2>&1
(timeout 600 python bench.py --workload build --steps 10000 --build-batch 64 --no-cpu) > $O/build_10k_b64.log 2>&1
(timeout 600 python bench.py --workload build --steps 40000 --build-batch 16 --no-cpu) > $O/build_40k_b16.log 2>&1
AVLMAPS_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 timeout 600 python bench.py --workload build --steps 10000 --build-batch 16 --no-cpu > $O/build_rccl_1rank.log 2>&1
AVLMAPS_DIST_BACKEND=nccl timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --workload build --steps 2000 --no-cpu > $O/rccl_two_ranks_one_gpu.log 2>&1
# --- round 4: eight ranks on this ONE GPU (gloo; a cross-process lock serialises the ranks' compute, see parallel._SharedGpuLock): the merge's
#     choreography and per-rank compute, on the loop (every shard sees the whole map) and on exploration trajectories
for cfg in "loop:" "spiral:--trajectory spiral" "spiral4:--trajectory spiral --spiral-radius 4"; do
  name=${cfg%%:*}; extra=${cfg#*:}
  (timeout 600 python bench.py --workload build --steps 10000 --no-cpu $extra) > $O/build_10k_${name}_1rank.log 2>&1
  AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1 AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu $extra > $O/build_8ranks_one_gpu_${name}.log 2> $O/build_8ranks_one_gpu_${name}.err
  grep "merge trace" $O/build_8ranks_one_gpu_${name}.err > $O/build_8ranks_one_gpu_${name}_trace.txt
  python tools/summarize_merge.py $O/build_8ranks_one_gpu_${name}.log > $O/build_8ranks_one_gpu_${name}_summary.txt 2>&1
done
AVLMAPS_DIST_BACKEND=gloo AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29714 bench.py --gpus 2 --no-pmc --no-cpu > $O/index_2ranks_one_gpu.log 2> $O/index_2ranks_one_gpu.err
timeout 600 python tools/probe_pipeline.py 2000 2>&1 | grep -v "Temporarily\|amdgpu.ids" > $O/pipeline_probe.txt
timeout 300 python tools/power_probe.py 3 2>&1 | grep -v '^/sys/class/drm\|amdgpu.ids' > $O/power_probe.txt
ls -la $O
