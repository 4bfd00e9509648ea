set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
python tools/probe_merge_plan.py 1250000 > gpurun_out/s2/probe_plan_gpu.txt 2>&1
head -40 gpurun_out/s2/probe_plan_gpu.txt
export AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu > gpurun_out/s2/build_8ranks.json 2> gpurun_out/s2/build_8ranks.err; echo rc=$?
grep "merge trace" gpurun_out/s2/build_8ranks.err | tail -20
