set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
for tr in loop spiral; do
timeout 900 python bench.py --gpus 1 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory $tr > gpurun_out/s4/build_1rank_$tr.json 2> gpurun_out/s4/build_1rank_$tr.err; echo rc=$?
done
export AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1
for tr in loop spiral; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory $tr > gpurun_out/s4/build_8ranks_$tr.json 2> gpurun_out/s4/build_8ranks_$tr.err; echo rc=$?
grep "merge trace" gpurun_out/s4/build_8ranks_$tr.err | grep "rank [03] " | tail -4
done
