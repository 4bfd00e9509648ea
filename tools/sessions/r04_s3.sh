set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
export AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1
for n in 2000 10000; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 8 --workload build --steps $n --warmup 8 --no-cpu > gpurun_out/s3/build_8ranks_$n.json 2> gpurun_out/s3/build_8ranks_$n.err; echo rc=$?
grep "merge trace" gpurun_out/s3/build_8ranks_$n.err | grep "rank [03] plan" | tail -4
done
