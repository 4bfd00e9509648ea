set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
timeout 1500 python -m pytest tests/test_api_gpu.py tests/test_builder_gpu.py -m gpu -x -q -k "multi_rank or rank or merge or checkpoint or sharded or bench_two" > gpurun_out/s1/pytest_ranks.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s1/pytest_ranks.log
tail -5 gpurun_out/s1/pytest_ranks.log
timeout 900 python tools/ab_sim.py --reps 3 --shapes 2000000x512x64 --modes raw,prepared,compact stock hionly > gpurun_out/s1/ab_hionly.txt 2>&1
tail -8 gpurun_out/s1/ab_hionly.txt
export AVLMAPS_DIST_BACKEND=gloo
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu > gpurun_out/s1/build_8ranks.json 2> gpurun_out/s1/build_8ranks.err; echo rc=$?
tail -c 3000 gpurun_out/s1/build_8ranks.json; tail -5 gpurun_out/s1/build_8ranks.err
