cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s9
export AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1 AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock AVLMAPS_MERGE_PROFILE=3
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 8 --workload build --steps 10000 --warmup 8 --no-cpu --trajectory spiral > gpurun_out/s9/r8_spiral.json 2> gpurun_out/s9/r8_spiral.err
grep -A48 "Self CPU %" gpurun_out/s9/r8_spiral.err | cut -c1-200 | head -60
