cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s8
export AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1 AVLMAPS_SHARED_GPU_LOCK=/tmp/avl_gpu.lock
run() { # name nproc extra...
  name=$1; np=$2; shift 2
  timeout 600 env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus $np --workload build --steps 10000 --warmup 8 --no-cpu $EXTRA > gpurun_out/s8/$name.json 2> gpurun_out/s8/$name.err
  echo "rc=$?"
  grep "merge trace" gpurun_out/s8/$name.err | grep "rank 3 " | tail -4
  python tools/summarize_merge.py gpurun_out/s8/$name.json
}
EXTRA="" run r8_loop 8 A=1
EXTRA="--trajectory spiral" run r8_spiral 8 A=1
