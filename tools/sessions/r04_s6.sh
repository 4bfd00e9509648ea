cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s6
export AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1
run() { # name nproc extra...
  name=$1; np=$2; shift 2
  timeout 600 env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus $np --workload build --steps 10000 --warmup 8 --no-cpu $EXTRA > gpurun_out/s6/$name.json 2> gpurun_out/s6/$name.err
  echo "rc=$?"
  python -c "
import json;d=json.loads([l for l in open('gpurun_out/s6/$name.json') if l.startswith('{')][-1]);print('$name', d['extra']['seconds'], d['extra']['merge_breakdown']['wall_s'])"
}
EXTRA="" run r8_poll_a 8 HSA_ENABLE_INTERRUPT=0
EXTRA="" run r8_poll_b 8 HSA_ENABLE_INTERRUPT=0
EXTRA="" run r8_poll_c 8 HSA_ENABLE_INTERRUPT=0
EXTRA="--trajectory spiral" run r8_poll_spiral 8 HSA_ENABLE_INTERRUPT=0
