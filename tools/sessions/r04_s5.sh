cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
nproc; free -g | head -2
( while true; do echo "$(date +%s.%N | cut -c8-14) $(rocm-smi --showmeminfo vram --csv 2>/dev/null | tail -2 | head -1)"; sleep 0.5; done ) > gpurun_out/s5/vram.log 2>&1 &
SAMPLER=$!
export AVLMAPS_DIST_BACKEND=gloo AVLMAPS_MERGE_TRACE=1
run() { # name nproc extra...
  name=$1; np=$2; shift 2
  echo "== $name $(date +%s.%N | cut -c8-14)" | tee -a gpurun_out/s5/vram.log
  timeout 600 env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus $np --workload build --steps 10000 --warmup 8 --no-cpu $EXTRA > gpurun_out/s5/$name.json 2> gpurun_out/s5/$name.err
  echo "rc=$?"
  grep "merge trace" gpurun_out/s5/$name.err | grep "rank 0 " | tail -2
  python -c "
import json;d=json.loads([l for l in open('gpurun_out/s5/$name.json') if l.startswith('{')][-1]);print('$name', d['extra']['seconds'], d['extra']['merge_breakdown']['wall_s'])"
}
EXTRA="" run r8 8 A=1
EXTRA="" run r4 4 A=1
EXTRA="--no-exact-rgb" run r8_nolog 8 A=1
EXTRA="" run r8_nosdma 8 HSA_ENABLE_SDMA=0
EXTRA="--capacity 1300000" run r8_cap 8 A=1
kill $SAMPLER
dmesg 2>/dev/null | tail -20 > gpurun_out/s5/dmesg.txt
awk '{print $2}' gpurun_out/s5/vram.log | sort -t, -k3 -n | tail -3
