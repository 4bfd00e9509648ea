#!/bin/bash
# GPU box, round 2 call A: builder / merge tests, build bench, RCCL-on-one-GPU probe
cd "$(dirname "$0")/.."
O=gpurun_out/r02a; mkdir -p $O
python -c "import h5py" > $O/h5py_probe.txt 2>&1; ls /opt/conda/lib/libhdf5.so* >> $O/h5py_probe.txt 2>&1
timeout 1500 python -m pytest tests/test_builder_gpu.py tests/test_api_gpu.py -x -q -m gpu > $O/pytest_build.log 2>&1; echo "pytest rc=$?" >> $O/pytest_build.log
tail -30 $O/pytest_build.log
timeout 600 python bench.py --workload build --steps 10000 --warmup 20 --no-cpu > $O/build_10k.json 2> $O/build_10k.err
timeout 600 python bench.py --workload build --steps 10000 --warmup 20 --no-cpu --build-batch 16 > $O/build_10k_b16.json 2> $O/build_10k_b16.err
timeout 600 python bench.py --workload build --steps 40000 --warmup 20 --no-cpu --build-batch 16 > $O/build_40k_b16.json 2> $O/build_40k_b16.err
# two ranks on ONE GPU: gloo (staged through the host) and real RCCL
AVLMAPS_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --workload build --steps 10000 --warmup 20 --no-cpu > $O/build_10k_2ranks_gloo.json 2> $O/build_10k_2ranks_gloo.err
AVLMAPS_DIST_BACKEND=nccl timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --workload build --steps 2000 --warmup 20 --no-cpu > $O/build_2ranks_rccl.json 2> $O/build_2ranks_rccl.err
echo "rccl rc=$?" >> $O/build_2ranks_rccl.err
tail -5 $O/build_2ranks_rccl.err
for f in $O/*.json; do echo "== $f"; head -c 1800 $f; echo; done
