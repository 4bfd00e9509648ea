cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s13
timeout 900 python tools/ab_sim.py --reps 3 --shapes 2000000x1024x64 --modes raw,prepared stock env:AVL_SIM_PAIR=1 > gpurun_out/s13/ab_pair.txt 2>&1; tail -8 gpurun_out/s13/ab_pair.txt
timeout 600 python -m pytest tests/test_api_gpu.py -m gpu -x -q -k "reproduces_reference_map" 2>&1 | tail -3
