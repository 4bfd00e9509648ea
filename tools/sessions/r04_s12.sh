cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s12
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/s12/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/s12/pytest_gpu.log
timeout 900 python tools/ab_sim.py --reps 3 --shapes 2000000x512x33,2000000x512x65 --modes raw,prepared,compact stock noxr > gpurun_out/s12/ab_q33_q65.txt 2>&1; tail -14 gpurun_out/s12/ab_q33_q65.txt
timeout 600 python tools/probe_pipeline.py 2000 > gpurun_out/s12/pipeline.txt 2>&1; grep -v "Temporarily\|save:" gpurun_out/s12/pipeline.txt | tail -11
