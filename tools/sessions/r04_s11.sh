cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s11
timeout 600 python tools/probe_pipeline.py 2000 > gpurun_out/s11/pipeline.txt 2>&1; cat gpurun_out/s11/pipeline.txt | grep -v "Temporarily\|save:" | tail -14
