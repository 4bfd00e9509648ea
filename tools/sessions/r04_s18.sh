cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s18
timeout 900 python tools/soak_pipeline.py 420 1 > gpurun_out/s18/soak_pipeline.txt 2>&1; echo "soak rc=$?"; tail -5 gpurun_out/s18/soak_pipeline.txt
timeout 900 python tools/fuzz_parity.py 420 4 > gpurun_out/s18/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -4 gpurun_out/s18/fuzz.txt
