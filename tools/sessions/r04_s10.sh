cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s10
timeout 1500 python -m pytest tests/test_api_gpu.py tests/test_builder_gpu.py -m gpu -x -q > gpurun_out/s10/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/s10/pytest.log
timeout 600 python tools/probe_pipeline.py 2000 > gpurun_out/s10/pipeline.txt 2>&1; cat gpurun_out/s10/pipeline.txt | grep -v "Temporarily" | tail -12
