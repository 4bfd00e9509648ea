cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_merge_kernels_gpu.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -5
