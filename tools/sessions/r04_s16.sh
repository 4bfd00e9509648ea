cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s16
timeout 600 python tools/probe_pipeline.py 2000 2>&1 | grep -v "Temporarily\|save:\|amdgpu" | tee gpurun_out/s16/pipeline.txt | tail -12
python - <<'PY'
import sys,time; sys.path.insert(0,'.')
import numpy as np
from avlmaps_amd.map.vlmap_builder import VLMapBuilder
np.random.seed(0)
for _ in range(3): VLMapBuilder.sample_pixels(777600,100)
t=time.perf_counter()
for _ in range(100): VLMapBuilder.sample_pixels(777600,100)
print("GPU-box host: shuffle_sample ms", (time.perf_counter()-t)/100*1e3)
t=time.perf_counter(); VLMapBuilder.skip_pixel_shuffles(200,777600); print("skip ms", (time.perf_counter()-t)/200*1e3)
import os
print(os.popen("grep -m1 'model name' /proc/cpuinfo").read())
PY
