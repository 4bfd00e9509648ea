cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s21
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/s21/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s21/pytest_gpu.log
python __graft_entry__.py --smoke 2>&1 | tail -6
(time timeout 1500 python bench.py) > gpurun_out/s21/bench_default.log 2> gpurun_out/s21/bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/s21/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/s21/bench_default.log') if l.startswith('{')][-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_source'))
ex=d['extra']
print('heat', ex.get('heatmap_from_mask'))
print('pipeline', {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('frames_per_s','frame_loop_frames_per_s','final_save_s')}) for k,v in ex['vlmapbuilder_pipeline'].items() if k!='what'})
print('c5', ex['fused_multimodal_config5']['ms'], ex['fused_multimodal_config5']['frac_of_hbm_peak'], ex['fused_multimodal_config5']['compact_resident_copy']['ms'], ex['fused_multimodal_config5']['compact_resident_copy']['frac_of_hbm_peak'])
print('build', ex['map_build_strong']['frames_per_s'], ex['map_build_strong_deferred_fuse']['frames_per_s'], ex['map_build_strong_batched64']['frames_per_s'])
print('cpu', d['cpu_baseline']['value'], d['extra'].get('speedup_vs_cpu'))
PY
