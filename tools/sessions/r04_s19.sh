cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_builder_gpu.py -m gpu -x -q -k "heat" 2>&1 | tail -5
python tools/probe_heat.py uniform 20; python tools/probe_heat.py clustered 20
cd /tmp; export TMPDIR=/tmp
for kind in uniform clustered; do
  rm -rf /tmp/ph; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph -o h -- python $GRAFT_REPO_ROOT/tools/probe_heat.py $kind 20 2>/dev/null | grep "ms per call"
  python - <<PY
import csv,glob
f=glob.glob('/tmp/ph/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if 'heat' in r['Name'] or 'fill' in r['Name'].lower():
        print('    ', r['Name'][:70], r['Calls'], r['AverageNs'])
PY
done
