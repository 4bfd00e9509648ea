cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in uniform clustered; do for env in 0 1; do
  if [ $env = 1 ]; then export AVL_HEAT_UNSORTED=1; else unset AVL_HEAT_UNSORTED; fi
  rm -rf /tmp/ph; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph -o h -- python $R/tools/probe_heat.py $kind 20 2>/dev/null | grep "ms per call"
  python - <<PY
import csv,glob
f=glob.glob('/tmp/ph/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if 'heat' in r['Name'] or 'rocprim' in r['Name'] or 'fill' in r['Name'].lower() or 'memset' in r['Name'].lower():
        print('    ', r['Name'][:70], r['Calls'], r['AverageNs'])
PY
done; done
