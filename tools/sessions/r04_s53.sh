#!/bin/bash
# compact-form evidence after the two-plane layout (subset of tools/collect_profiles.sh)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run() { timeout 420 "$@"; }
trace() { local name=$1; shift; run rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o $name -- python $R/bench.py "$@" > $O/${name}_bench.log 2>&1; cp /tmp/p_$name/${name}_kernel_stats.csv $O/ 2>/dev/null; }
pmc() { local name=$1; local cnt=$2; shift 2; run rocprofv3 --pmc $cnt --output-format csv -d /tmp/q_$name -o $name -- python $R/bench.py "$@" > $O/pmc_${name}.log 2>&1; }
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
C5="--feat-dim 1536 --queries 128"
trace index_compact --profile-run --resident compact
pmc ic_fetch FETCH_SIZE --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
pmc ic_write WRITE_SIZE --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
pmc ic_sq "$SQ" --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
python $R/tools/summarize_prof.py $O/pmc_index_compact.json /tmp/q_ic_fetch /tmp/q_ic_write /tmp/q_ic_sq > /dev/null
trace config5_compact $C5 --steps 100 --profile-run --resident compact
pmc c5c_fetch FETCH_SIZE $C5 --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
pmc c5c_write WRITE_SIZE $C5 --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
pmc c5c_sq "$SQ" $C5 --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
python $R/tools/summarize_prof.py $O/pmc_config5_compact.json /tmp/q_c5c_fetch /tmp/q_c5c_write /tmp/q_c5c_sq > /dev/null
pmc ic_tcc "TCC_REQ_sum" --steps 5 --warmup 2 --settle-steps 0 --profile-run --resident compact
f=$(find /tmp/q_ic_tcc -name "*counter_collection.csv" | head -1)
python - "$f" > $O/tcc_compact_new.txt <<'PY'
import csv, sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "sim_split" in (r.get("Kernel_Name") or "") and "prepare" not in (r.get("Kernel_Name") or "")]
print("compact (two planes per row) TCC_REQ_sum", sum(v)/len(v), len(v))
PY
cd $R
(timeout 900 python bench.py) > $O/bench_default.log 2>&1
(timeout 600 python bench.py $C5 --steps 200 --resident compact --no-pmc) > $O/config5_compact_line.log 2>&1
(timeout 600 python bench.py --resident compact --no-build-extra) > $O/index_compact_line.log 2>&1
