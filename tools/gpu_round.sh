#!/bin/bash
# One GPU-box session: parity tests, default bench, single-stream bench, rocprof kernel trace of the bench command.
# Every step writes under gpurun_out/ ; a failing/hanging GPU step stops the chain.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu" | tee gpurun_out/progress.log
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest exit=$rc" | tee -a gpurun_out/progress.log
tail -3 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then exit 1; fi
echo "== bench streams=1" | tee -a gpurun_out/progress.log
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline > gpurun_out/bench_s1_graph.json 2> gpurun_out/bench_s1_graph.err || { echo "bench s1 failed"; tail -5 gpurun_out/bench_s1_graph.err; exit 1; }
echo "== bench default" | tee -a gpurun_out/progress.log
timeout -k 10 500 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { echo "bench default failed"; tail -5 gpurun_out/bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print('default:', d['value'], 'c/s', d['ms_per_step'], 'ms/step', 'host issue', d['host_issue_ms_per_step'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'cpu', d['cpu_baseline'])"
echo "== rocprofv3 kernel trace of the bench command" | tee -a gpurun_out/progress.log
rm -rf gpurun_out/prof_bench
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 84 --warmup 42 --no-cpu-baseline > $R/gpurun_out/rocprof_bench.log 2>&1
echo "rocprof exit=$?" | tee -a $R/gpurun_out/progress.log
cd $R && python - <<'PY'
import csv,glob,os
f=sorted(glob.glob('gpurun_out/prof_bench/*/*kernel_stats.csv'), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1000)).rjust(9),'us', r['Percentage'])
PY
