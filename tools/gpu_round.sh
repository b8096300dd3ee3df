#!/bin/bash
# One GPU-box session: parity tests, bench variants, rocprof kernel trace.
# Every step writes under gpurun_out/ ; a failing/hanging GPU step stops the chain.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu" | tee gpurun_out/progress.log
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest exit=$rc" | tee -a gpurun_out/progress.log
tail -15 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then exit 1; fi
echo "== bench streams=1 graph" | tee -a gpurun_out/progress.log
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline > gpurun_out/bench_s1_graph.json 2> gpurun_out/bench_s1_graph.err || { echo "bench s1 failed"; tail -5 gpurun_out/bench_s1_graph.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_s1_graph.json'))
print('s1:', d['value'], 'c/s', d['ms_per_step'], 'ms/step')
for k,v in d['stage_ms_single_stream_eager'].items(): print('   ', k, v)
print('   roofline', d['roofline']['achieved'], d['roofline']['frac'])
PY
for lpp in 4 8 16; do
RC_JACOBI_LPP=$lpp timeout -k 10 300 python bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/bench_lpp$lpp.json 2> gpurun_out/bench_lpp$lpp.err
python -c "
import json; d=json.load(open('gpurun_out/bench_lpp$lpp.json')); print('lpp $lpp:', d['ms_per_step'], 'ms/step; jacobi', d['stage_ms_single_stream_eager']['op:jacobi_svd n=128'], 'sweeps', d['stage_ms_single_stream_eager'].get('info:jacobi_sweeps n=128'))"
done
echo "== bench default" | tee -a gpurun_out/progress.log
timeout -k 10 500 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { echo "bench default failed"; tail -5 gpurun_out/bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print('default:', d['value'], 'c/s', d['ms_per_step'], 'ms/step', 'host issue', d['host_issue_ms_per_step'], 'gemm frac', d['roofline']['frac'], 'cpu', d['cpu_baseline'])"
echo "== bench streams=8" | tee -a gpurun_out/progress.log
timeout -k 10 300 python bench.py --streams 8 --no-cpu-baseline > gpurun_out/bench_s8.json 2> gpurun_out/bench_s8.err
python -c "
import json; d=json.load(open('gpurun_out/bench_s8.json')); print('s8:', d['value'], 'c/s', d['ms_per_step'], 'ms/step')"
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python bench.py --streams 8 --no-cpu-baseline > gpurun_out/bench_s8q8.json 2> gpurun_out/bench_s8q8.err
python -c "
import json; d=json.load(open('gpurun_out/bench_s8q8.json')); print('s8 hwq8:', d['value'], 'c/s', d['ms_per_step'], 'ms/step')"
GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python bench.py --streams 16 --steps 64 --no-cpu-baseline > gpurun_out/bench_s16q16.json 2> gpurun_out/bench_s16q16.err
python -c "
import json; d=json.load(open('gpurun_out/bench_s16q16.json')); print('s16 hwq16:', d['value'], 'c/s', d['ms_per_step'], 'ms/step', 'host issue', d['host_issue_ms_per_step'])"
echo "== rocprofv3 kernel trace of bench" | tee -a gpurun_out/progress.log
rm -rf gpurun_out/prof_bench
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_bench.log 2>&1
echo "rocprof exit=$?" | tee -a $R/gpurun_out/progress.log
cd $R && python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_bench/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1000)).rjust(9),'us', r['Percentage'])
PY
