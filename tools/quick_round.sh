#!/bin/bash
# tests + single-stream and default bench lines (no rocprof)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout=600 ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest exit=$rc"; tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then exit 1; fi
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline > gpurun_out/bench_s1_graph.json 2> gpurun_out/bench_s1_graph.err || { tail -5 gpurun_out/bench_s1_graph.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_s1_graph.json'))
print('s1:', d['value'], 'c/s', d['ms_per_step'], 'ms/step')
for k,v in d['stage_ms_single_stream_eager'].items(): print('   ', k, v)
PY
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -5 gpurun_out/bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print('default:', d['value'], 'c/s', d['ms_per_step'], 'ms/step', 'roofline', d['roofline']['frac'])"
