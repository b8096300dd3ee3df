#!/bin/bash
# round 3: full GPU suite, then a headline A/B over one environment switch:  VAR=RC_QRCP_KEEP_DIRECT VALUES="0 1 0 1" bash tools/gpu_r3_ab.sh
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
if [ "${SUITE:-1}" = "1" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout=900 > $O/pytest_gpu.log 2>&1
  rc=$?; tail -4 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit 1
fi
for v in ${VALUES:-0 1 0 1}; do
  env $VAR=$v timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-gemm-lanes > $O/bench_ab_$v.json 2> $O/bench_ab_$v.err || { echo "bench failed"; tail -5 $O/bench_ab_$v.err; exit 1; }
  python - <<PY
import json; d=json.load(open('$O/bench_ab_$v.json'))
print('$VAR=$v:', d['value'], 'c/s frac', d['frac_of_f64_mfma_peak_whole_pipeline'], 'check', d['timed_results_check']['lanes_whose_last_replay_equals_their_eager_result_bitwise'])
PY
done
