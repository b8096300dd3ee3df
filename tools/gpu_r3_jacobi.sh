#!/bin/bash
# round 3: two-column block schedule of the fused Jacobi (k_jacobi_b2) -- SVD parity, timing, headline A/B
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_operator.py -m gpu -q -x --timeout=700 -k "svd or cfg3 or graph or smoke or fresh_workspace or operator or jacobi" > $O/pytest_jacobi.log 2>&1
rc=$?; tail -4 $O/pytest_jacobi.log; [ $rc -ne 0 ] && exit 1
for v in 0 1 0 1; do
  RC_JACOBI_BLOCK2=$v timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $O/bench_jb$v.json 2> $O/bench_jb$v.err || { echo "bench failed"; tail -5 $O/bench_jb$v.err; exit 1; }
  python - <<PY
import json; d=json.load(open('$O/bench_jb$v.json'))
print('BLOCK2=$v:', d['value'], 'c/s frac', d['frac_of_f64_mfma_peak_whole_pipeline'], 'check', d['timed_results_check']['lanes_whose_last_replay_equals_their_eager_result_bitwise'], {k: v for k, v in d['stage_ms_single_stream_eager'].items() if 'jacobi' in k})
PY
done
