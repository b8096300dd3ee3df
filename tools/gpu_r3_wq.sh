#!/bin/bash
# round 3: staged k_wq_coop -- parity of the wide QRCP paths, staged vs single launch, timing, headline bench A/B
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=700 -k "wide or staged or cfg3 or graph" > $O/pytest_wq.log 2>&1
rc=$?; tail -4 $O/pytest_wq.log; [ $rc -ne 0 ] && exit 1
for st in 0 1; do
  RC_WQ_STAGES=$st timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $O/bench_wq$st.json 2> $O/bench_wq$st.err || { echo "bench failed"; tail -5 $O/bench_wq$st.err; exit 1; }
  python - <<PY
import json; d=json.load(open('$O/bench_wq$st.json'))
print('stages=$st:', d['value'], 'c/s frac', d['frac_of_f64_mfma_peak_whole_pipeline'], 'check', d['timed_results_check']['lanes_whose_last_replay_equals_their_eager_result_bitwise'], {k: v for k, v in d['stage_ms_single_stream_eager'].items() if 'wide_coop' in k})
PY
done
