#!/bin/bash
# round 3: folded small factors in the tall QRs (RC_TSQR_FOLD) and Z of the ID in one launch (RC_ID_FUSED): full GPU suite, headline A/B
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout=900 > $O/pytest_fold.log 2>&1
rc=$?; tail -4 $O/pytest_fold.log; [ $rc -ne 0 ] && exit 1
for v in ${VARIANTS:-"0 0" "1 0" "0 1" "1 1" "0 0" "1 1"}; do
  set -- $v
  RC_TSQR_FOLD=$1 RC_ID_FUSED=$2 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $O/bench_fold$1$2.json 2> $O/bench_fold$1$2.err || { echo "bench failed"; tail -5 $O/bench_fold$1$2.err; exit 1; }
  python - <<PY
import json; d=json.load(open('$O/bench_fold$1$2.json'))
print('fold=$1 idfused=$2:', d['value'], 'c/s frac', d['frac_of_f64_mfma_peak_whole_pipeline'], 'check', d['timed_results_check']['lanes_whose_last_replay_equals_their_eager_result_bitwise'])
PY
done
