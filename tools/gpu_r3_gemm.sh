#!/bin/bash
# Round-3: the three-stage-ring GEMM (k_gemm_f64r) -- parity on its shapes, A/B against k_gemm_f64a, headline bench with it.
set -o pipefail
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
O=gpurun_out/r03
step() { echo "== $1 $(date +%T)" | tee -a $O/progress_gemm.log; }
step "gemm parity tests"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=600 -x -k "gemm or headline_size_sketch or cfg3" > $O/pytest_gemm.log 2>&1
rc=$?; tail -4 $O/pytest_gemm.log; [ $rc -ne 0 ] && exit 1
step "A/B lone launches"
VARIANTS="${VARIANTS:-RC_GEMM_RING=0;RC_GEMM_RING=1}" bash tools/gemm_pipe_ab.sh > $O/gemm_ab.log 2>&1; grep -v "^check\|amdgpu" $O/gemm_ab.log | grep "==\|k_gemm_mfma<f64> M=13[36] N=8192\|k_gemm_mfma<f64> M=128 N=8192 K=8192\|splitk_reduce M=13"
for ring in ${RINGS:-0 1}; do
  step "bench RC_GEMM_RING=$ring"
  RC_GEMM_RING=$ring timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $O/bench_ring$ring.json 2> $O/bench_ring$ring.err || { echo "bench failed"; tail -5 $O/bench_ring$ring.err; exit 1; }
  python - <<PY
import json; d=json.load(open('$O/bench_ring$ring.json'))
print('ring$ring:', d['value'], 'c/s', d['ms_per_step'], 'ms/step frac', d['frac_of_f64_mfma_peak_whole_pipeline'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel'][:40], 'check', d['timed_results_check']['lanes_whose_last_replay_equals_their_eager_result_bitwise'])
PY
done
step done
