#!/bin/bash
# Round-2 GPU session: parity tests, then the driver's bench command, then the default bench.  Steps are chained with &&-like
# exits: a failing / hanging GPU step stops the chain.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu $(date +%T)" | tee gpurun_out/progress.log
timeout -k 10 ${PYTEST_TIMEOUT:-1000} python -m pytest tests -m gpu -q --timeout=900 -x ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest exit=$rc $(date +%T)" | tee -a gpurun_out/progress.log
tail -15 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then exit 1; fi
if [ -n "$SKIP_BENCH" ]; then exit 0; fi
echo "== bench driver-style (--steps 20 --warmup 5)" | tee -a gpurun_out/progress.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err || { echo "bench failed"; tail -5 gpurun_out/bench_driver.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_driver.json')); print('driver-style:', d['value'], 'c/s', d['ms_per_step'], 'ms/step', 'frac', d['frac_of_f64_mfma_peak_whole_pipeline'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline_gemm_form']['value'], 'h2d', d['value_including_h2d'])"
echo "== smoke" | tee -a gpurun_out/progress.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
