#!/bin/bash
# Round-3 GPU session 1: parity suite on the memset-free library, the health-word isolation runs, the new bench paths.
set -o pipefail
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
O=gpurun_out/r03
step() { echo "== $1 $(date +%T)" | tee -a $O/progress.log; }
step "pytest -m gpu"
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout=900 -x > $O/pytest_gpu.log 2>&1
rc=$?; tail -4 $O/pytest_gpu.log; echo "pytest exit=$rc" | tee -a $O/progress.log
[ $rc -ne 0 ] && exit 1
for mode in 1 2 0; do
  step "rsvd_repeat_diag RC_DEBUG_MEMSET_PATH=$mode"
  RC_DEBUG_MEMSET_PATH=$mode LANES=16 ROUNDS=40 timeout -k 10 300 python tools/rsvd_repeat_diag.py > $O/memset_path_$mode.log 2>&1 || { echo "diag mode $mode failed"; tail -5 $O/memset_path_$mode.log; exit 1; }
  grep -c health $O/memset_path_$mode.log | sed "s/^/health lines: /"; grep health $O/memset_path_$mode.log | sort | uniq -c | sort -rn | sed -n 1,6p; tail -1 $O/memset_path_$mode.log
done
step "bench default"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { echo "bench failed"; tail -5 $O/bench_driver.err; exit 1; }
python - <<'PY'
import json; d=json.load(open('gpurun_out/r03/bench_driver.json'))
print('cfg3:', d['value'], 'c/s', d['ms_per_step'], 'ms/step frac', d['frac_of_f64_mfma_peak_whole_pipeline'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'check', d['timed_results_check'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['cores_source'])
PY
step "bench --gpus 2 rehearsal (self-launched)"
RC_BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --gpus 2 --steps 6 --warmup 2 --streams 20 --no-cpu-baseline > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2_rehearsal.err || { echo "rehearsal failed"; tail -8 $O/bench_gpus2_rehearsal.err; exit 1; }
cat $O/bench_gpus2_rehearsal.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rehearsal n_gpus', d['n_gpus'], d['value'], d['timed_results_check']['lanes_whose_last_replay_equals_their_eager_result_bitwise'])"
step "bench cfg5"
timeout -k 10 400 python bench.py --config cfg5 --steps 10 --warmup 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err || { echo "cfg5 failed"; tail -8 $O/bench_cfg5.err; exit 1; }
cat $O/bench_cfg5.json
step "bench cfg5 --gpus 2 rehearsal"
RC_BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --config cfg5 --gpus 2 --steps 5 --warmup 1 > $O/bench_cfg5_gpus2.json 2> $O/bench_cfg5_gpus2.err || { echo "cfg5 rehearsal failed"; tail -8 $O/bench_cfg5_gpus2.err; exit 1; }
cat $O/bench_cfg5_gpus2.json
step "bench cfg5 one-rank RCCL"
RC_BENCH_FORCE_DIST=1 timeout -k 10 400 python bench.py --config cfg5 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg5_rccl1.json 2> $O/bench_cfg5_rccl1.err || { echo "cfg5 rccl1 failed"; tail -8 $O/bench_cfg5_rccl1.err; exit 1; }
cat $O/bench_cfg5_rccl1.json
step done
