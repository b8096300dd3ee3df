#!/bin/bash
# cooperative blocked-QRCP panel: parity tests that reach it, then the cfg5 / cfg4 timings with and without it
mkdir -p gpurun_out
LOG=gpurun_out/qrcoop.log
: > $LOG
echo "== pytest (blocked / cfg5 / wide coop)" >> $LOG
RC_QRCP_DEBUG=${DBG:-0} timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=300 -k "blocked or cfg5 or wide or cfg4 or two_sided or pivoted" >> $LOG 2>&1
rc=$?; echo "pytest exit=$rc" >> $LOG
tail -25 $LOG
if [ $rc -ne 0 ]; then exit 1; fi
for coop in 1 0; do
  echo "== qrblk_bench RC_QRCP_COOP=$coop" >> $LOG
  RC_QRCP_COOP=$coop timeout -k 10 300 python tools/qrblk_bench.py > gpurun_out/qrblk_bench_coop$coop.log 2>&1 || { echo "bench failed"; tail -5 gpurun_out/qrblk_bench_coop$coop.log; exit 1; }
  grep -v "amdgpu.ids" gpurun_out/qrblk_bench_coop$coop.log | tail -32 | tee -a $LOG
done
