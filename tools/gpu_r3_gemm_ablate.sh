#!/bin/bash
# ablation of the ring GEMM's main loop (diagnostic build: -DRC_GEMM_PIPE_DEBUG); lone launches of the sketch shape
mkdir -p gpurun_out/r03
LOG=gpurun_out/r03/gemm_ablate.log
: > $LOG
for d in ${DBGS:-0 1 2 3 4 5 6 7 8 12}; do
  echo "== RC_GEMM_RING_DBG=$d" >> $LOG
  RC_GEMM_RING_DBG=$d REPS=8 timeout -k 10 120 python tools/gemm_sweep.py 2>&1 | grep "M=133 N=8192 K=8192" >> $LOG
done
cat $LOG
