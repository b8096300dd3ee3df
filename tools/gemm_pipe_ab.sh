#!/bin/bash
# A/B of the f64 GEMM main loops on the two headline shapes; VARIANTS = ';'-separated lists of ','-separated env settings
mkdir -p gpurun_out
LOG=gpurun_out/gemm_pipe_ab.log
: > $LOG
IFS=';'
for v in ${VARIANTS:-RC_GEMM_PIPE=0;RC_GEMM_PIPE=1}; do
  echo "== $v" >> $LOG
  env $(echo $v | tr ',' ' ') REPS=8 timeout -k 10 200 python tools/gemm_sweep.py >> $LOG 2>&1
  echo "== exit $?" >> $LOG
done
grep -v "amdgpu.ids\|splitk_reduce M=128" $LOG
