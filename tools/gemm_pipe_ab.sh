#!/bin/bash
# A/B of the f64 GEMM main loops on the two headline shapes: k_gemm_f64q (RC_GEMM_PIPE=0), the software-pipelined loop with
# register staging (RC_GEMM_PIPE=1 RC_GEMM_PIPE_DIRECT=0) and with direct-to-LDS copies (RC_GEMM_PIPE=1 RC_GEMM_PIPE_DIRECT=1).
mkdir -p gpurun_out
LOG=gpurun_out/gemm_pipe_ab.log
: > $LOG
for v in ${VARIANTS:-0:0 1:0 1:1}; do
  p=${v%%:*}; d=${v##*:}
  echo "== variant pipe=$p direct=$d" >> $LOG
  RC_GEMM_PIPE=$p RC_GEMM_PIPE_DIRECT=$d REPS=8 timeout -k 10 200 python tools/gemm_sweep.py >> $LOG 2>&1
  echo "== exit $?" >> $LOG
done
grep -v "amdgpu.ids\|splitk" $LOG
