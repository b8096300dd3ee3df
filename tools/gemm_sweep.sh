#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export REPS=${REPS:-40}
( for x in ${VARIANTS:-"RC_GEMM_F64Q_M=3" "RC_GEMM_F64Q_M=5" "RC_GEMM_F64Q_M=6"}; do
  env $x timeout -k 10 120 python tools/gemm_sweep.py 2>&1 | grep -E "K=8192|check" || exit 1
done ) > gpurun_out/gemm_sweep.log 2>&1
echo "sweep exit=$?"
cat gpurun_out/gemm_sweep.log
