#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export REPS=40
( for x in "RC_GEMM_SWAP_SKINNY=0" "RC_GEMM_SWAP_SKINNY=1 RC_GEMM_F64Q_M=4" "RC_GEMM_SWAP_SKINNY=0" "RC_GEMM_SWAP_SKINNY=1 RC_GEMM_F64Q_M=4" "RC_GEMM_SWAP_SKINNY=1 RC_GEMM_F64Q_M=3"; do
  env $x timeout -k 10 120 python tools/gemm_sweep.py 2>&1 | grep -E "K=8192" || exit 1
done ) > gpurun_out/gemm_sweep.log 2>&1
echo "sweep exit=$?"
cat gpurun_out/gemm_sweep.log
