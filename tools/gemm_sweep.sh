#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
( for x in "RC_GEMM_F64Q_N=0 RC_GEMM_F64Q_M=0" "RC_GEMM_F64Q_N=1 RC_GEMM_F64Q_M=1" "RC_GEMM_F64Q_N=1 RC_GEMM_F64Q_M=1 RC_GEMM_TARGET_WGS=512"; do
  env $x timeout -k 10 120 python tools/gemm_sweep.py 2>&1 | grep -E "k_gemm_mfma<f64> M=8192 N=133|k_gemm_mfma<f64> M=128 N=8192|check" || exit 1
done ) > gpurun_out/gemm_sweep.log 2>&1
echo "sweep exit=$?"
cat gpurun_out/gemm_sweep.log
