#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo skip-pytest

( for x in "RC_GEMM_F64X4=0" "RC_GEMM_F64X4=1 RC_GEMM_TARGET_WGS=256" "RC_GEMM_F64X4=1 RC_GEMM_TARGET_WGS=512"; do
  env $x timeout -k 10 120 python tools/gemm_sweep.py 2>&1 | grep -E "k_gemm_mfma<f64>|check" || exit 1
done ) > gpurun_out/gemm_sweep.log 2>&1
echo "sweep exit=$?"
cat gpurun_out/gemm_sweep.log
