timeout -k 10 300 python -m pytest tests -m gpu -q -x --timeout=300 -k "gemm or golden or fused or headline" > gpurun_out/pytest_gemm.log 2>&1; tail -2 gpurun_out/pytest_gemm.log
grep -q failed gpurun_out/pytest_gemm.log && exit 1
REPS=30 VARIANTS="RC_GEMM_GLDS=0 RC_GEMM_GLDS=1 RC_GEMM_GLDS=2 RC_GEMM_GLDS=0 RC_GEMM_GLDS=1 RC_GEMM_GLDS=2" bash tools/gemm_sweep.sh | grep "M=128 N=8192"
