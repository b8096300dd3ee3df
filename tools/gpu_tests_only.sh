#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout=900 --durations=8 ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit=$?"; tail -25 gpurun_out/pytest_gpu.log
