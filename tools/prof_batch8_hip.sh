#!/bin/bash
# HIP API trace of the cfg5 batch of 8 (tools/batch8_bench.py): which runtime calls the host spends a batch in
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_batch8_hip
cd /tmp && timeout -k 10 400 rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/prof_batch8_hip -- python3 $R/tools/batch8_bench.py > $R/gpurun_out/rocprof_batch8_hip.log 2>&1
echo "rocprof exit=$?"
cd $R
f=$(ls -t gpurun_out/prof_batch8_hip/*/*hip_api_stats.csv | head -1)
head -25 $f | tee gpurun_out/batch8_hip_api_stats.csv
tail -2 gpurun_out/rocprof_batch8_hip.log | cut -c1-200
