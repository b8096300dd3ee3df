#!/bin/bash
# kernel-trace of the cfg5 batch of 8 (tools/batch8_bench.py) + timeline analysis (tools/timeline.py): where a batch's time goes
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_batch8
timeout -k 10 200 python3 $R/tools/batch8_bench.py 2>&1 | tail -1
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_batch8 -- python3 $R/tools/batch8_bench.py > $R/gpurun_out/rocprof_batch8.log 2>&1
echo "rocprof exit=$?"
cd $R
f=$(ls -t gpurun_out/prof_batch8/*/*kernel_trace.csv | head -1)
python tools/timeline.py $f --frac 0.5 --top 40 | tee gpurun_out/timeline_batch8.txt
cp $f gpurun_out/batch8_kernel_trace.csv
tail -2 gpurun_out/rocprof_batch8.log
