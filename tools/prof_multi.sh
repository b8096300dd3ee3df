#!/bin/bash
# kernel-trace of the multi-stream bench + timeline analysis (tools/timeline.py)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${STREAMS:-16}
rm -rf gpurun_out/prof_multi
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_multi -- python3 $R/bench.py --streams $S --steps 16 --warmup 4 --no-cpu-baseline --no-h2d > $R/gpurun_out/rocprof_multi.log 2>&1
echo "rocprof exit=$?"
cd $R
f=$(ls -t gpurun_out/prof_multi/*/*kernel_trace.csv | head -1)
python tools/timeline.py $f --frac 0.25 | tee gpurun_out/timeline_s$S.txt
tail -2 gpurun_out/rocprof_multi.log
