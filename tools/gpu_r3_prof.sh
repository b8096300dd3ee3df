#!/bin/bash
# round-3 profiles: one-stream and 42-stream kernel traces of the headline bench + timeline analysis (tools/timeline.py)
set -o pipefail
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in ${STREAMS_LIST:-1 42}; do
  rm -rf gpurun_out/r03/prof_s$S
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/prof_s$S -- python3 $R/bench.py --streams $S --steps 16 --warmup 4 --no-cpu-baseline --no-h2d > $R/gpurun_out/r03/rocprof_s$S.log 2>&1 )
  rc=$?; echo "rocprof streams=$S exit=$rc"
  [ $rc -ne 0 ] && { tail -5 gpurun_out/r03/rocprof_s$S.log; exit 1; }
  f=$(ls -t gpurun_out/r03/prof_s$S/*/*kernel_trace.csv | head -1)
  python tools/timeline.py $f --frac 0.4 --top 40 > gpurun_out/r03/timeline_s$S.txt
  cp $(ls -t gpurun_out/r03/prof_s$S/*/*kernel_stats.csv | head -1) gpurun_out/r03/kernel_stats_s$S.csv
  tail -1 gpurun_out/r03/rocprof_s$S.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'frac', d['frac_of_f64_mfma_peak_whole_pipeline'])"
  # the raw traces are large: keep the summaries only
  rm -rf gpurun_out/r03/prof_s$S
done
cat gpurun_out/r03/timeline_s42.txt
