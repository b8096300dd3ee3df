#!/bin/bash
# bench at several stream counts (no CPU baseline); one line each
set -o pipefail
mkdir -p gpurun_out
for s in ${STREAMS_LIST:-1 16 32}; do
  timeout -k 10 300 python bench.py --streams $s --steps ${STEPS:-128} --no-cpu-baseline ${EXTRA:-} > gpurun_out/bench_s$s.json 2> gpurun_out/bench_s$s.err || { echo "bench s$s failed"; tail -5 gpurun_out/bench_s$s.err; exit 1; }
  python - $s <<'PY'
import json,sys
s=sys.argv[1]
d=json.load(open(f'gpurun_out/bench_s{s}.json'))
print(f"streams {s}: {d['value']} c/s {d['ms_per_step']} ms/step  gemm frac {d['roofline']['frac']}")
if s=='1':
    for k,v in d['stage_ms_single_stream_eager'].items(): print('   ',k,v)
PY
done
