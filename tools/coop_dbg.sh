#!/bin/bash
set -o pipefail
for d in 0 1 2 3 7; do
RC_COOP_DBG=$d timeout -k 10 120 python bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/coop_dbg$d.json || { echo "dbg $d failed"; }
python -c "
import json
d=json.load(open('gpurun_out/coop_dbg$d.json'))
print('dbg $d:', [ (k,v) for k,v in d['stage_ms_single_stream_eager'].items() if 'coop' in k])"
done
