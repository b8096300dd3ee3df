#!/bin/bash
# headline throughput vs streams in flight and hardware queues (each line one bench run, no CPU baseline)
set -o pipefail
for q in ${QUEUES:-8 16 24}; do for s in ${STREAMS:-32 40 48 64}; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --streams $s --steps 256 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues $q streams $s:', d['value'], d['ms_per_step'])" || exit 1
done; done
