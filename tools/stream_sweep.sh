#!/bin/bash
# headline throughput vs streams in flight and hardware queues (each line one bench run, no CPU baseline)
# PAIRS="queues:streams ..." or the full grid QUEUES x STREAMS
set -o pipefail
run() { GPU_MAX_HW_QUEUES=$1 timeout -k 10 200 python bench.py --streams $2 --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues $1 streams $2:', d['value'], d['ms_per_step'])" || exit 1; }
if [ -n "$PAIRS" ]; then for p in $PAIRS; do run ${p%%:*} ${p##*:}; done
else for q in ${QUEUES:-8 16 24}; do for s in ${STREAMS:-32 40 48 64}; do run $q $s; done; done; fi
