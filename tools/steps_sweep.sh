#!/bin/bash
# headline throughput vs the number of timed steps: the timed region starts with every lane idle and in lockstep (barrier +
# synchronize), and it takes several rounds until the lanes are out of phase and big GEMMs overlap the single-CU chains
set -o pipefail
for k in ${STEPS:-42 84 168 256 512 1024 2048}; do
  timeout -k 10 200 python bench.py --steps $k --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $k:', d['value'], d['ms_per_step'])" || exit 1
done
