#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_s1
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s1 -- python3 $R/bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline > $R/gpurun_out/rocprof_s1.log 2>&1
echo "rocprof exit=$?"
cd $R && python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_s1/*/*kernel_stats.csv'))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(r['Name'][:78].ljust(78), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1000)).rjust(9),'us', r['Percentage'])
PY
