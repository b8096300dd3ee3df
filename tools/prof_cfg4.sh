#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_cfg4
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg4 -- python3 $R/tools/qrblk_cfg4_debug.py > $R/gpurun_out/prof_cfg4.log 2>&1
echo "rocprof exit=$?"
cd $R
grep -v "^W2026\|^E2026" gpurun_out/prof_cfg4.log | tail -5
python - <<'PY'
import csv,glob,os
f=sorted(glob.glob('gpurun_out/prof_cfg4/*/*kernel_stats.csv'), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1000)).rjust(9),'us', ('%.1f'%(float(r['TotalDurationNs'])/1e6)).rjust(8),'ms', r['Percentage'])
PY
