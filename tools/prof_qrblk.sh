#!/bin/bash
# kernel trace of one cfg5 rank-64 column ID (blocked path)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_qrblk
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_qrblk -- python3 $R/tools/prof_cfg5.py > $R/gpurun_out/prof_qrblk.log 2>&1
echo "rocprof exit=$?"
cd $R
tail -30 gpurun_out/prof_qrblk.log
python - <<'PY'
import csv,glob,os
f=sorted(glob.glob('gpurun_out/prof_qrblk/*/*kernel_stats.csv'), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
for r in rows[:24]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1000)).rjust(9),'us', r['Percentage'])
PY
