#!/bin/bash
# effective clock + matrix-pipe busy of the ring GEMM's ablation variants (diagnostic build), one PMC pass each
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03/pmc_clock
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in ${DBGS:-0 1 7 4}; do
  RC_GEMM_RING_DBG=$d REPS=6 timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/dbg$d -- python3 $R/tools/gemm_sweep.py > $O/dbg$d.log 2>&1
  echo "dbg$d exit=$?"
done
cd $R && python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r03/pmc_clock/dbg*/')):
    cc = glob.glob(d + '*/*counter_collection.csv'); kt = glob.glob(d + '*/*kernel_trace.csv')
    if not cc or not kt: print(d, 'no csv'); continue
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        if 'k_gemm_f64r' in r['Kernel_Name'] and '136' in r['Kernel_Name']:
            dur[r['Kernel_Name'][:48]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if 'k_gemm_f64r' in r['Kernel_Name'] and '136' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for kn, v in dur.items():
        us = sorted(v)[len(v)//2]
        g = agg.get('GRBM_GUI_ACTIVE', [0]); mf = agg.get('SQ_VALU_MFMA_BUSY_CYCLES', [0])
        gm, mm = sorted(g)[len(g)//2], sorted(mf)[len(mf)//2]
        print(f"{d.split('/')[-2]:6s} {kn}  median {us:7.1f} us  GRBM_GUI_ACTIVE {gm:.4g} -> {gm/8/us/1e3:.3f} GHz   MFMA_BUSY {mm:.4g} -> {mm/1024/(gm/8):.3f} of SIMD-cycles")
PY
