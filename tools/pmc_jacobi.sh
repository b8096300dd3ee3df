#!/bin/bash
# PMC passes for k_jacobi_lds (one counter group per pass, kernel-trace only)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcj
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcj/$tag -- python3 $R/tools/jacobi_only.py > $R/gpurun_out/pmcj/$tag.log 2>&1
  echo "$tag exit=$?"
done
cd $R && python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmcj/*/*/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        kn = r.get('Kernel_Name', '')
        if 'k_jacobi_lds' in kn:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for cn, v in sorted(agg.items()):
        print(f"{cn:28s} launches={len(v)} mean per launch={sum(v)/len(v):.5g}")
PY
