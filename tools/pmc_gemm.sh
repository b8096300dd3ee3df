#!/bin/bash
# PMC passes for the dominant kernel (sketch GEMM), one counter group per pass (TCC: FETCH_SIZE costs 3 of 4 slots).
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$tag -- python3 $R/tools/gemm_sweep.py > $R/gpurun_out/pmc/$tag.log 2>&1
  echo "$tag exit=$?"
done
cd $R && python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*/*/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        kn = r.get('Kernel_Name', '')
        if 'k_gemm_f64q' in kn:
            agg[(kn[:60], r['Counter_Name'], r.get('Grid_Size'))].append(float(r['Counter_Value']))
    for (kn, cn, gs), v in sorted(agg.items()):
        print(f"{cn:32s} grid={gs} n={len(v)} mean={sum(v)/len(v):.4g}  {kn}")
PY
