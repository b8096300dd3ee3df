#!/bin/bash
# Round-2 profiling session on the GPU box: writes under gpurun_out/r02/ ; copy what is to be judged into profiles/.
#   1. bench.py default -> bench line
#   2. rocprofv3 kernel-trace stats of the driver's bench command
#   3. PMC passes (FETCH_SIZE, WRITE_SIZE; one group per pass, kernel-trace only) on the sketch GEMM -> traffic per launch
#   4. kernel-trace + PMC FETCH_SIZE / WRITE_SIZE of one cfg5 rank-64 column ID (blocked pivoted QR)
#   5. timings of the cfg5 / cfg4 workloads (tools/qrblk_bench.py)
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
echo "== bench default $(date +%T)" | tee $O/progress.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "bench failed"; tail -5 $O/bench_default.err; exit 1; }
echo "== rocprof kernel trace of the driver's bench command $(date +%T)" | tee -a $O/progress.log
rm -rf $O/prof_bench
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $O/rocprof_bench.log 2>&1
echo "rocprof exit=$?" | tee -a $O/progress.log
cd $R
echo "== PMC passes on the sketch GEMM $(date +%T)" | tee -a $O/progress.log
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  cd /tmp && timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_gemm/$tag -- python3 $R/tools/gemm_sweep.py > $O/pmc_gemm_$tag.log 2>&1
  echo "$tag exit=$?" | tee -a $O/progress.log
  cd $R
done
echo "== cfg5 blocked QRCP: kernel trace + PMC $(date +%T)" | tee -a $O/progress.log
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $R/tools/prof_cfg5.py > $O/prof_cfg5.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  cd /tmp && timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_cfg5/$grp -- python3 $R/tools/prof_cfg5.py > $O/pmc_cfg5_$grp.log 2>&1
  echo "cfg5 $grp exit=$?" | tee -a $O/progress.log
done
cd $R
echo "== qrblk bench $(date +%T)" | tee -a $O/progress.log
timeout -k 10 300 python tools/qrblk_bench.py > $O/qrblk_bench.json 2> $O/qrblk_bench.err || { echo "qrblk_bench failed"; tail -5 $O/qrblk_bench.err; exit 1; }
python - <<'PY'
import csv, glob, collections, json, os
O = 'gpurun_out/r02'
def agg(pattern, key):
    out = collections.defaultdict(list)
    for f in sorted(glob.glob(pattern)):
        for r in csv.DictReader(open(f)):
            kn = r.get('Kernel_Name', '')
            if key in kn:
                out[(kn.split('(')[0][:70], r['Counter_Name'])].append(float(r['Counter_Value']))
    return out
g = agg(O + '/pmc_gemm/*/*/*counter_collection.csv', 'k_gemm_f64')
lines = []
for (kn, cn), v in sorted(g.items()):
    lines.append(f"{cn:28s} n={len(v):3d} mean={sum(v)/len(v):.5g}  {kn}")
open(O + '/pmc_gemm_summary.txt', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
# per-kernel HBM traffic record read by bench.py (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE x 2 = the gfx950 correction for
# wide streaming reads of /opt/skills/guides/MI355X_MICROARCH.md, HBM section)
kern = {}
for (kn, cn), v in g.items():
    name = kn.replace('void rc::', '').strip()
    e = kern.setdefault(name, {})
    if cn == 'FETCH_SIZE': e['fetch_bytes'] = 2.0 * 1024.0 * sum(v) / len(v); e['launches'] = len(v)
    if cn == 'WRITE_SIZE': e['write_bytes'] = 1024.0 * sum(v) / len(v)
for e in kern.values():
    if 'fetch_bytes' in e and 'write_bytes' in e: e['bytes_per_launch'] = e['fetch_bytes'] + e['write_bytes']
json.dump({"source": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) with --kernel-trace over tools/gemm_sweep.py, means over the launches of each kernel; FETCH_SIZE x2 gfx950 correction (tools/gpu_round2_profiles.sh)",
           "kernels": kern}, open(O + '/pmc_traffic.json', 'w'), indent=1)
# cfg5: total bytes per call over ALL kernels of one rc_column_id_rank call (the script makes 4 calls + a profiled one)
c = collections.defaultdict(float)
for f in sorted(glob.glob(O + '/pmc_cfg5/*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        c[(r['Counter_Name'], r.get('Kernel_Name', '').split('(')[0][:60])] += float(r['Counter_Value'])
tot = collections.defaultdict(float)
rows = []
for (cn, kn), v in sorted(c.items(), key=lambda kv: -kv[1]):
    tot[cn] += v
    rows.append(f"{cn:12s} {v:14.4g} KiB  {kn}")
open(O + '/pmc_cfg5_summary.txt', 'w').write("totals over the whole process (5 calls of rc_column_id_rank_f32 on 4096 x 4096, + the input generator): "
                                            + json.dumps(tot) + "\n" + "\n".join(rows[:40]) + "\n")
print(json.dumps(tot))
PY
tail -3 $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default:', d['value'], 'c/s', d['ms_per_step'], 'ms/step', d['frac_of_f64_mfma_peak_whole_pipeline'], d['roofline']['achieved'], d['roofline']['frac'], d['cpu_baseline'], d['value_including_h2d'])"
cat $O/qrblk_bench.json
