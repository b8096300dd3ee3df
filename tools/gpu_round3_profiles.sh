#!/bin/bash
# Round-3 profiling session on the GPU box: writes under gpurun_out/r03p/ ; tools/collect_r03_profiles.sh copies what is to be judged
# into profiles/.  EVERY step's failure ends the script with a non-zero status (a failing step was reported as rc=0).
#   1. driver-style bench line
#   2. rocprofv3 kernel-trace stats of the bench with ONE stream (uncontended durations of the final kernels) + timeline
#   3. the same at the default 40 lanes (in-flight durations: the un-split GEMM launches the timed region runs) + timeline + in_flight.json
#   4. PMC passes on the two big products, lone launches (FETCH_SIZE; WRITE_SIZE; MFMA busy + clock), one group per pass
#   5. cfg5: kernel trace + FETCH_SIZE / WRITE_SIZE of rank-64 column IDs; cfg5 / cfg4 timings (tools/qrblk_bench.py); cfg5 bench line
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
rm -rf $O; mkdir -p $O
cd $R
die() { echo "FAILED: $1" | tee -a $O/progress.log; exit 1; }
step() { echo "== $1 $(date +%T)" | tee -a $O/progress.log; }
step "bench driver-style"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || die "bench"
for S in 1 40; do
  step "rocprof kernel trace, $S stream(s)"
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s$S -- python3 $R/bench.py --streams $S --steps 16 --warmup 4 --no-cpu-baseline --no-h2d --no-gemm-lanes > $O/rocprof_s$S.log 2>&1 ) || die "rocprof streams=$S"
  f=$(ls -t $O/prof_s$S/*/*kernel_trace.csv | head -1)
  python tools/timeline.py $f --frac 0.4 --top 45 > $O/timeline_s$S.txt || die "timeline $S"
  cp $(ls -t $O/prof_s$S/*/*kernel_stats.csv | head -1) $O/kernel_stats_s$S.csv
  if [ $S = 40 ]; then cp $f $O/kernel_trace_s40.csv; fi
  rm -rf $O/prof_s$S
done
python - <<'PY' || exit 1
# in-flight record of the two big products: launches of the un-split grid (32 workgroups) in the 40-lane trace
import csv, json, collections, re
O = 'gpurun_out/r03p'
rows = [r for r in csv.DictReader(open(O + '/kernel_trace_s40.csv')) if 'k_gemm_f64' in r['Kernel_Name']]
t0, t1 = min(int(r['Start_Timestamp']) for r in rows), max(int(r['End_Timestamp']) for r in rows)
lo = t1 - 0.4 * (t1 - t0)
agg = collections.defaultdict(list)
for r in rows:
    if int(r['Start_Timestamp']) < lo: continue
    wgs = 1
    for ax in 'XYZ': wgs *= max(1, int(r[f'Grid_Size_{ax}']) // max(1, int(r[f'Workgroup_Size_{ax}'])))
    name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void rc::', ''))
    agg[(name, wgs)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = {}
for (name, wgs), v in agg.items():
    avg = sum(v) / len(v)
    out.setdefault(name.replace(' ', ''), []).append({"workgroups": wgs, "launches": len(v), "avg_us": round(avg, 2), "cu_time_us": round(avg * min(wgs, 256) / 256.0, 2)})
json.dump({"source": "rocprofv3 --kernel-trace of `bench.py --streams 40 --steps 16 --warmup 4` (last 40 % of the trace), tools/gpu_round3_profiles.sh; "
                     "cu_time_us = average duration x min(workgroups, 256) / 256; NOTE: tracing serialises part of the concurrency, durations are upper bounds",
           "kernels": out}, open(O + '/in_flight.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
rm -f $O/kernel_trace_s40.csv
step "un-traced ablation of the pipeline + lane sweep"
timeout -k 10 300 python tools/ablate_pipeline.py > $O/ablation.txt 2> $O/ablation.err || die "ablation"
cat $O/ablation.txt
for s in 24 32 40 44 48 64 68; do
  timeout -k 10 200 python bench.py --streams $s --steps 16 --warmup 4 --no-cpu-baseline --no-h2d --no-gemm-lanes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $s (24 hardware queues):', d['value'], 'compressions/s')" >> $O/lane_sweep.txt || die "lane sweep $s"
done
cat $O/lane_sweep.txt
step "PMC passes on the big products (lone launches)"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_gemm/$tag -- python3 $R/tools/gemm_sweep.py > $O/pmc_gemm_$tag.log 2>&1 ) || die "pmc $tag"
done
step "cfg5: kernel trace + PMC"
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $R/tools/prof_cfg5.py > $O/prof_cfg5.log 2>&1 ) || die "cfg5 trace"
cp $(ls -t $O/prof_cfg5/*/*kernel_stats.csv | head -1) $O/cfg5_kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_cfg5/$grp -- python3 $R/tools/prof_cfg5.py > $O/pmc_cfg5_$grp.log 2>&1 ) || die "cfg5 pmc $grp"
done
step "qrblk bench + cfg5 bench line"
timeout -k 10 300 python tools/qrblk_bench.py > $O/qrblk_bench.json 2> $O/qrblk_bench.err || die "qrblk_bench"
timeout -k 10 300 python bench.py --config cfg5 --steps 10 --warmup 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err || die "bench cfg5"
python - <<'PY' || exit 1
import csv, glob, collections, json
O = 'gpurun_out/r03p'
def agg(pattern, key):
    out = collections.defaultdict(list)
    for f in sorted(glob.glob(pattern)):
        for r in csv.DictReader(open(f)):
            kn = r.get('Kernel_Name', '')
            if key in kn:
                out[(kn.split('(')[0][:80], r['Counter_Name'])].append(float(r['Counter_Value']))
    return out
g = agg(O + '/pmc_gemm/*/*/*counter_collection.csv', 'k_gemm_f64')
dur = collections.defaultdict(list)
for f in sorted(glob.glob(O + '/pmc_gemm/SQ_*/*/*kernel_trace.csv')):
    for r in csv.DictReader(open(f)):
        if 'k_gemm_f64' in r['Kernel_Name']:
            dur[r['Kernel_Name'].split('(')[0][:80]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
lines = []
for (kn, cn), v in sorted(g.items()):
    lines.append(f"{cn:28s} n={len(v):3d} mean={sum(v)/len(v):.5g}  {kn}")
for kn, v in sorted(dur.items()):
    us = sorted(v)[len(v) // 2]
    gui = g.get((kn, 'GRBM_GUI_ACTIVE')); mf = g.get((kn, 'SQ_VALU_MFMA_BUSY_CYCLES'))
    if gui and mf:
        gm, mm = sum(gui) / len(gui), sum(mf) / len(mf)
        lines.append(f"derived: {kn}: median {us:.1f} us under the counters; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration = {gm/8/us/1e3:.3f} GHz; "
                     f"matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = {mm/1024/(gm/8):.3f}")
open(O + '/pmc_gemm_summary.txt', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
kern = {}
for (kn, cn), v in g.items():
    name = kn.replace('void rc::', '').strip()
    e = kern.setdefault(name, {})
    if cn == 'FETCH_SIZE': e['fetch_bytes'] = 2.0 * 1024.0 * sum(v) / len(v); e['launches'] = len(v)
    if cn == 'WRITE_SIZE': e['write_bytes'] = 1024.0 * sum(v) / len(v)
for e in kern.values():
    if 'fetch_bytes' in e and 'write_bytes' in e: e['bytes_per_launch'] = e['fetch_bytes'] + e['write_bytes']
json.dump({"source": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) with --kernel-trace over tools/gemm_sweep.py (lone launches, K split to cover the chip), means over the launches of each kernel; "
                     "FETCH_SIZE x2 gfx950 correction (tools/gpu_round3_profiles.sh)", "kernels": kern}, open(O + '/pmc_traffic.json', 'w'), indent=1)
c = collections.defaultdict(float)
for f in sorted(glob.glob(O + '/pmc_cfg5/*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        c[(r['Counter_Name'], r.get('Kernel_Name', '').split('(')[0][:60])] += float(r['Counter_Value'])
tot = collections.defaultdict(float)
rows = []
for (cn, kn), v in sorted(c.items(), key=lambda kv: -kv[1]):
    tot[cn] += v
    rows.append(f"{cn:12s} {v:14.4g} KiB  {kn}")
gen = sum(v for (cn, kn), v in c.items() if cn == 'WRITE_SIZE' and 'k_fill_gaussian' in kn)
calls = 5
per = {"fetch_GB_per_matrix": round(2.0 * 1024 * tot['FETCH_SIZE'] / calls / 1e9, 4), "write_GB_per_matrix": round(1024 * (tot['WRITE_SIZE'] - gen) / calls / 1e9, 4)}
open(O + '/pmc_cfg5_summary.txt', 'w').write("totals over the whole process (5 calls of rc_column_id_rank_f32 on 4096 x 4096, + the input generator, whose writes are subtracted below): "
                                            + json.dumps(tot) + "\nper matrix (FETCH_SIZE x2 gfx950 correction): " + json.dumps(per) + "\n" + "\n".join(rows[:40]) + "\n")
print(json.dumps(per))
PY
rm -rf $O/pmc_gemm $O/pmc_cfg5 $O/prof_cfg5
python -c "
import json; d=json.load(open('$O/bench_driver.json')); print('driver-style:', d['value'], 'c/s', d['ms_per_step'], 'ms/step', d['frac_of_f64_mfma_peak_whole_pipeline'], d['roofline']['achieved'], d['roofline']['frac'], d['cpu_baseline']['value'], d['value_including_h2d']['value_including_h2d'], d['roofline']['in_flight_live'])"
cat $O/qrblk_bench.json | head -12
step done
