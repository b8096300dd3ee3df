#!/bin/bash
# round 3, cfg5: parity of the column-ID paths, single-matrix timing, batch bench
set -o pipefail
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=700 -k "cfg5 or batch or column_id or ids_match or qr_compression or blocked" > $O/pytest_cfg5.log 2>&1
rc=$?; tail -4 $O/pytest_cfg5.log; [ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/prof_cfg5.py > $O/prof_cfg5.txt 2>&1; grep "one call" $O/prof_cfg5.txt; sort -k2 -n -r -t$'\t' $O/prof_cfg5.txt | grep -v "one call" | awk '{print}' | sort -t' ' -k1,1 | head -0
python - <<'PY'
rows=[]
for ln in open('gpurun_out/r03/prof_cfg5.txt'):
    if ' ms  x' in ln:
        name, rest = ln.rsplit(' ms  x', 1)[0].rsplit(None, 1), ln.rsplit(' ms  x', 1)[1]
        rows.append((float(name[1]), int(rest), name[0].strip()))
tot=sum(r[0] for r in rows if r[2].startswith('kernel:') or True)
for ms, calls, nm in sorted(rows, reverse=True)[:22]: print(f"{ms:8.3f} ms x{calls:<3d} {nm}")
PY
for v in ${VARIANTS:-0 1}; do
  RC_COLUMN_ID_FORM_Q=$v timeout -k 10 300 python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_formq$v.json 2> $O/bench_cfg5_formq$v.err || { echo "cfg5 bench failed"; tail -5 $O/bench_cfg5_formq$v.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_cfg5_formq$v.json')); print('FORM_Q=$v', d['value'], 'matrices/s', d['ms_per_step'], 'ms/batch', d['timed_results_check']['failed'], d['timed_results_check']['last_timed_batch_equals_first_bitwise'])"
done
