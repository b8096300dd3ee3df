for v in 16 8; do RC_JACOBI_LPP=$v timeout -k 10 200 python bench.py --steps 256 --no-cpu-baseline 2>/dev/null > gpurun_out/ab.json; python - $v <<'PY'
import json,sys
d=json.load(open('gpurun_out/ab.json')); st=d['stage_ms_single_stream_eager']
print('lpp', sys.argv[1], d['value'], d['ms_per_step'], 'jacobi', st.get('op:jacobi_svd n=128'), 'sweeps', st.get('info:jacobi_sweeps n=128'))
PY
done
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
