timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
grep -q failed gpurun_out/pytest_gpu.log && exit 1
for v in 0 1 0 1; do RC_CHOLQR_SKIP=$v timeout -k 10 200 python bench.py --steps 256 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); st=d['stage_ms_single_stream_eager']; print('skip $v:', d['value'], d['ms_per_step'], st.get('op:cholqr2 8192x133'), st.get('op:chol_inv n=133'))" || echo "failed"; done
