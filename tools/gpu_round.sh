#!/bin/bash
# One GPU-box session: parity tests, developer checks, bench variants, rocprof kernel trace.
# Every step writes under gpurun_out/ ; steps are joined so a hang stops the chain.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee gpurun_out/progress.log
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout=300 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit=$?" | tee -a gpurun_out/progress.log
tail -5 gpurun_out/pytest_gpu.log
echo "== dev checks" | tee -a gpurun_out/progress.log
timeout -k 10 400 python tools/dev_gpu_check.py id sampling big > gpurun_out/dev2.log 2>&1
echo "dev exit=$?" | tee -a gpurun_out/progress.log
echo "== bench streams=1 eager" | tee -a gpurun_out/progress.log
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-graph --no-cpu-baseline > gpurun_out/bench_s1_eager.json 2> gpurun_out/bench_s1_eager.err
echo "exit=$?" | tee -a gpurun_out/progress.log
echo "== bench streams=1 graph" | tee -a gpurun_out/progress.log
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline > gpurun_out/bench_s1_graph.json 2> gpurun_out/bench_s1_graph.err
echo "exit=$?" | tee -a gpurun_out/progress.log
echo "== bench default (streams=4 graph, cpu baseline)" | tee -a gpurun_out/progress.log
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "exit=$?" | tee -a gpurun_out/progress.log
cat gpurun_out/bench_default.json
echo "== rocprofv3 kernel trace of bench" | tee -a gpurun_out/progress.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
echo "rocprof exit=$?" | tee -a $GRAFT_REPO_ROOT/gpurun_out/progress.log
cd $GRAFT_REPO_ROOT && find gpurun_out/prof_bench -name "*stats*" | head
