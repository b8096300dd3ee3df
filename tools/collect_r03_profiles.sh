#!/bin/bash
# copies the summaries of tools/gpu_round3_profiles.sh (gpurun_out/r03p/, scratch) into profiles/ (tracked) under round-3 names
set -e
S=gpurun_out/r03p
cp $S/bench_driver.json profiles/r03_bench_driver_style.json
cp $S/kernel_stats_s1.csv profiles/r03_bench_streams1_kernel_stats.csv
cp $S/timeline_s1.txt profiles/r03_timeline_streams1.txt
cp $S/kernel_stats_s40.csv profiles/r03_bench_streams40_kernel_stats.csv
cp $S/timeline_s40.txt profiles/r03_timeline_streams40.txt
cp $S/in_flight.json profiles/r03_in_flight.json
grep -v amdgpu.ids $S/ablation.txt > profiles/r03_ablation_untraced.txt
cp $S/lane_sweep.txt profiles/r03_lane_sweep.txt
cp $S/pmc_gemm_summary.txt profiles/r03_pmc_gemm.txt
cp $S/pmc_traffic.json profiles/r03_pmc_traffic.json
cp $S/cfg5_kernel_stats.csv profiles/r03_cfg5_blocked_kernel_stats.csv
cp $S/pmc_cfg5_summary.txt profiles/r03_pmc_cfg5_blocked_qrcp.txt
cp $S/qrblk_bench.json profiles/r03_qrblk_bench.json
cp $S/bench_cfg5.json profiles/r03_bench_cfg5.json
ls -la profiles/r03_*
