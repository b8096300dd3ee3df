#!/bin/bash
# Builds diagnostic libraries with parts of the Jacobi round removed (RC_JAC_ABL bit mask, kernels_svd.hip) into tools/_dbg/.
# Run tools/jacobi_ablation.py on the GPU box afterwards.  The results of these builds are wrong by construction; only the time counts.
set -e
cd "$(dirname "$0")/../rusty_compression_amd/csrc"
make -s
mkdir -p ../../tools/_dbg
for abl in 32 33 34 36 40 48 63; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRC_JAC_ABL=$abl -c kernels_svd.hip -o /tmp/svd_abl_$abl.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_dbg/librc_jac_abl_$abl.so /tmp/svd_abl_$abl.o $(ls _build/*.o | grep -v kernels_svd)
  echo built $abl
done
