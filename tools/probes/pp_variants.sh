#!/bin/bash
# builds ablation variants of the ping-pong kernel (VDB_PP_EXP bits) as separate libraries under tools/probes/out/
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/tools/probes/out
mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -w"
OBJS=$(ls $R/velesdb_amd/lib/obj/*.o | grep -v sweep_gemm_bf16.o)
for e in "$@"; do
  /opt/rocm/bin/hipcc $FL -DVDB_PP_EXP=$e -c $R/velesdb_amd/csrc/sweep_gemm_bf16.hip -o $O/g16_exp$e.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libvelesdb_hip_exp$e.so $OBJS $O/g16_exp$e.o
  echo built exp$e
done
