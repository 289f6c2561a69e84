#!/bin/bash
# builds experiment variants of sweep_gemm_bf16.hip (extra -D flags) as separate libraries under tools/probes/out/
# usage: pp_variants.sh name "-DFLAG=..." [name flags ...]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/tools/probes/out
mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -w"
OBJS=$(ls $R/velesdb_amd/lib/obj/*.o | grep -v sweep_gemm_bf16.o)
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc $FL $f -c $R/velesdb_amd/csrc/sweep_gemm_bf16.hip -o $O/g16_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libvelesdb_hip_$n.so $OBJS $O/g16_$n.o
  echo built $n
done
