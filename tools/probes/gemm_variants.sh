#!/bin/bash
# Builds kernel-variant copies of libvelesdb_hip.so (sweep_gemm.hip recompiled with -D flags, every other object
# taken from the product build) into tools/probes/out/, for A/B probes on the GPU:
#   tools/probes/gemm_variants.sh noepi=-DVDB_GEMM_ABL_NOEPI nomfma=-DVDB_GEMM_ABL_NOMFMA
#   VELESDB_HIP_LIB=tools/probes/out/libvelesdb_hip_noepi.so python tools/sweep_probe.py
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/tools/probes/out
mkdir -p $OUT
python -m velesdb_amd.build >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
SRC=${SRC:-sweep_gemm}
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  /opt/rocm/bin/hipcc $FLAGS ${defs//,/ } -c $ROOT/velesdb_amd/csrc/$SRC.hip -o $OUT/${SRC}_$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%=*}
  objs=$(ls $ROOT/velesdb_amd/lib/obj/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libvelesdb_hip_$name.so $objs $OUT/${SRC}_$name.o
  echo built $OUT/libvelesdb_hip_$name.so
done
