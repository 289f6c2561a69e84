#!/bin/bash
# Ablation builds of the bf16 LDS-DMA GEMM sweep for A/B probes on the GPU.  The product source carries no ablation
# hooks: every variant is a sed-patched COPY of csrc/sweep_gemm_bf16.hip compiled into tools/probes/out/ and linked with
# the product's other objects.  VELESDB_HIP_LIB=tools/probes/out/libvelesdb_hip_g16_<name>.so selects one.
#   noload   no global -> LDS traffic at all (multiply + barriers + epilogue on whatever the LDS holds)
#   nob      no query-tile loads           noa    no row-tile loads
#   l2rows   every row tile re-reads the same 8 tiles (L2-resident rows: isolates HBM from the load path)
#   noepi    no epilogue (quick test never fires, nothing appended)
# variants combine with '+': noload+noepi
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/tools/probes/out
mkdir -p $OUT
python -m velesdb_amd.build >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -I$ROOT/velesdb_amd/csrc"
SRC=$ROOT/velesdb_amd/csrc/sweep_gemm_bf16.hip
declare -A PATCH
PATCH[base]='s/XXXXNOPE//'
PATCH[noload]='s/^#define VDB_G16_GLDS(J) do { \\$/#define VDB_G16_GLDS(J) do { break; \\/'
PATCH[nob]='s/    else glds_b128(rb, /    else if (0) glds_b128(rb, /'
PATCH[noa]='s/    if ((J) < 4) glds_b128(ra, /    if (0) glds_b128(ra, /'
PATCH[l2rows]='s/make_rsrc(rows_b + ((size_t)ld_rt \* BM/make_rsrc(rows_b + ((size_t)(ld_rt \& 7u) * BM/'
PATCH[noepi]='s/        hm\[t\] = __ballot(hot);/        hm[t] = a.nq > 100000u ? __ballot(hot) : 0ull;/'
for name in "$@"; do
  cp $SRC $OUT/g16_$name.hip
  for part in ${name//+/ }; do sed -i -e "${PATCH[$part]}" $OUT/g16_$name.hip; done
  /opt/rocm/bin/hipcc $FLAGS -c $OUT/g16_$name.hip -o $OUT/g16_$name.o &
done
wait
for name in "$@"; do
  objs=$(ls $ROOT/velesdb_amd/lib/obj/*.o | grep -v "/sweep_gemm_bf16.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libvelesdb_hip_g16_$name.so $objs $OUT/g16_$name.o
  echo built $OUT/libvelesdb_hip_g16_$name.so
done
