#!/bin/bash
# LDS-array and vector-memory-path counters of the headline step's selection kernel (sweep_topk_gemm_bf16_pp): is the array as busy
# as DESIGN 4.1c's arithmetic says (~50-60 % at a full matrix pipe), and do the LDS-DMA requests queue in front of the texture
# addresser?  Separate --pmc passes of bench.py's headline child (kernel-trace only), each under its own short timeout.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-r04pmc}
mkdir -p $O
cd /tmp
run() {
  timeout ${T:-28} rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python $R/bench.py --pmc-child headline --steps 3 --warmup 1 > $O/$1.log 2>&1
  echo "$1 rc=$?"
  find $O/$1 -name "*_kernel_trace.csv" -size +4M -delete
}
# PASSES="name:COUNTER COUNTER ...;name:..." (default: the LDS / vector-memory passes)
PASSES=${PASSES:-"lds:SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES;vmem:SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES;lvl:SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES"}
IFS=';' read -ra PL <<< "$PASSES"
NAMES=""
for p in "${PL[@]}"; do
  run "${p%%:*}" "${p#*:}"
  NAMES="$NAMES ${p%%:*}"
done
export NAMES
python - <<PY > $O/summary.txt 2>&1
import csv, glob, collections
import os
for p in os.environ["NAMES"].split():
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % p, recursive=True)
    if not fs:
        print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, cs in sorted(acc.items(), key=lambda kv: -max(kv[1].values())):
        if "sweep_topk" not in k and "seed_scores" not in k and "split_rerank" not in k and "merge_topk" not in k:
            continue
        d = max(n[(k, c)] for c in cs)
        print(p, k[:60], "dispatches", d, " ".join(f"{c}={v / d:.4g}" for c, v in sorted(cs.items())))
    for r in csv.DictReader(open(glob.glob("$O/%s/**/*kernel_trace.csv" % p, recursive=True)[0])):
        if "gemm_bf16_pp" in r["Kernel_Name"]:
            acc["dur"][r["Kernel_Name"].split("(")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(p, "total us of the pp dispatches:", {k: round(v, 1) for k, v in acc["dur"].items()})
PY
cat $O/summary.txt | head -40
