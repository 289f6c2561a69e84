# round 6: stamped blocks of wide_rerank_verify (probe build) at k = 10 and k = 100
P=$GRAFT_REPO_ROOT/velesdb_amd/lib/libvelesdb_hip_probe.so
for K in 10 50 100; do
VELESDB_HIP_LIB=$P VELESDB_WIDE_STAMPS=1 python tools/probes/wide_small_k_probe.py $K 2>&1 | grep "wide stamps" | tail -3 | sed "s/^/k=$K /" >> gpurun_out/r06u_stamps.log
done
cat gpurun_out/r06u_stamps.log
