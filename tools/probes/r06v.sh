# round 6: size of the WIDE selection's sample seed at small k
P=$GRAFT_REPO_ROOT/velesdb_amd/lib/libvelesdb_hip_probe.so
for R in 4096 2048 1024 3072 4096 2048; do
VELESDB_HIP_LIB=$P VELESDB_WIDE_SEED_ROWS=$R python tools/probes/wide_small_k_probe.py 10,20,32 2>&1 | grep ms_per | sed "s/^/seed_rows $R /" >> gpurun_out/r06v_seed_rows.log
done
cat gpurun_out/r06v_seed_rows.log
