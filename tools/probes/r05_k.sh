export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
timeout 900 python -m pytest tests/test_gpu_storage_modes.py -x -q > gpurun_out/r05k/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05k/pytest.log
timeout 300 python tools/fuzz_storage.py --seconds 90 --seed 531 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python tools/fuzz_storage.py --select --seconds 60 --seed 532 2>&1 | grep -v amdgpu.ids | tail -1
for LIBV in "" tools/probes/out/libvelesdb_hip_oldsq8.so; do
echo "== library: ${LIBV:-product}"
LIBV=$LIBV python - <<'PY'
import sys, time, os
sys.path.insert(0, ".")
if os.environ.get('LIBV'):
    from velesdb_amd import _ffi
    _ffi.use_library(os.path.abspath(os.environ['LIBV']))
import torch, velesdb_amd as va
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(42)
N, D, K = 1_000_000, 768, 10
st = torch.cuda.current_stream().cuda_stream
for mname, metric in (("cosine", va.DistanceMetric.Cosine), ("euclidean", va.DistanceMetric.Euclidean)):
    corpus = torch.randn((N, D), generator=g, device=dev)
    queries = torch.randn((64, D), generator=g, device=dev)
    ix = va.HnswIndex(D, metric, va.HnswParams(16, 100, N))
    torch.cuda.synchronize()
    ix.upload_dev(0, corpus.data_ptr(), N, st); torch.cuda.synchronize(); del corpus
    ix.set_storage_mode(va.StorageMode.SQ8)
    va.set_split_selector(0)     # the exact code sweep for every batch size
    ids = torch.empty((64, K), dtype=torch.int64, device=dev); sc = torch.empty((64, K), dtype=torch.float32, device=dev); n = torch.empty((64,), dtype=torch.int32, device=dev)
    for nq in (1, 4, 8, 16, 64):
        for _ in range(2):
            ix.search_batch_dev(queries.data_ptr(), nq, K, 0, va.MODE_BRUTE_SQ8, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ix.search_batch_dev(queries.data_ptr(), nq, K, 0, va.MODE_BRUTE_SQ8, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{mname} exact sq8 sweep nq={nq:3d}: {dt*1e3:7.3f} ms per call = {dt*1e3/nq:6.3f} ms per query", flush=True)
    va.set_split_selector(-1)
    ix.close(); torch.cuda.empty_cache()
PY
done
