set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 100 --timeout-method thread > gpurun_out/noalloc_alone.log 2>&1
echo "noalloc alone: $(tail -1 gpurun_out/noalloc_alone.log)"
for b in 8800000 8801000 8802000 8803000 8804000 8805000 8806000 8807000 8808000 8809000 8810000 8811000; do
  CRT_FUZZ_SEED_BASE=$b timeout -k 10 240 python -u -m pytest tests/test_gpu_fuzz.py -m gpu -q --timeout 120 --timeout-method thread > gpurun_out/soak_$b.log 2>&1 || { echo "base $b: FAILED"; tail -40 gpurun_out/soak_$b.log; exit 1; }
  echo "base $b: $(tail -1 gpurun_out/soak_$b.log)"
done
