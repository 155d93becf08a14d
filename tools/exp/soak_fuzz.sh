#!/bin/bash
# tools/exp/soak_fuzz.sh [first_base [count]] -- on the GPU box (via gpurun): the randomised parity families of
# tests/test_gpu_fuzz.py with fresh seed bases, one pytest process per base, stopping at the first failure.
set -e
cd "$GRAFT_REPO_ROOT"
FIRST=${1:-8800000}; COUNT=${2:-12}
if [ "$FIRST" = 8800000 ]; then
  timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 100 --timeout-method thread > gpurun_out/noalloc_alone.log 2>&1
  echo "noalloc alone: $(tail -1 gpurun_out/noalloc_alone.log)"
fi
for k in $(seq 0 $((COUNT - 1))); do
  b=$((FIRST + 1000 * k))
  CRT_FUZZ_SEED_BASE=$b timeout -k 10 240 python -u -m pytest tests/test_gpu_fuzz.py -m gpu -q --timeout 120 --timeout-method thread > gpurun_out/soak_$b.log 2>&1 || { echo "base $b: FAILED"; tail -40 gpurun_out/soak_$b.log; exit 1; }
  echo "base $b: $(tail -1 gpurun_out/soak_$b.log)"
done
