#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes
# of bench.py, each counter set in a run of its own (rocprofv3 does not split counters over passes).
# Raw output goes to gpurun_out/prof_<tag>/; tools/pmc_summary.py condenses it into profiles/.
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cp cudaraytracer_amd/isa_mix_k_trace8.json $OUT/isa_mix.json     # which k_trace8 these passes profile (its kernel_hash)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_kt.json 2> $OUT/kt.err || { echo "kernel trace failed"; tail -5 $OUT/kt.err; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline "$@" > $OUT/bench_pmc$i.json 2> $OUT/pmc$i.err
  echo "pmc$i [$set] rc=$?"
done
