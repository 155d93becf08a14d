#!/bin/bash
# tools/ubench/pmc_calib.sh -- on the GPU box: what do the SQ VALU counters report for instruction streams of known
# length and class?  (full-rate v_fma_f32, half-rate v_perm_b32, the k_trace8 node-step mix, the slow VOP2 v_cndmask)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ubench/pmc
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
i=0
for cls in "v_fma_f32" "v_perm_b32" "node step" "vcc written by s_mov"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $OUT/a$i -- tools/ubench/valu_issue "$cls" > $OUT/a$i.jsonl 2> $OUT/a$i.err
  echo "a$i [$cls] rc=$?"
  timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b$i -- tools/ubench/valu_issue "$cls" > $OUT/b$i.jsonl 2> $OUT/b$i.err
  echo "b$i [$cls] rc=$?"
done
