#!/bin/bash
# tools/ubench/run.sh -- on the GPU box: run the two ceiling microbenchmarks (built beforehand with hipcc) and
# a kernel trace of each; results land in gpurun_out/ubench/ and are copied into profiles/ by hand.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ubench
mkdir -p $OUT
timeout -k 10 240 tools/ubench/valu_issue > $OUT/valu_issue.jsonl 2> $OUT/valu_issue.err || { echo "valu_issue failed"; tail -3 $OUT/valu_issue.err; exit 1; }
echo "valu_issue done: $(wc -l < $OUT/valu_issue.jsonl) lines"
timeout -k 10 240 tools/ubench/l1_gather > $OUT/l1_gather.jsonl 2> $OUT/l1_gather.err || { echo "l1_gather failed"; tail -3 $OUT/l1_gather.err; exit 1; }
echo "l1_gather done: $(wc -l < $OUT/l1_gather.jsonl) lines"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_valu -- tools/ubench/valu_issue > /dev/null 2> $OUT/kt_valu.err || { echo "kt valu failed"; exit 1; }
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_l1 -- tools/ubench/l1_gather > /dev/null 2> $OUT/kt_l1.err || { echo "kt l1 failed"; exit 1; }
echo ok
