#!/bin/bash
# dev experiment: what k_shade waits for -- two extra PMC passes of bench.py (memory pipeline, opcode classes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_shade; mkdir -p $OUT
i=0
for set in "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > $OUT/bench$i.json 2> $OUT/pmc$i.err
  echo "pmc$i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/prof_shade/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "k_shade" in k or "k_resolve" in k or "k_trace8<false" in k.replace("ILb0", "<false"):
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
