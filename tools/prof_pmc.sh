#!/bin/bash
# PMC passes for the conv kernels on the GPU box (separate runs, kernel-trace only; see guide §7)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
CMD="${PMC_CMD:-python $R/tools/bench_conv.py 16}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/p1 -o p1 --output-format csv -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU -d $OUT/p2 -o p2 --output-format csv -- $CMD > $OUT/p2.log 2>&1
ls -R $OUT | head -30
python - <<PY
import csv, glob, collections
for tag in ("p1","p2"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    print(tag, files)
    for f in files:
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        for k, d in agg.items():
            print(k, {c: round(v) for c, v in d.items()})
PY
