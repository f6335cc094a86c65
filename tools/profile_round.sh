#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel-trace stats of the bench command, then separate PMC
# passes (HBM traffic; MFMA/LDS activity).  Outputs go to gpurun_out/prof_$1/ (scratch); the
# summaries worth judging are copied into profiles/ afterwards from the build container.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# per-kernel numbers are taken with the RRDB trunk on one stream (no overlapping launches), as in
# bench.py's profiled pass; trace2 is the default two-chain run of the same command
export NEOSR_AMD_STREAMS=1
BENCH="python $R/bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-other-configs --no-roofline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-budget 0 --no-other-configs > $OUT/trace.log 2>&1
NEOSR_AMD_STREAMS=2 rocprofv3 --kernel-trace --stats -d $OUT/trace2 -o trace --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-budget 0 --no-other-configs > $OUT/trace2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- $BENCH > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq -o sq --output-format csv -- $BENCH > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/lds -o lds --output-format csv -- $BENCH > $OUT/lds.log 2>&1
tail -2 $OUT/trace.log
python - <<PY
import csv, glob, collections, json
out = {}
def agg(tag):
    res = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            res[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key); n[k] += 1
    return res, n
summary = {}
for tag in ("fetch", "write", "sq", "lds"):
    res, n = agg(tag)
    for k, d in res.items():
        if "conv3x3" not in k and "adamw" not in k and "l1_" not in k and "sumsq" not in k: continue
        s = summary.setdefault(k[:80], {})
        s["dispatches_" + tag] = n[k]
        for c, v in d.items():
            s[c] = v
            s[c + "_per_dispatch"] = v / max(1, n[k])
json.dump(summary, open("$OUT/pmc_summary.json", "w"), indent=1)
for k, v in summary.items():
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a.endswith("_per_dispatch") or a.startswith("dispatches")})
PY
ls $OUT/trace
