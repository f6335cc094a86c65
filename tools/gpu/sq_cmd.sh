# SQ counters of an arbitrary command on the GPU box (two rocprofv3 --pmc passes, kernel-trace + pmc only):
#   gpurun -- 'bash tools/gpu/sq_cmd.sh TAG "python tools/bench_gemm.py 8"'  ->  gpurun_out/sq_TAG.txt (per-kernel averages)
TAG=$1; CMD=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/sq1 -o sq1 --output-format csv -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU -d $OUT/sq2 -o sq2 --output-format csv -- $CMD > $OUT/sq2.log 2>&1
python - <<PY > $R/gpurun_out/sq_$TAG.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
dur = collections.defaultdict(float)
for tag in ("sq1", "sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key); n[k][tag] += 1
                if tag == "sq1": dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
for k, d in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:12]:
    disp = max(1, n[k]["sq1"]); disp2 = max(1, n[k]["sq2"])
    print(k, "dispatches", disp, "avg_us", round(dur[k] / disp / 1e3, 2))
    for c, v in sorted(d.items()):
        per = v / (disp if c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16") else disp2)
        print("    %-34s %14.1f" % (c, per))
PY
rm -rf $OUT/sq1 $OUT/sq2
tail -3 $OUT/sq1.log
