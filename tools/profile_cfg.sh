#!/bin/bash
# Round profile of ONE bench config on the GPU box: rocprofv3 --kernel-trace --stats of the bench command (trunk on one
# stream, as in bench.py's profiled pass; and the default launch chains), then FETCH_SIZE / WRITE_SIZE in separate
# --pmc passes (kernel-trace + pmc only).  usage: tools/profile_cfg.sh ROUND CONFIG
#   -> gpurun_out/prof_ROUND_CONFIG/{kernel_stats.csv, kernel_stats_chains.csv, pmc_summary.json}
# (scratch; copy what should be judged into profiles/ROUND_CONFIG_*.)
ROUND=$1; CFG=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${ROUND}_${CFG}
mkdir -p $OUT
ARGS="--config $CFG --cpu-budget 0 --no-other-configs"
NEOSR_AMD_STREAMS=1 NEOSR_AMD_BLOCK_TAIL=0 NEOSR_AMD_D_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python $R/bench.py $ARGS --steps 5 --warmup 2 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace2 -o trace --output-format csv -- python $R/bench.py $ARGS --steps 5 --warmup 2 > $OUT/trace2.log 2>&1
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/trace2 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_chains.csv
export NEOSR_AMD_STREAMS=1 NEOSR_AMD_BLOCK_TAIL=0 NEOSR_AMD_D_OVERLAP=0   # (serial per-kernel durations: no side-by-side work)
BENCH="python $R/bench.py $ARGS --steps 2 --warmup 1 --no-roofline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- $BENCH > $OUT/write.log 2>&1
python - <<PY
import csv, glob, collections, json
summary = {}
for tag in ("fetch", "write"):
    res = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            res[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key); n[k] += 1
    for k, d in res.items():
        if n[k] * max(d.values()) < 1e5: continue   # drop the init-time noise
        s = summary.setdefault(k[:160], {})
        s["dispatches_" + tag] = n[k]
        for c, v in d.items():
            s[c] = v
            s[c + "_per_dispatch"] = v / max(1, n[k])
json.dump(summary, open("$OUT/pmc_summary.json", "w"), indent=1)
PY
rm -rf $OUT/trace $OUT/trace2 $OUT/fetch $OUT/write
tail -1 $OUT/trace.log | cut -c1-300
head -8 $OUT/kernel_stats.csv | cut -c1-160
