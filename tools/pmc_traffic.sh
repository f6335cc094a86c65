#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace + pmc only) of one bench config.
# usage: tools/pmc_traffic.sh TAG [bench args...]   -> gpurun_out/pmc_TAG/summary.json
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export NEOSR_AMD_STREAMS=1
BENCH="python $R/bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-other-configs --no-roofline $@"
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
        s = summary.setdefault(k[:120], {})
        s["dispatches_" + tag] = n[k]
        for c, v in d.items():
            s[c + "_per_dispatch"] = v / max(1, n[k])
json.dump(summary, open("$OUT/summary.json", "w"), indent=1)
rows = sorted(summary.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_per_dispatch", 0) * kv[1].get("dispatches_fetch", 0))
for k, v in rows[:14]:
    # FETCH_SIZE counts 128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md HBM section): x2; both in KiB
    print("%-100s n=%5d rd=%8.1f MB wr=%8.1f MB" % (k[:100], v.get("dispatches_fetch", 0), 2 * v.get("FETCH_SIZE_per_dispatch", 0) / 1024, v.get("WRITE_SIZE_per_dispatch", 0) / 1024))
PY
