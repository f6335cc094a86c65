#!/bin/bash
# SQ counters of the Winograd conv micro-benchmark (GPU box): tools/pmc_wino.sh "COUNTER ..." [lib.so]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$2" ] && export NEOSR_AMD_LIB=$R/$2
rm -rf /tmp/pmc_wino
timeout 200 rocprofv3 --kernel-trace --pmc $1 -d /tmp/pmc_wino -o wino --output-format csv -- python $R/tools/bench_conv.py > /tmp/pmc_wino.log 2>&1
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_wino/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "wino" in row["Kernel_Name"]:
            res[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in res.items():
    print(k)
    for c, v in d.items():
        v = sorted(v)
        print(f"   {c:32s} n={len(v)} max={v[-1]:.4g} median={v[len(v)//2]:.4g}")
if not res:
    print(open("/tmp/pmc_wino.log").read()[-1500:])
PY
