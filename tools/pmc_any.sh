#!/bin/bash
# SQ counters of one kernel family in any micro-benchmark (GPU box): tools/pmc_any.sh "COUNTER ..." FILTER SCRIPT [args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=$1; F=$2; shift 2
rm -rf /tmp/pmc_any
timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_any -o p --output-format csv -- python $R/$@ > /tmp/pmc_any.log 2>&1
FILTER=$F python - <<'PY'
import csv, glob, collections, os
flt = os.environ["FILTER"]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_any/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if flt in row["Kernel_Name"]:
            res[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in res.items():
    print(k)
    for c, v in d.items():
        v = sorted(v)
        print(f"   {c:32s} n={len(v)} max={v[-1]:.4g} median={v[len(v)//2]:.4g}")
if not res:
    print(open("/tmp/pmc_any.log").read()[-1500:])
PY
