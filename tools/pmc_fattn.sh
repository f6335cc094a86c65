#!/bin/bash
# SQ counters of the TN GEMM micro-benchmark (GPU box): tools/pmc_fa.sh "COUNTER ..." [lib.so]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$2" ] && export NEOSR_AMD_LIB=$R/$2
rm -rf /tmp/pmc_fa
timeout 200 rocprofv3 --kernel-trace --pmc $1 -d /tmp/pmc_fa -o tn --output-format csv -- python $R/tools/bench_fattn.py > /tmp/pmc_fa.log 2>&1
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_fa/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "flash_wattn" in row["Kernel_Name"] and "16, 16" in row["Kernel_Name"]:
            res[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in res.items():
    print(k)
    for c, v in d.items():
        v = sorted(v)
        print(f"   {c:32s} n={len(v)} max={v[-1]:.4g} median={v[len(v)//2]:.4g}")
if not res:
    print(open("/tmp/pmc_fa.log").read()[-1500:])
PY
