#!/bin/bash
# kernel-level durations of the TN GEMM micro-benchmark (GPU box): tools/prof_tn.sh [lib.so]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$1" ] && export NEOSR_AMD_LIB=$R/$1
rm -rf /tmp/prof_tn
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_tn -o tn --output-format csv -- python $R/tools/bench_tn.py > /tmp/prof_tn.log 2>&1
f=$(find /tmp/prof_tn -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then head -6 "$f" | cut -c1-160; else tail -5 /tmp/prof_tn.log; fi
