#!/bin/bash
# rocprofv3 kernel stats of tools/bench_chain_ab.py (chain launches vs per-convolution launches in one process)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_chain
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/t -o t --output-format csv -- python $R/tools/bench_chain_ab.py > $OUT/log.txt 2>&1
cp $(find $OUT/t -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
grep mode $OUT/log.txt
head -8 $OUT/kernel_stats.csv | cut -c1-160
