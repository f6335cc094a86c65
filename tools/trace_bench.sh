#!/bin/bash
# kernel + memory-copy trace of a short bench run (no counters): gpurun_out/trace_bench/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_bench
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/t -o t --output-format csv -- python $R/bench.py --config ${1:-bench_esrgan} --cpu-budget 0 --no-other-configs --no-roofline --steps 3 --warmup 2 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-300
ls $OUT/t
