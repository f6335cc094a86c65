#!/bin/bash
# rocprofv3 --kernel-trace of a bench config, aggregated by (kernel, grid, workgroup): which launch SHAPES the time goes to
# (kernel_stats.csv only has one row per symbol).  usage: tools/trace_by_grid.sh CONFIG [steps]  -> gpurun_out/bygrid_CONFIG.txt
CFG=$1; STEPS=${2:-3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/bygrid_$CFG
mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/t -o t --output-format csv -- python $R/bench.py --config $CFG --cpu-budget 0 --no-other-configs --no-roofline --steps $STEPS --warmup 2 > $OUT/log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
tmin = None
rows = []
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(r)
# keep the last STEPS/(STEPS+2) of the launches by start time as "steady state": simpler — use everything, report per step
for r in rows:
    k = (r["Kernel_Name"][:70], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))), r["Workgroup_Size_X"])
    a = agg[k]; a[0] += 1; a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
with open("$R/gpurun_out/bygrid_$CFG.txt", "w") as fo:
    fo.write("total kernel ms %.2f over %d launches (whole process: warm-up + %d steps)\n" % (tot / 1e6, len(rows), $STEPS))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        fo.write("%-70s wgs %7d x %4s  n %6d  avg %8.1f us  total %8.2f ms  %5.1f%%\n" % (k[0], k[1], k[2], a[0], a[1] / a[0] / 1e3, a[1] / 1e6, 100 * a[1] / tot))
PY
rm -rf $OUT/t
