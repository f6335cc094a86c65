#!/bin/bash
# Do kernels of different streams overlap in time?  rocprofv3 --kernel-trace of a bench config; per queue: launches, busy time;
# and the total time during which two or more kernels were running.  usage: tools/trace_overlap.sh CONFIG [ENV=VAL ...]
CFG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/overlap_$CFG
mkdir -p $OUT
env "$@" rocprofv3 --kernel-trace -d $OUT/t -o t --output-format csv -- python $R/bench.py --config $CFG --cpu-budget 0 --no-other-configs --no-roofline --steps 2 --warmup 2 > $OUT/log 2>&1
python - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
q = collections.defaultdict(lambda: [0, 0.0])
ev = []
for r in rows:
    a, b = float(r["Start_Timestamp"]), float(r["End_Timestamp"])
    k = (r.get("Queue_Id"), r.get("Stream_Id"))
    q[k][0] += 1; q[k][1] += b - a
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
depth, last, t = 0, None, collections.defaultdict(float)
for ts, d in ev:
    if last is not None: t[depth] += ts - last
    depth += d; last = ts
for k, v in sorted(q.items(), key=lambda kv: -kv[1][1]): print("queue/stream", k, "launches", v[0], "busy ms %.1f" % (v[1] / 1e6))
for d in sorted(t): print("kernels in flight", d, "ms %.1f" % (t[d] / 1e6))
PY
rm -rf $OUT/t
