#!/bin/bash
# SQ-counter summary of ONE bench config on the GPU box (matrix-pipe busy, vector / LDS activity per kernel): two separate
# rocprofv3 --pmc passes (kernel-trace + pmc only, 8 SQ slots each), trunk on one stream as in bench.py's profiled pass.
# usage: tools/profile_sq.sh ROUND CONFIG  ->  gpurun_out/prof_ROUND_CONFIG/sq_summary.json
# (scratch; copy into profiles/ROUND_CONFIG_sq_summary.json to have bench.py cite it.)
ROUND=$1; CFG=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${ROUND}_${CFG}
mkdir -p $OUT
export NEOSR_AMD_STREAMS=1 NEOSR_AMD_BLOCK_TAIL=0 NEOSR_AMD_D_OVERLAP=0   # (serial per-kernel durations: no side-by-side work)
BENCH="python $R/bench.py --config $CFG --cpu-budget 0 --no-other-configs --steps 2 --warmup 1 --no-roofline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/sq1 -o sq1 --output-format csv -- $BENCH > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU -d $OUT/sq2 -o sq2 --output-format csv -- $BENCH > $OUT/sq2.log 2>&1
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
dur = collections.defaultdict(float)
for tag in ("sq1", "sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key); n[k][tag] += 1
                if tag == "sq1": dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
kern = {}
total = sum(dur.values()) or 1.0
for k, d in agg.items():
    if dur[k] / total < 0.004: continue
    disp = max(1, n[k]["sq1"]); disp2 = max(1, n[k]["sq2"])
    per = {c: v / (disp if c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                 "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32") else disp2) for c, v in d.items()}
    busy = per.get("SQ_BUSY_CYCLES", 0.0)
    e = {"dispatches": disp, "avg_us_under_pmc": round(dur[k] / disp / 1e3, 2), "share_of_kernel_time": round(dur[k] / total, 4)}
    e.update({c + "_per_dispatch": round(v, 1) for c, v in sorted(per.items())})
    if busy > 0:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (cycles)
        e["mfma_busy_frac"] = round(per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * busy), 4)
        wc = per.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            e["wave_wait_any_frac"] = round(per.get("SQ_WAIT_ANY", 0.0) / wc, 4)
            e["wave_wait_inst_any_frac"] = round(per.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
    # one MOP = 512 FLOP of an fp32 MFMA (v_mfma_f32_32x32x2_f32 = 8 MOPs, 16x16x4 = 4): the counters' own count of the
    # multiplications the matrix pipe executed
    e["executed_mfma_gflop_per_dispatch"] = round(per.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512 / 1e9, 4)
    if per.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        e["lds_bank_conflict_frac_of_lds_cycles"] = round(per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"], 4)
    if per.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) > 0 and per.get("SQ_INSTS_VALU", 0) > 0:
        e["valu_insts_per_mfma_mop"] = round(per["SQ_INSTS_VALU"] / per["SQ_INSTS_VALU_MFMA_MOPS_F32"], 3)
    kern[k[:160]] = e
json.dump({"command": "NEOSR_AMD_STREAMS=1 NEOSR_AMD_BLOCK_TAIL=0 NEOSR_AMD_D_OVERLAP=0 $BENCH", "definitions": {
    "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): matrix-pipe busy cycles summed over the 1024 SIMDs against the kernel's busy cycles (summed over the 32 shader engines)",
    "executed_mfma_gflop_per_dispatch": "SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 FLOP"}, "kernels": kern},
    open("$OUT/sq_summary.json", "w"), indent=1)
for k, e in sorted(kern.items(), key=lambda kv: -kv[1]["share_of_kernel_time"])[:8]:
    print(k[:70], e.get("dispatches"), e.get("avg_us_under_pmc"), "mfma_busy", e.get("mfma_busy_frac"), "GF/disp", e.get("executed_mfma_gflop_per_dispatch"))
PY
rm -rf $OUT/sq1 $OUT/sq2
