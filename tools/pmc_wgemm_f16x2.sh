#!/bin/bash
# counters of the f16x2 batched GEMM alone on one shape (Mt = 8 x tiles per utterance): matrix-pipe busy, wave-cycle split, LDS, HBM bytes -> gpurun_out/pmcW2_<Mt>_<N>_<K>.json
# usage: tools/pmc_wgemm_f16x2.sh Mt N K      (separate passes, --kernel-trace only)
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
P="python $R/tools/wgemm_one.py $1 $2 $3 64 3 f16x2"
rm -rf $OUT/pw_*
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pw_m -o m -- $P > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pw_f -o f -- $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pw_w -o w -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pw_l -o l -- $P > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, json, collections
def load(d):
    f = glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
tot = collections.defaultdict(float); n = collections.Counter(); dur = 0.0; seen = set()
for d in ("pw_m", "pw_f", "pw_w", "pw_l"):
    for r in load(d):
        if "wgemm_f16x2_" not in r["Kernel_Name"]: continue
        key = r["Counter_Name"] + ("@l" if d == "pw_l" and r["Counter_Name"] == "SQ_WAVE_CYCLES" else "")
        tot[key] += float(r["Counter_Value"]); n[key] += 1
        if d == "pw_m" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
L = max(n["SQ_WAVE_CYCLES"], 1); gui = tot["GRBM_GUI_ACTIVE"] / 8.0
out = {"shape": "$1 x $2 x $3 x 64 positions", "launches": L, "avg_launch_ms_under_pmc": dur / L * 1e-6,
       "mfma_utilisation": tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0) if gui else None, "effective_clock_GHz": gui / dur if dur else None,
       "wave_cycle_split": {k: tot[k] / tot["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if tot.get("SQ_WAVE_CYCLES")},
       "lds_issue_stall_share_of_wave_cycles": tot["SQ_WAIT_INST_LDS"] / tot["SQ_WAVE_CYCLES@l"] if tot.get("SQ_WAVE_CYCLES@l") else None,
       "lds_bank_conflict_over_active": tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"] if tot.get("SQ_LDS_IDX_ACTIVE") else None,
       "fetch_GB_per_launch_x2_corrected": 2.0 * tot["FETCH_SIZE"] * 1024 / max(n["FETCH_SIZE"], 1) / 1e9, "write_GB_per_launch": tot["WRITE_SIZE"] * 1024 / max(n["WRITE_SIZE"], 1) / 1e9,
       "algorithmic_GB": 4.0 * 64 * $1 * ($2 + $3) / 1e9}
json.dump(out, open("$OUT/pmcW2_$1_$2_$3.json", "w"), indent=1); print(json.dumps(out))
PY
rm -rf $OUT/pw_*
