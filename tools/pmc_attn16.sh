#!/bin/bash
# MFMA-utilisation / HBM-traffic PMC passes of the 16-bit attention kernels (tools/probes/attn16_probe_base, B = 4 x T = 15 008, C = 256, f16)
# -> gpurun_out/<tag>_attn16_pmc.json.  Separate passes, --kernel-trace only (MI355X_MICROARCH.md rocprofv3 section).
TAG=${1:-r05}
# the probe binary is not tracked: built here from tools/probes/attn16_probe.hip + the product's attn16.hip when missing (hipcc is on the box)
if [ ! -x tools/probes/attn16_probe_base ]; then
  F="--offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -Ibuddy_amd/csrc -Iinclude"
  hipcc $F -c tools/probes/attn16_probe.hip -o /tmp/attn16_probe.o && hipcc $F -c buddy_amd/csrc/attn16.hip -o /tmp/attn16.o && hipcc --offload-arch=gfx950 /tmp/attn16_probe.o /tmp/attn16.o -o tools/probes/attn16_probe_base || exit 1
fi
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
P="$R/tools/probes/attn16_probe_base time 4 15008 256 2 3"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pa_m -o m -- $P > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pa_f -o f -- $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pa_w -o w -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pa_l -o l -- $P > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, json, collections
def load(d):
    f = glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
out = {}
for key in ("fa16_fwd", "fa16_dq", "fa16_dkv"):
    tot = collections.defaultdict(float); n = collections.Counter(); dur = 0.0; seen = set()
    for d in ("pa_m", "pa_f", "pa_w", "pa_l"):
        for r in load(d):
            if key not in r["Kernel_Name"]: continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            if d == "pa_m" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    L = max(n["SQ_WAVE_CYCLES"], 1); gui = tot["GRBM_GUI_ACTIVE"] / 8.0
    out[key] = {"launches": L, "avg_launch_ms_under_pmc": dur / L * 1e-6, "mfma_utilisation": tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0) if gui else None,
                "effective_clock_GHz": gui / dur if dur else None,
                "wave_cycle_split": {k: tot[k] / tot["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if tot.get("SQ_WAVE_CYCLES")},
                "fetch_GB_per_launch_x2_corrected": 2.0 * tot["FETCH_SIZE"] * 1024 / max(n["FETCH_SIZE"], 1) / 1e9, "write_GB_per_launch": tot["WRITE_SIZE"] * 1024 / max(n["WRITE_SIZE"], 1) / 1e9,
                "lds_bank_conflict_over_active": tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"] if tot.get("SQ_LDS_IDX_ACTIVE") else None}
out["note"] = ("separate rocprofv3 --pmc passes (SQ set; FETCH_SIZE; WRITE_SIZE; LDS) over tools/probes/attn16_probe_base time 4 15008 256 2 3; v_mfma_f32_32x32x16_f16 issues every 32 "
               "cycles per SIMD: utilisation 1.0 = 2.5 PFLOP/s at 2.4 GHz; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-byte requests at 64 B); operand arrays "
               "92 MB (forward) / 215 MB (backward) + fp32 outputs: L2 / Infinity-Cache resident, the HBM side is small by design")
json.dump(out, open("$OUT/${TAG}_attn16_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
rm -rf $OUT/pa_m $OUT/pa_f $OUT/pa_w $OUT/pa_l
