"""MFMA utilisation of the dominant kernel from a rocprofv3 --pmc pass of bench.py.
usage: python tools/pmc_mfma_summary.py <counter_collection.csv> <out.json>
Counters (one pass: 5 SQ + 1 GRBM slot): SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs), GRBM_GUI_ACTIVE (cycles the GPU was busy: the
kernel's duration in shader clocks), SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY (quad-cycles; MI355X_MICROARCH.md).
GRBM_GUI_ACTIVE comes summed over the 8 XCDs (one GRBM each; 8.3e6 "cycles" for a 0.46 ms launch), so per-XCD cycles = GUI_ACTIVE / 8:
MFMA utilisation = MFMA busy cycles / (GUI_ACTIVE / 8 x 1024 SIMDs); effective clock = GUI_ACTIVE / 8 / kernel duration (CSV timestamps)."""
import csv, json, sys
KEY = sys.argv[3] if len(sys.argv) > 3 else None     # kernel-name substring; default: the batched GEMM that ran (f16x2, else bf16x3; fp32 reference run: "2, 2, 36>")
if KEY is None:
    names = {r["Kernel_Name"] for r in csv.DictReader(open(sys.argv[1]))}
    KEY = "wgemm_f16x2_rt2_kernel" if any("wgemm_f16x2_rt2_kernel" in k for k in names) else "wgemm_bf16x3_kernel<false"
tot, n, dur = {}, {}, 0.0
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    if KEY not in r["Kernel_Name"]:
        continue
    c = r["Counter_Name"]
    tot[c] = tot.get(c, 0.0) + float(r["Counter_Value"]); n[c] = n.get(c, 0) + 1
    did = r.get("Dispatch_Id")
    if did not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
        seen.add(did); dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
launches = max(n.values()) if n else 0
gui = tot.get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # per XCD
out = {"kernel": KEY + " (batched Winograd-domain GEMMs)", "launches": launches, "counters_sum": tot, "avg_launch_ms_under_pmc": dur / max(1, launches) * 1e-6,
       "mfma_utilisation": tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0) if gui else None,
       "effective_clock_GHz": (gui / dur) if dur else None,
       "wave_cycle_split": {k: tot[k] / tot["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in tot and tot.get("SQ_WAVE_CYCLES")},
       "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace over "
               "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-concurrent 0`; v_mfma_f32_32x32x2_f32 issues every 64 cycles per SIMD, so "
               "utilisation 1.0 = the 157.3 TFLOP/s nominal rate at 2.4 GHz; achieved TFLOP/s = utilisation x 157.3 x (effective clock / 2.4).  "
               "v_mfma_f32_32x32x16_bf16 / _f16 (the bf16x3 / f16x2 kernels) issue every 32 cycles: utilisation 1.0 = 2.5 PFLOP/s = 417 (bf16x3) / 833 (f16x2) TFLOP/s fp32-equivalent"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out)[:1200])
