"""Summarise rocprofv3 --pmc passes of bench.py into per-launch HBM traffic of the dominant kernel.
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [kernel-substring]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section) -> doubled here."""
import csv, json, sys

KERNEL = sys.argv[4] if len(sys.argv) > 4 else "wino3_kernel"


def avg(path, counter, kernel=None):
    kernel = kernel or KERNEL
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)

f, nf = avg(sys.argv[1], "FETCH_SIZE")
w, nw = avg(sys.argv[2], "WRITE_SIZE")
out = {"kernel": KERNEL, "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_bytes_per_launch": 2 * f * 1024, "write_bytes_per_launch": w * 1024,
       "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline`; FETCH_SIZE x2 (gfx950 correction)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
