"""Print a rocprofv3 --stats kernel_stats.csv as a short table.  usage: python tools/stats_table.py <dir-or-csv> [top] [divide_calls_by]"""
import csv, glob, os, sys
f = sys.argv[1]
if os.path.isdir(f):
    f = glob.glob(os.path.join(f, "**", "*kernel_stats.csv"), recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(f)))
print("total ms", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6)
for r in rows[:top]:
    n = r["Name"].replace("buddy::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]
    print("%-46s %6d %9.2f ms  avg %8.1f us  %5.1f%%" % (n, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
