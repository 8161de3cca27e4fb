"""Print the per-kernel table of rocprofv3 --stats csv outputs: python tools/kstats.py <dir-or-csv> [...]"""
import csv, glob, os, sys

def show(path):
    rows = list(csv.DictReader(open(path)))
    print("==", path)
    for r in rows[:16]:
        name = r["Name"].replace("void bh::", "").replace("bh::", "").replace("(anonymous namespace)::", "")
        name = name.split("(")[0][:58]
        print("  %-60s calls %5s avg %10.1f us  tot %10.1f us  %6s%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))

for a in sys.argv[1:]:
    if os.path.isdir(a):
        for f in sorted(glob.glob(os.path.join(a, "**", "*kernel_stats.csv"), recursive=True)):
            show(f)
    else:
        show(a)
