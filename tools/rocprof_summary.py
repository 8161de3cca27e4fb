"""Summarise a rocprofv3 kernel trace (rocpd sqlite .db or *_kernel_trace.csv) into a per-kernel table."""
import csv, glob, os, sqlite3, sys

def from_db(path):
    db = sqlite3.connect(path)
    return db.execute("select name, count(*), avg(end-start), sum(end-start), min(end-start), max(end-start) from kernels group by name").fetchall()

def from_csv(path):
    agg = {}
    for r in csv.DictReader(open(path)):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    return [(k, v[0], v[1] / v[0], v[1], v[2], v[3]) for k, v in agg.items()]

def main():
    src = sys.argv[1]
    rows = []
    if os.path.isdir(src):
        for f in glob.glob(os.path.join(src, "**", "*.db"), recursive=True): rows += from_db(f)
        for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True): rows += from_csv(f)
    elif src.endswith(".db"): rows = from_db(src)
    else: rows = from_csv(src)
    rows.sort(key=lambda r: -r[3])
    tot = sum(r[3] for r in rows) or 1
    print("%-78s %6s %12s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us", "pct"))
    for name, calls, avg, total, mn, mx in rows:
        print("%-78s %6d %12.1f %12.1f %12.1f %12.1f %7.2f" % (name[:78], calls, avg / 1e3, mn / 1e3, mx / 1e3, total / 1e3, 100.0 * total / tot))

main()
