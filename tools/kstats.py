"""Print a rocprofv3 kernel_stats.csv compactly: tools/kstats.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print("%-64s calls=%6s avg_us=%9.1f pct=%5.1f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                     100 * float(r["TotalDurationNs"]) / tot))
