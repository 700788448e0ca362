"""tools/hbm_table.py <rocprof dir or kernel_stats.csv> <bytes.json> : GB/s per HBM-bound kernel vs 8 TB/s."""
import csv, glob, json, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(p + "/**/*kernel_stats.csv", recursive=True)[0]
meta = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
by = meta["algorithmic_bytes"]
print("kernel,calls,avg_us,algorithmic_MB,GB_per_s,frac_of_8TBps")
for r in csv.DictReader(open(f)):
    name = r["Name"]
    key = None
    for k in by:  # longest matching key wins (k_fresca_apply also matches k_fresca_apply_pow2, ...)
        if k in name and (key is None or len(k) > len(key)):
            key = k
    if key is None:
        continue
    us = float(r["AverageNs"]) / 1e3
    b = by[key]
    gbs = b / (us * 1e-6) / 1e9
    short = name.split("(")[0].replace("void ", "").replace("ffd::", "")
    print(f"{short[:48]},{r['Calls']},{us:.1f},{b / 1e6:.1f},{gbs:.0f},{gbs / 8000:.3f}")
