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
    for k in by:
        base = k.split("(")[0]
        if base in name:
            key = base if key is None or len(base) > len(key) else key
    if key is None:
        continue
    us = float(r["AverageNs"]) / 1e3
    b = by[key]
    if key == "k_sde_step" and "k_unembed_sde" not in name:
        b = by["k_sde_step"]  # mixed Philox / injected launches share one row: quote the 12 B/element figure
    gbs = b / (us * 1e-6) / 1e9
    print(f"{name.split('(')[0][-48:]},{r['Calls']},{us:.1f},{b / 1e6:.1f},{gbs:.0f},{gbs / 8000:.3f}")
