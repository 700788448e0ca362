"""tools/pmc_summary.py <dir> [kernel-substr ...] : per-kernel mean of every PMC counter in a rocprofv3 --pmc run."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0][-72:]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
filt = sys.argv[2:]
for name, cs in acc.items():
    if filt and not any(s in name for s in filt):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):5d} mean={sum(v)/len(v):.5g}")
