import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"])
for w in d.get("other_workloads", []): print(w["workload"][:30], w["batch"], round(w["value"],2), round(w["ms_per_step"],3), w["roofline"])
print(d.get("api_e2e"), d.get("api_e2e_over_value"))
print(d["harness_b1"])
print(d["cpu_baseline"])
print([ (r["kernel"], round(r["frac"],3), r["ms_per_launch"]) for r in d["roofline_kernels"]])
print(d.get("cache_ratio"))
