"""Merge rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/r04_traffic.json.
tools/traffic_json.py <key-prefix e.g. ecg:512> <fetch dir> <write dir> [out.json]
HBM bytes per launch = 2 x FETCH_SIZE (gfx950 reports half of wide streaming reads, MI355X_MICROARCH.md section HBM)
+ WRITE_SIZE; both counters are in KB."""
import collections, csv, glob, json, os, sys
prefix, fdir, wdir = sys.argv[1:4]
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_traffic.json")
NAMES = [("k_ffn_rows", "k_ffn_rows"), ("k_lstm_wave", "k_lstm_wave"), ("k_ffn_ln", "k_ffn_ln"), ("k_qkv_attention", "k_qkv_attention"), ("k_linear_res_ln", "k_linear_res_ln"),
         ("k_embed", "k_embed"), ("k_unembed_mfma<72, true>", "k_unembed_mfma<sde>"), ("k_unembed_mfma<72, false>", "k_unembed"),
         ("k_lstm_mfma", "k_lstm_mfma"), ("k_lstm_layer", "k_lstm_layer"), ("k_linear_rm", "k_linear_rm"), ("k_sde_step", "k_sde_step"),
         ("k_rfft_pow2", "k_rfft_pow2")]
def means(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = means(fdir, "FETCH_SIZE"), means(wdir, "WRITE_SIZE")
tab = json.load(open(out)) if os.path.exists(out) else {}
for kname in fe:
    for sub, key in NAMES:
        if sub in kname:
            b = 2.0 * fe[kname] * 1024.0 + wr.get(kname, 0.0) * 1024.0
            tab[f"{prefix}:{key}"] = b
            tab[f"{prefix}:{key}:detail"] = {"FETCH_SIZE_KB": fe[kname], "WRITE_SIZE_KB": wr.get(kname, 0.0),
                                              "formula": "2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes)"}
            break
json.dump(tab, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in tab.items() if k.startswith(prefix) and not k.endswith("detail")}, indent=1))
