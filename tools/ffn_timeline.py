"""Timeline of ONE launch of the fused FFN at the bench shape from in-kernel 100 MHz timestamps (ffd_probe_ffn_clock):
when workgroups enter, how long prologue / main loop / epilogue of their first tile take, when they exit.
tools/ffn_timeline.py [B] [ffn_rows] [persist]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from fastfourierdiffusion_amd import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # 1: k_ffn_rows (default), 0: k_ffn_ln
persist = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
model, sch, _ = bench.build_model(dev, "ecg")
ctx = model._ctx()
s = N.current_stream_ptr(dev)
assert ctx.lib.ffd_tune(b"ffn_persist", persist) == 0 and ctx.lib.ffd_tune(b"ffn_rows", rows) == 0
for kv in os.environ.get("FFD_TUNE", "").split(","):
    if kv:
        assert ctx.lib.ffd_tune(kv.split("=")[0].encode(), int(kv.split("=")[1])) == 0, kv
if os.environ.get("FFN_SPLIT") == "1":  # the opt-in bf16x3-split kernel (its packs are made by a first forward)
    from fastfourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    assert ctx.lib.ffd_tune(b"ffn_split", 1) == 0
    model(DiffusableBatch(X=torch.randn(2, model.max_len, model.n_channels, device=dev), y=None,
                          timesteps=torch.full((2,), 0.5, device=dev)))
cap = 4096
raw = (C.c_uint64 * (8 * cap))()
ghz, us, n = C.c_double(), C.c_double(), C.c_int()
N.check(ctx.lib.ffd_probe_ffn_clock(ctx.handle, B, 1.0, C.byref(ghz), C.byref(us), raw, cap, C.byref(n), s), ctx.handle, "probe")
ru = np.frombuffer(raw, dtype=np.uint64).reshape(cap, 8)[: n.value]
cu = (ru[:, 7] >> np.uint64(32)).astype(np.int64)  # (xcc, se, cu) of the workgroup
r = ru.astype(np.float64)
r[:, 7] = (ru[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.float64)
t0 = r[:, 2].min()
q = lambda a: [round(float(v), 2) for v in np.percentile(a, [0, 10, 50, 90, 100])]
tk = 0.01  # us per tick
out = {"B": B, "ffn_rows": rows, "persist": persist, "workgroups": int(n.value), "shader_clock_ghz": ghz.value,
       "tiles_per_workgroup": q(r[:, 7]),
       "entry_us": q((r[:, 2] - t0) * tk), "exit_us": q((r[:, 6] - t0) * tk),
       "first_prologue_us": q((r[:, 3] - r[:, 2]) * tk), "first_main_loop_us": q((r[:, 4] - r[:, 3]) * tk),
       "first_epilogue_us": q((r[:, 5] - r[:, 4]) * tk), "main_loops_total_us": q(r[:, 1] * tk),
       "lifetime_us": q((r[:, 6] - r[:, 2]) * tk), "percentiles": [0, 10, 50, 90, 100]}
# per CU: how many workgroups it hosted, when its last one left, how far apart its workgroups started
cus = {}
for i in range(len(cu)):
    cus.setdefault(int(cu[i]), []).append(i)
last = np.array([max((r[i, 6] - t0) * tk for i in v) for v in cus.values()])
cnt = np.array([len(v) for v in cus.values()])
tiles_cu = np.array([sum(r[i, 7] for i in v) for v in cus.values()])
out["cus_used"] = len(cus)
out["workgroups_per_cu"] = q(cnt)
out["tiles_per_cu"] = q(tiles_cu)
out["cu_last_exit_us"] = q(last)
late = [k for k, v in cus.items() if max((r[i, 6] - t0) * tk for i in v) > np.percentile(last, 85)]
out["late_cus_tiles"] = sorted(float(sum(r[i, 7] for i in cus[k])) for k in late)[:40]
out["late_cus_workgroups"] = sorted(len(cus[k]) for k in late)[:40]
# which block indices share a CU (record i belongs to blockIdx.x = i when every workgroup of the grid wrote one)
diffs = {}
for v in cus.values():
    if len(v) == 2:
        d = abs(v[0] - v[1])
        diffs[d] = diffs.get(d, 0) + 1
out["cu_pair_block_distance"] = dict(sorted(diffs.items(), key=lambda kv: -kv[1])[:6])
out["first_on_cu_is_faster"] = int(sum(1 for v in cus.values() if len(v) == 2 and r[min(v), 6] < r[max(v), 6]))
print(json.dumps(out))
