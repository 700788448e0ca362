"""Per-unit timeline of one k_lstm_wave launch (ffd_lstm_trace): when each (chunk, layer, tile) unit started, finished
its start-up and ended, and how long it waited on progress words.  tools/lstm_trace.py [B=512]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from fastfourierdiffusion_amd import _native as N
from fastfourierdiffusion_amd.utils.dataclasses import DiffusableBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
model, sch, _ = bench.build_model(dev, "nasa_lstm")
ctx = model._ctx()
L, Cn, NL = model.max_len, model.n_channels, 10
x = torch.randn(B, L, Cn, device=dev)
t = torch.full((B,), 0.4, device=dev)
for _ in range(3):
    model(DiffusableBatch(X=x, y=None, timesteps=t))
torch.cuda.synchronize()
cap = 16 * NL * ((B + 15) // 16)
N.check(ctx.lib.ffd_lstm_trace(ctx.handle, None, cap, None), ctx.handle, "arm")
model(DiffusableBatch(X=x, y=None, timesteps=t))
raw = (C.c_uint64 * (4 * cap))()
n = C.c_int()
N.check(ctx.lib.ffd_lstm_trace(ctx.handle, raw, cap, C.byref(n)), ctx.handle, "read")
r = np.frombuffer(raw, dtype=np.uint64).reshape(cap, 4)
used = r[:, 2] != 0
r = r[used]
idx = np.nonzero(used)[0]
tiles = (B + 15) // 16
V = NL * tiles
t0 = r[:, 0].min()
start = (r[:, 0] - t0).astype(np.float64) * 0.01
up = (r[:, 1] - r[:, 0]).astype(np.float64) * 0.01
end = (r[:, 2] - t0).astype(np.float64) * 0.01
wait = (r[:, 3] & np.uint64((1 << 48) - 1)).astype(np.float64) * 0.01
wg = (r[:, 3] >> np.uint64(48)).astype(np.int64)
kc, rem = idx // V, idx % V
layer = rem // tiles
out = {"B": B, "units": int(len(idx)), "chunks": int(kc.max() + 1), "launch_span_us": round(float(end.max()), 1)}
q = lambda a: [round(float(v), 1) for v in np.percentile(a, [0, 10, 50, 90, 100])]
out["unit_duration_us"] = q(end - start)
out["startup_us"] = q(up)
out["wait_us_per_unit"] = q(wait)
out["by_layer_chunk0"] = {int(l): {"start": q(start[(kc == 0) & (layer == l)]), "wait": q(wait[(kc == 0) & (layer == l)]),
                                   "dur": q((end - start)[(kc == 0) & (layer == l)])} for l in range(NL)}
last = int(kc.max())
out["by_layer_last_chunk"] = {int(l): {"start": q(start[(kc == last) & (layer == l)]), "end": q(end[(kc == last) & (layer == l)]),
                                       "wait": q(wait[(kc == last) & (layer == l)])} for l in range(NL)}
# busy fraction per workgroup: sum of unit durations minus waits over the launch span
nwg = int(wg.max() + 1)
busy = np.zeros(nwg); waited = np.zeros(nwg)
np.add.at(busy, wg, end - start); np.add.at(waited, wg, wait)
out["wg_busy_frac"] = q(busy / end.max())
out["wg_wait_frac"] = q(waited / end.max())
out["total_wait_share"] = round(float(waited.sum() / (nwg * end.max())), 3)
out["idle_share"] = round(float(1.0 - busy.sum() / (nwg * end.max())), 3)
print(json.dumps(out))
