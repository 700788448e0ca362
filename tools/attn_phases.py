"""Where the time of ONE launch of the fused in-projection + attention kernel goes, per wave, from in-kernel stamps
(ffd_probe_attn): projection / table fill + norms / attention / epilogue, and inside the key-tile loop the shader cycles
of QK^T (until the scores are readable), mask + softmax, and P.V.  Also the launch time without stamps (HIP events).
tools/attn_phases.py [workload=ecg|syn512] [B] [n_recompute=-1]   (FFD_TUNE=key=v,... for ffd_tune knobs)"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from fastfourierdiffusion_amd import _native as N

wl = sys.argv[1] if len(sys.argv) > 1 else "ecg"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nrec = int(sys.argv[3]) if len(sys.argv) > 3 else -1
dev = torch.device("cuda", 0)
model, sch, _ = bench.build_model(dev, wl)
ctx = model._ctx()
s = N.current_stream_ptr(dev)
for kv in os.environ.get("FFD_TUNE", "").split(","):
    if kv:
        assert ctx.lib.ffd_tune(kv.split("=")[0].encode(), int(kv.split("=")[1])) == 0, kv
if nrec >= 0:  # cached modes: enable the cache and run one full step so that the tables exist
    from fastfourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    model.enable_caching()
    L = model.max_len
    model(DiffusableBatch(X=torch.randn(2, L, model.n_channels, device=dev), y=None,
                          timesteps=torch.full((2,), 0.5, device=dev)), recompute_tokens=set(range(L)), step=0)
cap = B * model.n_head * 4
raw = (C.c_uint64 * (16 * cap))()
ms, n = C.c_float(), C.c_int()
want_stamps = os.environ.get("NO_STAMPS") != "1"
N.check(ctx.lib.ffd_probe_attn(ctx.handle, B, nrec, 1.0, 200, C.byref(ms), raw if want_stamps else None, cap, C.byref(n), s),
        ctx.handle, "probe")
out = {"workload": wl, "B": B, "n_recompute": nrec, "us_per_launch": round(ms.value * 1e3, 2), "waves": int(n.value)}
if n.value:
    ru = np.frombuffer(raw, dtype=np.uint64).reshape(cap, 16)[: n.value]
    r = ru.astype(np.float64)
    q = lambda a: [round(float(v), 1) for v in np.percentile(a, [0, 10, 50, 90, 100])]
    life_cyc = r[:, 6] - r[:, 1]
    life_us = (r[:, 12] - r[:, 0]) * 0.01
    ghz = np.median(life_cyc / np.maximum(life_us, 1e-9)) * 1e-3
    t0 = r[:, 0].min()
    hw = ru[:, 11].astype(np.int64)
    # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]; XCC from HW_REG_XCC_ID is not in it
    simd = (hw >> 4) & 3
    out.update({
        "percentiles": [0, 10, 50, 90, 100],
        "shader_clock_ghz": round(float(ghz), 3),
        "launch_span_us": round(float((r[:, 12].max() - t0) * 0.01), 2),
        "entry_us": q((r[:, 0] - t0) * 0.01), "exit_us": q((r[:, 12] - t0) * 0.01),
        "wave_lifetime_us": q(life_us),
        "cycles": {
            "entry_to_projection": q(r[:, 2] - r[:, 1]),
            "projection": q(r[:, 3] - r[:, 2]),
            "tables_norms_barriers": q(r[:, 4] - r[:, 3]),
            "attention": q(r[:, 5] - r[:, 4]),
            "epilogue": q(r[:, 6] - r[:, 5]),
            "lifetime": q(life_cyc),
            "sum_qk": q(r[:, 7]), "sum_softmax": q(r[:, 8]), "sum_pv": q(r[:, 9]),
            "key_tiles": q(r[:, 10]),
            "per_tile_qk": q(r[:, 7] / np.maximum(r[:, 10], 1)), "per_tile_softmax": q(r[:, 8] / np.maximum(r[:, 10], 1)),
            "per_tile_pv": q(r[:, 9] / np.maximum(r[:, 10], 1)),
        },
        "share_of_lifetime": {k: round(float(np.sum(v) / np.sum(life_cyc)), 3) for k, v in {
            "entry_to_projection": r[:, 2] - r[:, 1], "projection": r[:, 3] - r[:, 2],
            "tables_norms_barriers": r[:, 4] - r[:, 3], "qk": r[:, 7], "softmax": r[:, 8], "pv": r[:, 9],
            "epilogue": r[:, 6] - r[:, 5]}.items()},
        "waves_per_simd_id": [int((simd == i).sum()) for i in range(4)],
    })
print(json.dumps(out))
