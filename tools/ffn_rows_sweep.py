"""Row-owning FFN configurations (waves per workgroup x row blocks per wave): launch time and in-kernel main-loop
efficiency (MFMA issue cycles of a tile / stamped cycles of its main loop).\ntools/ffn_rows_sweep.py [B] [cfgs nw:cps[:prio[:fuse]],..]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from fastfourierdiffusion_amd import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfgs = [tuple(int(v) for v in c.split(":")) for c in (sys.argv[2] if len(sys.argv) > 2 else "4:1,4:2,8:1,8:2,12:1,12:2,6:1").split(",")]
dbgs = [0]
dev = torch.device("cuda", 0)
model, sch, _ = bench.build_model(dev, "ecg")
ctx = model._ctx(); lib = ctx.lib
s = N.current_stream_ptr(dev)
out = []
for cfg in cfgs:
  nw, mb = cfg[:2]
  prio = cfg[2] if len(cfg) > 2 else 1
  fuse = cfg[3] if len(cfg) > 3 else 1
  assert lib.ffd_tune(b"ffn_rows_fuse", fuse) == 0
  for dbg in dbgs:
    assert lib.ffd_tune(b"ffn_rows_nw", nw) == 0 and lib.ffd_tune(b"ffn_rows_cps", mb) == 0
    ms = C.c_float(); best = 1e9
    for _ in range(3):
        N.check(lib.ffd_bench_ffn(ctx.handle, B, 30, C.byref(ms), s), ctx.handle)
        best = min(best, ms.value * 1e3)
    cap = 1024
    raw = (C.c_uint64 * (8 * cap))()
    ghz, us, n = C.c_double(), C.c_double(), C.c_int()
    N.check(lib.ffd_probe_ffn_clock(ctx.handle, B, 0.5, C.byref(ghz), C.byref(us), raw, cap, C.byref(n), s), ctx.handle)
    r = np.frombuffer(raw, dtype=np.uint64).reshape(cap, 8)[: n.value].astype(np.float64)
    loop_us = float(np.median((r[:, 4] - r[:, 3]) * 0.01))
    wps = {4: 1, 8: 2, 12: 3}[nw]  # waves per SIMD
    mf = (68 * 64 + (0 if dbg & 1 else 32 * 8)) * 64 * wps  # MFMA issue cycles per SIMD and tile (32 rows per wave)
    out.append({"nw": nw, "mb": mb, "prio": prio, "fuse": fuse, "kernel_us": round(best, 1), "ghz": round(ghz.value, 3),
                "tile_loop_us": round(loop_us, 1), "loop_eff": round(mf / (loop_us * ghz.value * 1e3), 3),
                "exit_us_max": round(float(((r[:, 6] - r[:, 2].min()) * 0.01).max()), 1),
                # per-workgroup phases (median us): entry ramp over the grid, prologue (LN parameters + first ring
                # slots), main loop, epilogue (LN2 + stores), exit spread
                "entry_spread_us": round(float(((r[:, 2] - r[:, 2].min()) * 0.01).max()), 1),
                "prologue_us": round(float(np.median((r[:, 3] - r[:, 2]) * 0.01)), 1),
                "epilogue_us": round(float(np.median((r[:, 5] - r[:, 4]) * 0.01)), 1),
                "loop_us_min_max": [round(float(((r[:, 4] - r[:, 3]) * 0.01).min()), 1), round(float(((r[:, 4] - r[:, 3]) * 0.01).max()), 1)],
                "exit_spread_us": round(float(((r[:, 6].max() - r[:, 6]) * 0.01).max()), 1)})

print(json.dumps(out))
