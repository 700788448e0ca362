import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch
import bench
from fastfourierdiffusion_amd import _native as N
dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, "ecg")
ctx = model._ctx(); lib = ctx.lib
s = N.current_stream_ptr(dev)
for B in (28, 32, 40, 50, 64, 80, 100, 128):
    row = []
    for name, t in (("auto", {}), ("hpw1", {"attn_hpw": 1}), ("hpw1 qg1", {"attn_hpw": 1, "attn_qg": 1}), ("hpw1 qg2", {"attn_hpw": 1, "attn_qg": 2})):
        lib.ffd_tune(b"reset", 0)
        for k, v in t.items():
            assert lib.ffd_tune(k.encode(), v) == 0, k
        ms, n = C.c_float(), C.c_int()
        N.check(lib.ffd_probe_attn(ctx.handle, B, -1, 0.3, 200, C.byref(ms), None, 0, C.byref(n), s), ctx.handle, "probe")
        row.append(f"{name} {ms.value*1e3:.1f}")
    print(f"B={B}: " + "  ".join(row), flush=True)
