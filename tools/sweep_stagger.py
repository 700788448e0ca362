"""Sweep the co-resident-workgroup start stagger (ffd_tune "ffn_stagger") for k_ffn_ln / k_layer."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from fastfourierdiffusion_amd import _native as N
kern = 0
vals = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 200, 400, 800, 1200]
model, sch, sd = bench.build_model(torch.device("cuda", 0), "ecg")
ctx = model._ctx(); lib = ctx.lib
res = {v: [] for v in vals}
for rnd in range(4):
    for v in vals:
        assert lib.ffd_tune(b"ffn_stagger", v) == 0
        ms = C.c_float()
        N.check(lib.ffd_bench_ffn(ctx.handle, 512, 20, C.byref(ms), None), ctx.handle)
        res[v].append(ms.value)
for v in vals:
    print(f"kernel={kern} stagger={v:4d} (x64 cycles): median {statistics.median(res[v])*1e3:.1f} us  min {min(res[v])*1e3:.1f}")
