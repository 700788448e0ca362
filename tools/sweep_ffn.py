"""Interleaved A/B of the fused-FFN tile heights in one process (cdna guide rule 24)."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wl = sys.argv[2] if len(sys.argv) > 2 else "ecg"
model, sch, sd = bench.build_model(dev, wl)
ctx = model._ctx(); lib = ctx.lib
fl = lib.ffd_ffn_flops_per_launch(ctx.handle, B)
res = {(mb, rem): [] for mb in (4, 2, 1) for rem in (0, 1)}
for rnd in range(5):
    for (mb, rem) in res:
        lib.ffd_tune(b"ffn_mb", mb)
        lib.ffd_tune(b"ffn_rem", rem)
        ms = C.c_float()
        N.check(lib.ffd_bench_ffn(ctx.handle, B, 20, C.byref(ms), None), ctx.handle)
        res[(mb, rem)].append(ms.value)
lib.ffd_tune(b"ffn_mb", 0)
lib.ffd_tune(b"ffn_rem", 1)
for mb, v in res.items():
    med = statistics.median(v)
    print(f"MB,rem={mb}: median {med*1e3:.1f} us  min {min(v)*1e3:.1f} us  -> {fl/med/1e9:.1f} TFLOP/s ({fl/med/1e9/157.3*100:.1f}% of fp32 MFMA peak)")
