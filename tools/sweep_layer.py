"""A/B of the fused-FFN kernel vs the fused layer kernel (with/without the QKV epilogue) in one process."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model, sch, sd = bench.build_model(torch.device("cuda", 0), "ecg")
ctx = model._ctx(); lib = ctx.lib
names = {0: "k_ffn_ln", 1: "k_layer (no next-QKV)", 2: "k_layer + next QKV"}
res = {k: [] for k in names}
for rnd in range(5):
    for k in names:
        lib.ffd_tune(b"bench_kernel", k)
        ms = C.c_float()
        N.check(lib.ffd_bench_ffn(ctx.handle, B, 20, C.byref(ms), None), ctx.handle)
        res[k].append(ms.value)
lib.ffd_tune(b"bench_kernel", 0)
for k, v in res.items():
    print(f"{names[k]:28s} median {statistics.median(v)*1e3:.1f} us  min {min(v)*1e3:.1f} us")
