"""A/B of the fused FFN's launch forms on one box: one workgroup per tile (ffn_persist=0) vs the persistent grid
(1 = resident workgroups, 2 = twice that), crossed with the start stagger.  tools/sweep_ffn_persist.py [workload] [B]"""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from fastfourierdiffusion_amd import _native as N
wl = sys.argv[1] if len(sys.argv) > 1 else "ecg"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
model, sch, sd = bench.build_model(torch.device("cuda", 0), wl)
ctx = model._ctx(); lib = ctx.lib
fl = lib.ffd_ffn_flops_per_launch(ctx.handle, B)
cfgs = [(p, dy, st) for (p, dy) in ((0, 0), (0, 1), (1, 0), (1, 1)) for st in (-1, 0)]  # dy doubles as the ffn_prio flag here
res = {c: [] for c in cfgs}
for rnd in range(3):
    for (p, dy, st) in cfgs:
        assert lib.ffd_tune(b"ffn_persist", p) == 0 and lib.ffd_tune(b"ffn_stagger", st) == 0
        assert lib.ffd_tune(b"ffn_prio", dy) == 0 and lib.ffd_tune(b"ffn_dynamic", 0) == 0
        ms = C.c_float()
        N.check(lib.ffd_bench_ffn(ctx.handle, B, 20, C.byref(ms), None), ctx.handle)
        res[(p, dy, st)].append(ms.value)
for c in cfgs:
    m = statistics.median(res[c])
    print(f"{wl} B={B} persist={c[0]} prio={c[1]} stagger={c[2]:5d}: median {m*1e3:7.1f} us  min {min(res[c])*1e3:7.1f}  {fl/m/1e9:6.1f} TFLOP/s  frac {fl/m/1e9/157.3:.3f}")
