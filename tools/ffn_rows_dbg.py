"""Timing experiments on the row-owning FFN (ffd_tune "ffn_rows_dbg" bits: 1 no remainder MFMAs, 2 no slot barrier /
waits, 4 no ring DMA): results are wrong with any bit set; launch time only."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, "ecg")
ctx = model._ctx(); lib = ctx.lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
out = {}
def t(tag):
    ms = C.c_float()
    best = 1e9
    for _ in range(3):
        N.check(lib.ffd_bench_ffn(ctx.handle, B, 30, C.byref(ms), None), ctx.handle)
        best = min(best, ms.value * 1e3)
    out[tag] = round(best, 1)
lib.ffd_tune(b"ffn_rows", 0); t("k_ffn_ln")
lib.ffd_tune(b"ffn_rows", 1)
for dbg in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,4,6,7".split(","))]:
    assert lib.ffd_tune(b"ffn_rows_dbg", dbg) == 0
    t(f"rows_dbg{dbg}")
lib.ffd_tune(b"ffn_rows_dbg", 0)
print(json.dumps(out))
