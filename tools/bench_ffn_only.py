"""Run only the fused-FFN kernel (for rocprofv3 --pmc passes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wl = sys.argv[3] if len(sys.argv) > 3 else "ecg"
model, sch, sd = bench.build_model(torch.device("cuda", 0), wl)
ctx = model._ctx(); lib = ctx.lib
lib.ffd_tune(b"ffn_mb", mb)
if os.environ.get("FFN_ROWS") is not None:
    assert lib.ffd_tune(b"ffn_rows", int(os.environ["FFN_ROWS"])) == 0
for kv in os.environ.get("FFD_TUNE", "").split(","):
    if kv:
        assert lib.ffd_tune(kv.split("=")[0].encode(), int(kv.split("=")[1])) == 0, kv
ms = C.c_float()
N.check(lib.ffd_bench_ffn(ctx.handle, B, 10, C.byref(ms), None), ctx.handle)
fl = lib.ffd_ffn_flops_per_launch(ctx.handle, B)
print(f"B={B} mb={mb} {ms.value*1e3:.1f} us {fl/ms.value/1e9:.1f} TFLOP/s")
