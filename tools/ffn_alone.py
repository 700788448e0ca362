"""Main-loop time of the fused FFN with ONE workgroup per CU (255 tiles of 64 rows, ffn_mb = 4) against two per CU
(510 tiles): how much of the matrix pipe a single wave per SIMD fills.  In-kernel stamps (ffd_probe_ffn_clock)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
dev = torch.device("cuda", 0)
model, sch, _ = bench.build_model(dev, "ecg")
ctx = model._ctx()
s = N.current_stream_ptr(dev)
assert ctx.lib.ffd_tune(b"ffn_rows", 0) == 0 and ctx.lib.ffd_tune(b"ffn_mb", 4) == 0  # (k_ffn_ln, the F-split form)
out = []
for B in (87, 174, 512):  # 16 269 rows = 255 tiles; 32 538 rows = 509 tiles; the bench shape
    ghz, us = C.c_double(), C.c_double()
    N.check(ctx.lib.ffd_probe_ffn_clock(ctx.handle, B, 1.0, C.byref(ghz), C.byref(us), None, 0, None, s), ctx.handle, "probe")
    ms = C.c_float()
    N.check(ctx.lib.ffd_bench_ffn(ctx.handle, B, 50, C.byref(ms), s), ctx.handle, "bench")
    tiles = (B * 187 + 63) // 64
    out.append({"B": B, "tiles": tiles, "shader_clock_ghz": ghz.value, "main_loops_us_median_per_workgroup": us.value,
                "kernel_us": ms.value * 1e3, "mfma_floor_us_per_tile_at_clock": 150.6e3 / (ghz.value * 1e3)})
print(json.dumps(out))
