"""Un-profiled shader clock under the dominant kernel (fused FFN, ECG B=512 shape): in-kernel s_memtime / s_memrealtime
stamps after `warm` seconds of back-to-back launches (ffd_probe_ffn_clock).  tools/ffn_clock.py [B] [warm_seconds]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
warm = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
dev = torch.device("cuda", 0)
model, sch, _ = bench.build_model(dev, "ecg")
ctx = model._ctx()
s = N.current_stream_ptr(dev)
out = []
for w in (0.0, warm, warm):
    ghz, us = C.c_double(), C.c_double()
    N.check(ctx.lib.ffd_probe_ffn_clock(ctx.handle, B, w, C.byref(ghz), C.byref(us), None, 0, None, s), ctx.handle, "probe")
    ms = C.c_float()
    N.check(ctx.lib.ffd_bench_ffn(ctx.handle, B, 50, C.byref(ms), s), ctx.handle, "bench")
    out.append({"warm_s": w, "shader_clock_ghz": ghz.value, "main_loop_us_median": us.value, "kernel_us_back_to_back": ms.value * 1e3})
fl = ctx.lib.ffd_ffn_flops_per_launch(ctx.handle, B)
for o in out:
    o["tflops"] = fl / (o["kernel_us_back_to_back"] * 1e-6) / 1e12
    o["fp32_mfma_peak_at_this_clock_tflops"] = 256 * 4 * 64 * o["shader_clock_ghz"] * 1e9 / 1e12
    o["frac_of_peak_at_this_clock"] = o["tflops"] / o["fp32_mfma_peak_at_this_clock_tflops"]
print(json.dumps({"B": B, "probes": out}))
