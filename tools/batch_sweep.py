"""samples/s of the ECG line over the batch size (throughput should not fall when the batch
grows): one model, `bench.run_steps` timed with HIP events per batch, the FFN form the forward pass picked printed next
to it.  tools/batch_sweep.py [B ...]   (FFD_TUNE=key=v,... for ffd_tune knobs)"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from fastfourierdiffusion_amd import _native as N

Bs = [int(a) for a in sys.argv[1:]] or [200, 256, 320, 384, 448, 512, 640, 768, 1024]
dev = torch.device("cuda", 0)
model, sch, _ = bench.build_model(dev, "ecg")
ctx = model._ctx()
lib = ctx.lib
for kv in os.environ.get("FFD_TUNE", "").split(","):
    if kv:
        assert lib.ffd_tune(kv.split("=")[0].encode(), int(kv.split("=")[1])) == 0, kv
stream = N.current_stream_ptr(dev)
sch.set_timesteps(1000)
ts = sch.timesteps.to(torch.float32).contiguous()
ts_c = ts.numpy().ctypes.data_as(C.POINTER(C.c_float))
step_size = float(sch.step_size)
L, Cn = model.max_len, model.n_channels
rows = []
best = 0.0
for B in Bs:
    X = torch.randn(B, L, Cn, device=dev)
    steps, warm = 30, 5
    bench.run_steps(model, X, ts_c, 1000, step_size, 0, warm, False, stream, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    bench.run_steps(model, X, ts_c, 1000, step_size, warm, steps, False, stream, 0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    v = B / ms
    fl, by = C.c_double(), C.c_double()
    name = lib.ffd_kernel_work(ctx.handle, N.K_FFN, B, 0, C.byref(fl), C.byref(by))
    rows.append({"B": B, "samples_per_s": round(v, 2), "ms_per_step": round(ms, 4), "ffn_form": name.decode() if name else None,
                 "below_running_max_pct": round(100.0 * (1.0 - v / best), 1) if best > 0 and v < best else 0.0})
    best = max(best, v)
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"sweep": rows}))
