"""ms/step of the ECG sampler at B=512 under one ffd_tune key (A/B in one process): tools/probes/step_ab.py key v0 v1"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from fastfourierdiffusion_amd import _native as N
key = sys.argv[1].encode() if len(sys.argv) > 1 else b""
vals = [int(v) for v in sys.argv[2:]] or [0]
dev = torch.device("cuda", 0)
model, sch, _ = bench.build_model(dev, "ecg")
ctx = model._ctx(); lib = ctx.lib
B, L, Cn = 512, model.max_len, model.n_channels
sch.set_timesteps(1000)
ts_c = (C.c_float * 1000)(*sch.timesteps.tolist())
x = torch.randn(B, L, Cn, device=dev)
s = N.current_stream_ptr(dev)
for rep in range(3):
    for v in vals:
        if key: lib.ffd_tune(key, v)
        N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 1000, float(sch.step_size), 0, 10, 1, 0, None, 0, 0, s), ctx.handle, "w")
        torch.cuda.synchronize(); t0 = time.time()
        N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 1000, float(sch.step_size), 0, 100, 1, 0, None, 0, 0, s), ctx.handle, "r")
        torch.cuda.synchronize()
        print(f"{key.decode()}={v}: {(time.time()-t0)/100*1e3:.3f} ms/step", flush=True)
