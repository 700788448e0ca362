"""Is the small-batch sampling loop bound by the host's launch rate?  Time for ffd_sample_batch to RETURN (all launches
enqueued) against the time until the stream has drained, per diffusion step.  tools/probes/enqueue_vs_gpu.py [B,...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N

dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, "ecg")
ctx = model._ctx(); lib = ctx.lib
sch.set_timesteps(1000)
ts_c = (C.c_float * 1000)(*sch.timesteps.tolist())
s = N.current_stream_ptr(dev)
for B in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,8,50").split(",")]:
    x = torch.randn(B, 187, 1, device=dev)
    n = 300
    for rep in range(2):
        N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 1000, float(sch.step_size), 0, n, 1, 0, None, 0, 0, s), ctx.handle, "w")
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 1000, float(sch.step_size), 0, n, 1, 0, None, 0, 0, s), ctx.handle, "t")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: enqueue {1e3 * (t1 - t0) / n:.4f} ms/step, drained {1e3 * (t2 - t0) / n:.4f} ms/step", flush=True)
