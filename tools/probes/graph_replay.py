"""Does replaying the small-batch sampling loop as a HIP graph shorten the step?  Captures n steps of ffd_sample_batch
(philox noise: step indices are baked into the captured launches, fine for timing) and times stream launches against
graph replays.  tools/probes/graph_replay.py [B,...]"""
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
n = 100
for B in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,8").split(",")]:
    x = torch.randn(B, 187, 1, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        s = st.cuda_stream
        for _ in range(2):
            N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 1000, float(sch.step_size), 0, n, 1, 0, None, 0, 0, s), ctx.handle, "w")
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 1000, float(sch.step_size), 0, n, 1, 0, None, 0, 0, s), ctx.handle, "t")
        st.synchronize()
        t_stream = (time.perf_counter() - t0) / (3 * n)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=st):
            N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 1000, float(sch.step_size), 0, n, 1, 0, None, 0, 0, st.cuda_stream), ctx.handle, "c")
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / (3 * n)
        print(f"B={B}: stream launches {t_stream*1e3:.4f} ms/step, graph replay {t_graph*1e3:.4f} ms/step", flush=True)
    except Exception as e:
        print(f"B={B}: stream launches {t_stream*1e3:.4f} ms/step, capture failed: {e}", flush=True)
