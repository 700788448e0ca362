"""In-situ HIP-event time per kernel class for a workload / batch: tools/probes/attn_time.py ecg 512"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from fastfourierdiffusion_amd import _native as N
wl = sys.argv[1] if len(sys.argv) > 1 else "ecg"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda", 0)
model, sch, _ = bench.build_model(dev, wl)
ctx = model._ctx(); lib = ctx.lib
L, Cn, NL = model.max_len, model.n_channels, model.num_layers
sch.set_timesteps(50)
ts_c = (C.c_float * 50)(*sch.timesteps.tolist())
x = torch.randn(B, L, Cn, device=dev)
s = N.current_stream_ptr(dev)
nst = 4
for rep in range(3):
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, 2, 1, 0, None, 0, 0, s), ctx.handle, "warm")
    N.check(lib.ffd_kernel_timing_begin(ctx.handle, 0xFF, nst * (3 * NL + 3)), ctx.handle, "begin")
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, nst, 1, 0, None, 0, 0, s), ctx.handle, "sample")
    N.check(lib.ffd_kernel_timing_end(ctx.handle), ctx.handle, "end")
    row = []
    for name, cls in (("ffn", N.K_FFN), ("attention", N.K_ATTN), ("out-proj", N.K_OUTPROJ), ("embed", N.K_EMBED), ("tail", N.K_SDE)):
        ms, n = C.c_float(), C.c_int()
        lib.ffd_kernel_timing_get(ctx.handle, cls, C.byref(ms), C.byref(n))
        row.append(f"{name} {ms.value*1e3:.1f}")
    print(f"{wl} B={B}: us per launch (in-situ HIP events): " + "  ".join(row), flush=True)
