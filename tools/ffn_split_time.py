"""FFN launch time, fp32 MFMA kernel vs the opt-in bf16x3-split kernel (in-situ HIP events), and whole-step time."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N

wl = sys.argv[1] if len(sys.argv) > 1 else "ecg"
Bs = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512]
dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, wl)
ctx = model._ctx(); lib = ctx.lib
L, Cn = model.max_len, model.n_channels
NL = model.num_layers
sch.set_timesteps(50)
ts_c = (C.c_float * 50)(*sch.timesteps.tolist())
for B in Bs:
    x = torch.randn(B, L, Cn, device=dev)
    s = N.current_stream_ptr(dev)
    for split in (0, 1, 0, 1):
        lib.ffd_tune(b"ffn_split", split)
        nst = 6
        N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, 3, 1, 0, None, 0, 0, s), ctx.handle, "warm")
        torch.cuda.synchronize(); t0 = time.time()
        N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, 20, 1, 0, None, 0, 0, s), ctx.handle, "run")
        torch.cuda.synchronize(); step_ms = (time.time() - t0) / 20 * 1e3
        N.check(lib.ffd_kernel_timing_begin(ctx.handle, 0xFF, nst * (3 * NL + 3)), ctx.handle, "begin")
        N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, nst, 1, 0, None, 0, 0, s), ctx.handle, "sample")
        N.check(lib.ffd_kernel_timing_end(ctx.handle), ctx.handle, "end")
        ms, n = C.c_float(), C.c_int()
        lib.ffd_kernel_timing_get(ctx.handle, N.K_FFN, C.byref(ms), C.byref(n))
        fl, by = C.c_double(), C.c_double()
        name = lib.ffd_kernel_work(ctx.handle, N.K_FFN, B, 0, C.byref(fl), C.byref(by))
        print(f"{wl} B={B} split={split}: {name.decode()} {ms.value*1e3:.1f} us  {fl.value/ms.value/1e9:.1f} TFLOP/s(fp32-equivalent)  step {step_ms:.3f} ms", flush=True)
lib.ffd_tune(b"ffn_split", 0)
