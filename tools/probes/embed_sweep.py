import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from fastfourierdiffusion_amd import _native as N
dev = torch.device("cuda", 0)
for wl, B in (("syn512", 2048), ("ecg", 512)):
    model, sch, _ = bench.build_model(dev, wl)
    ctx = model._ctx(); lib = ctx.lib
    L, Cn, d = model.max_len, model.n_channels, model.d_model
    x = torch.randn(B, L, Cn, device=dev); sc = torch.empty_like(x)
    s = N.current_stream_ptr(dev)
    for thr in (65536, 131072, 262144, 524288, 1048576, 2097152, 4194304):
        lib.ffd_tune(b"embed_threads", thr)
        for _ in range(2):
            N.check(lib.ffd_score_forward(ctx.handle, x.data_ptr(), 0.5, sc.data_ptr(), B, s), ctx.handle, "fwd")
        N.check(lib.ffd_kernel_timing_begin(ctx.handle, 1 << N.K_EMBED, 8), ctx.handle, "b")
        for _ in range(4):
            N.check(lib.ffd_score_forward(ctx.handle, x.data_ptr(), 0.5, sc.data_ptr(), B, s), ctx.handle, "fwd")
        N.check(lib.ffd_kernel_timing_end(ctx.handle), ctx.handle, "e")
        ms, n = C.c_float(), C.c_int()
        lib.ffd_kernel_timing_get(ctx.handle, N.K_EMBED, C.byref(ms), C.byref(n))
        by = 4.0 * B * L * (Cn + d)
        print(f"{wl} B={B} threads={thr}: {ms.value*1e3:.1f} us  {by/ms.value/1e9:.2f} TB/s", flush=True)
