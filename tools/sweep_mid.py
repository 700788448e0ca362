"""Mid-size batches: per-launch time of the FFN class (and the out-projection) for tile heights / the F-split pair."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N

dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, "ecg")
ctx = model._ctx(); lib = ctx.lib
NL, L = 10, 187
sch.set_timesteps(50)
ts_c = (C.c_float * 50)(*sch.timesteps.tolist())
def run(B):
    x = torch.randn(B, L, 1, device=dev)
    s = N.current_stream_ptr(dev)
    nst = 6
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, 2, 1, 0, None, 0, 0, s), ctx.handle, "warm")
    N.check(lib.ffd_kernel_timing_begin(ctx.handle, 0xFF, nst * (3 * NL + 3)), ctx.handle, "begin")
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, nst, 1, 0, None, 0, 0, s), ctx.handle, "sample")
    N.check(lib.ffd_kernel_timing_end(ctx.handle), ctx.handle, "end")
    out = {}
    for cls in (N.K_FFN, N.K_OUTPROJ, N.K_ATTN):
        ms, n = C.c_float(), C.c_int()
        lib.ffd_kernel_timing_get(ctx.handle, cls, C.byref(ms), C.byref(n))
        out[cls] = ms.value * 1e3  # average per launch, us
    return out
Bs = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (16, 32, 50, 64, 96, 128, 200, 256, 320, 384, 512, 768)
# FFN class + out-projection, us per layer: "old" = the round-2 forms (sliced rows off, k_ffn_rows unfused off), "auto" =
# today's heuristics, "nw:S" = the sliced fused kernel forced (waves per workgroup : slices of the hidden dimension)
for B in Bs:
    row = []
    variants = [("old", {"rows_slices": -1, "ffn_rows_fuse": 0}), ("auto", {})]
    for nw in (12, 8):
        for S in (2, 3, 4, 5, 6, 8):
            if -(-B * L // (32 * nw)) * S <= 256:
                variants.append((f"{nw}:{S}", {"ffn_rows_nw": nw, "rows_slices": S}))
    for name, tunes in variants:
        lib.ffd_tune(b"reset", 0)
        for k, v in tunes.items():
            assert lib.ffd_tune(k.encode(), v) == 0, k
        t = run(B)
        row.append(f"{name} {t[N.K_FFN] + t[N.K_OUTPROJ]:.1f}")
    print(f"B={B} M={B*L}: " + "  ".join(row) + f"   attn {t[N.K_ATTN]:.1f}", flush=True)
