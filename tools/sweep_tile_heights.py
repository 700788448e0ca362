"""k_ffn_ln at 16 MB rows per workgroup, MB = 1 ... 4 (the four waves split F): FFN class + out-projection per layer
against the forms the heuristics pick.  tools/sweep_tile_heights.py [B,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]] + sys.argv[1:]
import ctypes as C
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


def run(B, tunes):
    lib.ffd_tune(b"reset", 0)
    for k, v in tunes.items():
        assert lib.ffd_tune(k.encode(), v) == 0, (k, v)
    x = torch.randn(B, L, 1, device=dev)
    s = N.current_stream_ptr(dev)
    nst = 8
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, 3, 1, 0, None, 0, 0, s), ctx.handle, "warm")
    N.check(lib.ffd_kernel_timing_begin(ctx.handle, 0xFF, nst * (6 * NL + 4)), ctx.handle, "begin")
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, nst, 1, 0, None, 0, 0, s), ctx.handle, "sample")
    N.check(lib.ffd_kernel_timing_end(ctx.handle), ctx.handle, "end")
    out = {}
    for cls in (N.K_FFN, N.K_OUTPROJ, N.K_ATTN):
        ms, n = C.c_float(), C.c_int()
        lib.ffd_kernel_timing_get(ctx.handle, cls, C.byref(ms), C.byref(n))
        out[cls] = ms.value * 1e3 * n.value / (nst * NL)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, 40, 1, 0, None, 0, 0, s), ctx.handle, "sample")
    e1.record(); torch.cuda.synchronize()
    out["step"] = e0.elapsed_time(e1) / 40
    return out


Bs = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (32, 50, 64, 80, 100, 128, 160, 200)
for B in Bs:
    row = []
    r = run(B, {})
    row.append(f"auto {r[N.K_FFN] + r[N.K_OUTPROJ]:.1f} ({r['step']:.3f})")
    r = run(B, {"ffn_height": 2})
    row.append(f"height-unfused {r[N.K_FFN] + r[N.K_OUTPROJ]:.1f} ({r['step']:.3f})")
    r = run(B, {"ffn_height": 0})
    row.append(f"height-off {r[N.K_FFN] + r[N.K_OUTPROJ]:.1f} ({r['step']:.3f})")
    for mb in (1, 2, 3, 4):
        r = run(B, {"small_path": 0, "rows_slices": -1, "mid_path": 0, "ffn_rows": 0, "ffn_mb": mb})
        row.append(f"mb{mb} {r[N.K_FFN]:.1f}+{r[N.K_OUTPROJ]:.1f} ({r['step']:.3f})")
    print(f"B={B} M={B * L} tiles16={-(-B * L // 16)}: " + "  ".join(row), flush=True)
