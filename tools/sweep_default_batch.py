"""The reference's default sample_batch_size (50, cmd/conf/sampler/default.yaml:3) and its neighbours: per-layer time of
the attention class and of the FFN class (+ out-projection) for every existing kernel form, forced through ffd_tune.
tools/sweep_default_batch.py [B,...]"""
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


def run(B, tunes):
    lib.ffd_tune(b"reset", 0)
    for k, v in tunes.items():
        if lib.ffd_tune(k.encode(), v) != 0:
            return None
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
        out[cls] = ms.value * 1e3 * n.value / (nst * NL)  # us per layer (a class may be several launches)
    # whole step, un-instrumented
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 50, float(sch.step_size), 0, 40, 1, 0, None, 0, 0, s), ctx.handle, "sample")
    e1.record(); torch.cuda.synchronize()
    out["step"] = e0.elapsed_time(e1) / 40
    return out


Bs = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (32, 50, 64, 100, 128)
for B in Bs:
    M = B * L
    att = [("auto", {}), ("hpw1", {"attn_hpw": 1}), ("split2", {"attn_small": 2}), ("split4", {"attn_small": 4})]
    row = []
    for name, t in att:
        r = run(B, t)
        row.append(f"{name} {r[N.K_ATTN]:.1f}" if r else f"{name} -")
    print(f"B={B} attention us/layer: " + "  ".join(row), flush=True)
    ffn = [("auto", {})]
    for wgs in (640, 1536, 3072, 6144):
        ffn.append((f"pair<= {wgs}", {"small_wgs": wgs}))
    ffn.append(("mid4", {"small_path": 0, "rows_slices": -1, "mid_path": 4}))
    for fuse in (1, 2):
        for nw in (12, 8):
            for S in (2, 3, 4, 6, 8, 11, 16):
                if -(-M // (32 * nw)) * S <= 1024:
                    ffn.append((f"{nw}:{S}:{'f' if fuse == 1 else 'u'}", {"small_path": 0, "ffn_rows_nw": nw, "rows_slices": S, "rows_slices_fuse": fuse}))
    row = []
    for name, t in ffn:
        r = run(B, t)
        row.append(f"{name} {r[N.K_FFN] + r[N.K_OUTPROJ]:.1f} ({r['step']:.3f})" if r else f"{name} -")
    print(f"B={B} M={M} FFN class + out-projection us/layer (ms/step): " + "  ".join(row), flush=True)
