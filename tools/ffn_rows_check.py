"""Row-owning FFN (ffd_tune "ffn_rows") against k_ffn_ln on the ECG score at B = 512 / 513 (ragged last tile),
against the oracle on a slice, and both kernels' launch times (ffd_bench_ffn)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
from fastfourierdiffusion_amd.utils import synthetic
from fastfourierdiffusion_amd.utils.dataclasses import DiffusableBatch
from oracle import ffd_oracle as O

dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, sys.argv[1] if len(sys.argv) > 1 else "ecg")
ctx = model._ctx(); lib = ctx.lib
L, Cn = model.max_len, model.n_channels
for kv in os.environ.get("FFD_TUNE", "").split(","):
    if kv:
        assert lib.ffd_tune(kv.split("=")[0].encode(), int(kv.split("=")[1])) == 0, kv
out = {}
for B in (512, 513):
    x = torch.from_numpy(next(synthetic.noise_stream((B, L, Cn), 1, 77))).to(dev)
    t = torch.full((B,), 0.6, device=dev)
    res = {}
    for rows in (0, 1):
        assert lib.ffd_tune(b"ffn_rows", rows) == 0
        res[rows] = model(DiffusableBatch(X=x, y=None, timesteps=t)).cpu()
    rel = float((res[1] - res[0]).abs().max() / res[0].abs().max())
    out[f"B{B}_rows_vs_ffn_ln_rel"] = rel
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    sl = slice(B - 3, B)
    ref = O.score_forward(x[sl].cpu(), torch.full((3,), 0.6), sdt, 10, 12)
    for rows in (0, 1):
        out[f"B{B}_rows{rows}_vs_oracle_rel"] = float((res[rows][sl] - ref).abs().max() / ref.abs().max())
    out[f"B{B}_finite"] = bool(torch.isfinite(res[1]).all())
for rows in (0, 1):
    assert lib.ffd_tune(b"ffn_rows", rows) == 0
    ms = C.c_float()
    N.check(lib.ffd_bench_ffn(ctx.handle, 512, 50, C.byref(ms), None), ctx.handle)
    fl = lib.ffd_ffn_flops_per_launch(ctx.handle, 512)
    out[f"rows{rows}_us"] = ms.value * 1e3
    out[f"rows{rows}_tflops"] = fl / ms.value / 1e9
print(json.dumps(out, indent=1))
