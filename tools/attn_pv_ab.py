"""P.V placement of the two-head attention kernel (ffd_tune "attn_pv": 0 VALU, 1 head dims 0..3 on the 4x4x1 MFMA, 2 all):
score agreement with the VALU form and with the oracle, and the kernel's launch time (HIP-event pairs in situ),
interleaved rounds in one process.  tools/attn_pv_ab.py [workload] [B]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
from fastfourierdiffusion_amd.utils import synthetic
from fastfourierdiffusion_amd.utils.dataclasses import DiffusableBatch
from oracle import ffd_oracle as O
wl = sys.argv[1] if len(sys.argv) > 1 else "ecg"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, wl)
ctx = model._ctx(); lib = ctx.lib
L, Cn = model.max_len, model.n_channels
x = torch.from_numpy(next(synthetic.noise_stream((B, L, Cn), 1, 77))).to(dev)
t = torch.full((B,), 0.6, device=dev)
out = {}
res = {}
for pv in (0, 1, 2, 3):
    assert lib.ffd_tune(b"attn_pv", pv) == 0
    res[pv] = model(DiffusableBatch(X=x, y=None, timesteps=t)).cpu()
sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
ref = O.score_forward(x[:3].cpu(), torch.full((3,), 0.6), sdt, 10, 12)
for pv in (0, 1, 2, 3):
    out[f"pv{pv}_vs_oracle"] = float((res[pv][:3] - ref).abs().max() / ref.abs().max())
    out[f"pv{pv}_vs_pv0"] = float((res[pv] - res[0]).abs().max() / res[0].abs().max())
sch.set_timesteps(1000)
ts_c = (C.c_float * 1000)(*sch.timesteps.tolist())
s = N.current_stream_ptr(dev)
X = torch.randn(B, L, Cn, device=dev)
times = {0: [], 1: [], 2: [], 3: []}
for rnd in range(3):
    for pv in (0, 1, 2, 3):
        assert lib.ffd_tune(b"attn_pv", pv) == 0
        N.check(lib.ffd_kernel_timing_begin(ctx.handle, 1 << N.K_ATTN, 10 * 10), ctx.handle)
        bench.run_steps(model, X, ts_c, 1000, float(sch.step_size), 100, 10, False, s, 0)
        N.check(lib.ffd_kernel_timing_end(ctx.handle), ctx.handle)
        ms, n = C.c_float(), C.c_int()
        N.check(lib.ffd_kernel_timing_get(ctx.handle, N.K_ATTN, C.byref(ms), C.byref(n)), ctx.handle)
        times[pv].append(round(ms.value * 1e3, 1))
out["attn_us_per_launch"] = times
print(json.dumps(out))
