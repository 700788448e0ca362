"""LSTM layer wavefront (ffd_tune "lstm_wave") against the per-layer kernels and the oracle, and ms per score
evaluation over a batch sweep.  tools/lstm_wave_check.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
from fastfourierdiffusion_amd.utils import synthetic
from fastfourierdiffusion_amd.utils.dataclasses import DiffusableBatch
from oracle import ffd_oracle as O
dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, "nasa_lstm")
ctx = model._ctx(); lib = ctx.lib
L, Cn = model.max_len, model.n_channels
sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
out = {}
for B in (3, 37, 512):
    x = torch.from_numpy(next(synthetic.noise_stream((B, L, Cn), 1, 5))).to(dev)
    t = torch.full((B,), 0.45, device=dev)
    res = {}
    for w in (0, 1):
        assert lib.ffd_tune(b"lstm_wave", w) == 0
        res[w] = model(DiffusableBatch(X=x, y=None, timesteps=t)).cpu()
    ref = O.lstm_score_forward(x[:2].cpu(), torch.full((2,), 0.45), sdt, 10)
    out[f"B{B}"] = {"wave_vs_layers": float((res[1] - res[0]).abs().max() / res[0].abs().max()),
                    "wave_vs_oracle": float((res[1][:2] - ref).abs().max() / ref.abs().max()),
                    "layers_vs_oracle": float((res[0][:2] - ref).abs().max() / ref.abs().max())}
times = {}
for B in (1, 16, 64, 128, 256, 512, 1024, 1536):
    x = torch.randn(B, L, Cn, device=dev); t = torch.full((B,), 0.45, device=dev)
    row = {}
    for w in (0, 1):
        assert lib.ffd_tune(b"lstm_wave", w) == 0
        for _ in range(3): model(DiffusableBatch(X=x, y=None, timesteps=t))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): model(DiffusableBatch(X=x, y=None, timesteps=t))
        torch.cuda.synchronize(); row[w] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
    times[B] = row
out["ms_per_forward_layers_vs_wave"] = times
print(json.dumps(out))
