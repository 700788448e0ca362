"""Cell-step time of k_lstm_wave (one layer: no wavefront lag) and the lag per extra layer."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastfourierdiffusion_amd import _native as N
from fastfourierdiffusion_amd.models.score_models import LSTMScoreModule
from fastfourierdiffusion_amd.schedulers.sde import VPScheduler
from fastfourierdiffusion_amd.utils import synthetic
from fastfourierdiffusion_amd.utils.dataclasses import DiffusableBatch
dev = torch.device("cuda", 0)
lib = N.lib()
out = {}
L, C_, d = 251, 4, 72
for NL in (1, 2, 5, 10):
    sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True); sch.set_noise_scaling(L)
    sd = synthetic.lstm_state_dict(C_, L, d, NL, seed=1)
    m = LSTMScoreModule(n_channels=C_, max_len=L, noise_scheduler=sch, d_model=d, num_layers=NL)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True); m = m.to(dev).eval()
    for B in (16, 256):
        x = torch.randn(B, L, C_, device=dev); t = torch.full((B,), 0.4, device=dev)
        assert lib.ffd_tune(b"lstm_wave", 2) == 0
        for _ in range(3): m(DiffusableBatch(X=x, y=None, timesteps=t))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(DiffusableBatch(X=x, y=None, timesteps=t))
        torch.cuda.synchronize()
        out[f"NL{NL}_B{B}_us"] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
print(json.dumps(out))
