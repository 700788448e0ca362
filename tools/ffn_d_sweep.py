"""Launch time and matrix-pipe fraction of the fused FFN kernel (out-projection + LN1 + FFN + LN2) per d_model at L = 187,
B = 512: ffd_bench_ffn (HIP events around back-to-back launches of layer 0's kernel as the forward pass selects it).
tools/ffn_d_sweep.py [B]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastfourierdiffusion_amd import _native as N
from fastfourierdiffusion_amd.models.score_models import ScoreModule
from fastfourierdiffusion_amd.schedulers.sde import VPScheduler
from fastfourierdiffusion_amd.utils import synthetic
from fastfourierdiffusion_amd.utils.dataclasses import DiffusableBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
rows = []
for d, H in ((72, 12), (64, 8), (60, 12), (48, 12)):
    L, Cn, NL = 187, 1, 2
    sch = VPScheduler(fourier_noise_scaling=True)
    sch.set_noise_scaling(L)
    m = ScoreModule(n_channels=Cn, max_len=L, noise_scheduler=sch, d_model=d, num_layers=NL, n_head=H)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.transformer_state_dict(Cn, L, d, NL, seed=1).items()}, strict=True)
    m = m.to(dev).eval()
    m(DiffusableBatch(X=torch.randn(2, L, Cn, device=dev), y=None, timesteps=torch.full((2,), 0.5, device=dev)))
    ctx = m._ctx()
    s = N.current_stream_ptr(dev)
    for knob in ({}, {"ffn_rows": 0}):  # (ffn_rows = 0: k_ffn_ln alone, WITHOUT the out-projection + LN1 launch it needs in front)
        for k, v in knob.items():
            assert ctx.lib.ffd_tune(k.encode(), v) == 0
        ms = C.c_float()
        N.check(ctx.lib.ffd_bench_ffn(ctx.handle, B, 3000, C.byref(ms), s), ctx.handle, "bench_ffn")  # > 1 s: the clock under this load
        N.check(ctx.lib.ffd_bench_ffn(ctx.handle, B, 300, C.byref(ms), s), ctx.handle, "bench_ffn")
        fl, by = C.c_double(), C.c_double()
        name = ctx.lib.ffd_kernel_work(ctx.handle, N.K_FFN, B, 0, C.byref(fl), C.byref(by)).decode()
        flops = 4.0 * B * L * d * 2048 + (2.0 * B * L * d * d if not knob else 0.0)
        rows.append({"d_model": d, "knobs": knob, "form": "k_ffn_rows<oproj> (unsliced)" if not knob else "k_ffn_ln", "us_per_launch": round(ms.value * 1e3, 1),
                     "tflops": round(flops / (ms.value * 1e-3) / 1e12, 1), "frac_of_157.3": round(flops / (ms.value * 1e-3) / 157.3e12, 3)})
        print(json.dumps(rows[-1]), flush=True)
        assert ctx.lib.ffd_tune(b"reset", 0) == 0
