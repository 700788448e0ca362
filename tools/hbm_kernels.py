"""HBM-bound kernels of the sampling path at the config-5 shard size (B x 512 x 8), for
rocprofv3 --kernel-trace --stats:

    rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 tools/hbm_kernels.py [B] [iters]

Launches (through the C ABI) dft, idft, the SDE step (Philox and injected z), the prior draw,
FreSca (spatial and energy) and -- through one score evaluation at B/4 -- embed / unembed.
tools/hbm_table.py turns the kernel_stats.csv into GB/s per kernel against the algorithmic bytes
printed here as JSON on stdout."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from fastfourierdiffusion_amd import _native as N  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L, Cn = 512, 8
dev = torch.device("cuda", 0)
lib = N.lib()
s = N.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(1)
# NP buffer pairs in rotation: one launch's working set is 2-3 x 134 MB at the default size, so with a single pair a
# good part of every read came out of the 256 MiB Infinity Cache; four pairs (> 1 GB) are all evicted before reuse
NP = int(os.environ.get("HBM_PAIRS", "4"))
xs = [torch.randn(B, L, Cn, device=dev, generator=g) for _ in range(NP)]
ys = [torch.empty_like(xs[0]) for _ in range(NP)]
Gh = (C.c_float * L)()
lib.ffd_host_noise_scaling(L, 1, Gh)
Gd = torch.tensor(list(Gh), device=dev)
sde = N.SdeDesc(0, 0, 0.1, 20.0)
works = [torch.empty(B * Cn * (L // 2 + 1) + 4, device=dev) for _ in range(NP)]
n = B * L * Cn
for i in range(iters * NP):
    x, y = xs[i % NP], ys[i % NP]
    assert lib.ffd_dft(x.data_ptr(), y.data_ptr(), B, L, Cn, s) == 0
for i in range(iters * NP):
    x, y = xs[i % NP], ys[i % NP]
    assert lib.ffd_idft(y.data_ptr(), x.data_ptr(), B, L, Cn, s) == 0
for i in range(iters * NP):
    x, y = xs[i % NP], ys[i % NP]
    assert lib.ffd_sde_step(C.byref(sde), x.data_ptr(), y.data_ptr(), Gd.data_ptr(), 0.5, 1e-3, None, 42, 0, i, B, L, Cn, s) == 0
for i in range(iters * NP):
    assert lib.ffd_prior(C.byref(sde), xs[i % NP].data_ptr(), None, Gd.data_ptr(), 42, 0, B, L, Cn, s) == 0
for mode in (0, 1):
    for i in range(iters * NP):
        x, y, work = xs[i % NP], ys[i % NP], works[i % NP]
        assert lib.ffd_fresca(y.data_ptr(), x.data_ptr(), work.data_ptr(), B, L, Cn, 1.0, 1.5, 0.5, mode, s) == 0
x = xs[0]
torch.cuda.synchronize()

# embed / unembed / out-proj+LN1 through one score evaluation (and the sampling loop's fused tail)
import bench  # noqa: E402

Bs = max(1, B // 4)
model, sch, _ = bench.build_model(dev, "syn512")
ctx = model._ctx()
xs = x[:Bs].contiguous()
sc = torch.empty_like(xs)
for _ in range(2):
    N.check(lib.ffd_score_forward(ctx.handle, xs.data_ptr(), 0.5, sc.data_ptr(), Bs, s), ctx.handle, "fwd")
sch.set_timesteps(1000)
ts_c = (C.c_float * 1000)(*sch.timesteps.tolist())
N.check(lib.ffd_sample_batch(ctx.handle, xs.data_ptr(), Bs, ts_c, 1000, float(sch.step_size), 0, 3, 42, 0, None, 0, 0, s),
        ctx.handle, "sample")
torch.cuda.synchronize()
d = 72
print(json.dumps({"B": B, "L": L, "C": Cn, "B_score": Bs, "algorithmic_bytes": {
    "k_fft<false>": 8 * n, "k_fft<true>": 8 * n, "k_rfft_pow2": 8 * n,
    "k_sde_step": 12 * n, "k_prior": 4 * n,
    "k_fresca_apply": 8 * n, "k_fresca_spectrum": 4 * n,
    "k_embed": 4 * Bs * L * (Cn + d), "k_unembed_mfma<72, false>": 4 * Bs * L * (Cn + d), "k_unembed(": 4 * Bs * L * (Cn + d),
    "k_unembed_mfma<72, true>": 4 * Bs * L * (d + 2 * Cn),
    "k_linear_res_ln": 4 * Bs * L * 3 * d}}))
