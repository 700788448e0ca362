"""cmd/benchmark_cache.py regime: sample_batch_size=1, cache off vs on (reference ratio 1.05-1.12)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd.sampling.sampler import DiffusionSampler
dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, "ecg")
def run(use_cache, B, n_samples, steps, rng):
    s = DiffusionSampler(model, B, use_cache=use_cache, cache_kwargs={}, rng=rng)
    if use_cache and model.cache is not None:
        model.cache.reset()
    s.sample(B, 10)  # warm-up like benchmark_cache.py:85
    torch.cuda.synchronize()
    t0 = time.time()
    s.sample(n_samples, steps)
    torch.cuda.synchronize()
    return time.time() - t0
for B, n, steps in ((1, 4, 100), (8, 8, 100), (32, 32, 100)):
    for rng in ("philox", "torch"):
        off = run(False, B, n, steps, rng)
        on = run(True, B, n, steps, rng)
        model.disable_caching()
        print(f"B={B} rng={rng}: no-cache {off/ (n//B) / steps*1e3:.3f} ms/step  cache {on/(n//B)/steps*1e3:.3f} ms/step  ratio {off/on:.3f}")
