"""Small-batch thresholds: ms/step of the ECG sampler for B x attn_small (0 = one workgroup per head pair, 2 / 4 = key
pieces of the split kernel) x small_path (the F-split out-proj + FFN pair)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
from fastfourierdiffusion_amd.sampling.sampler import DiffusionSampler

dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, "ecg")
lib = N.lib()
for B in (1, 2, 4, 8, 12, 16, 24, 32, 48):
    row = []
    for sp in (1, 0):
        for a in (0, 2, 4):
            lib.ffd_tune(b"attn_small", a); lib.ffd_tune(b"small_path", sp)
            s = DiffusionSampler(model, B, use_cache=False, rng="philox")
            s.sample(B, 10)
            torch.cuda.synchronize()
            t0 = time.time()
            s.sample(B, 200)
            torch.cuda.synchronize()
            row.append(f"sp{sp}/a{a} {(time.time() - t0) / 200 * 1e3:.3f}")
    print(f"B={B}: " + "  ".join(row), flush=True)
