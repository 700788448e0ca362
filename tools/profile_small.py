"""One sampler at a small batch for rocprofv3: tools/profile_small.py <B> <steps> [cache]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd.sampling.sampler import DiffusionSampler

B, steps = int(sys.argv[1]), int(sys.argv[2])
use_cache = len(sys.argv) > 3 and sys.argv[3] == "cache"
dev = torch.device("cuda", 0)
model, sch, sd = bench.build_model(dev, "ecg")
s = DiffusionSampler(model, B, use_cache=use_cache, cache_kwargs={})
s.sample(B, 10)
torch.cuda.synchronize()
s.sample(B, steps)
torch.cuda.synchronize()
