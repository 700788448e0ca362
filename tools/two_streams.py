"""Experiment: one B=512 batch on one stream vs two B=256 half-batches on two streams
(MFMA-bound FFN of one half overlapping the VALU-bound attention of the other)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fastfourierdiffusion_amd import _native as N
from fastfourierdiffusion_amd.sampling.sampler import DiffusionSampler
dev = torch.device("cuda", 0)
K = 100
def setup(B):
    model, sch, sd = bench.build_model(dev, "ecg")
    sch.set_timesteps(1000)
    ts_c = (C.c_float * 1000)(*sch.timesteps.tolist())
    s = DiffusionSampler(model, B, rng="philox", seed=1)
    X = s.sample_prior(B)
    return model, s, X, ts_c, float(sch.step_size)
def run(model, s, X, ts_c, dt, first, n, stream):
    bench.run_steps(model, X, ts_c, 1000, dt, first, n, False, stream, 0)
full = setup(512)
run(*full, 0, 10, N.current_stream_ptr(dev)); torch.cuda.synchronize()
t0 = time.perf_counter(); run(*full, 10, K, N.current_stream_ptr(dev)); torch.cuda.synchronize()
t_full = (time.perf_counter() - t0) / K
a, b = setup(256), setup(256)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
lib = N.lib()
for tag, tunes in (("default kernels", {}), ("unsliced FFN", {"rows_slices": -1})):
    lib.ffd_tune(b"reset", 0)
    for k, v in tunes.items():
        assert lib.ffd_tune(k.encode(), v) == 0
    for st, h in ((s1, a), (s2, b)):
        run(*h, 0, 10, st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # interleave enqueues in chunks so that neither stream's queue starves
    for c in range(0, K, 5):
        run(*a, 10 + c, 5, s1.cuda_stream)
        run(*b, 10 + c, 5, s2.cuda_stream)
    torch.cuda.synchronize()
    t_two = (time.perf_counter() - t0) / K
    t0 = time.perf_counter(); run(*a, 10, K, s1.cuda_stream); torch.cuda.synchronize()
    t_half = (time.perf_counter() - t0) / K
    print(f"{tag}: B=512 one stream: {t_full*1e3:.3f} ms/step | 2 x B=256 two streams: {t_two*1e3:.3f} ms/step | B=256 alone: {t_half*1e3:.3f} ms/step", flush=True)
lib.ffd_tune(b"reset", 0)
