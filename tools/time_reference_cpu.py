"""CONTAINER-ONLY measurement (needs /root/reference; never runs on the GPU box): the UNMODIFIED reference, imported
through oracle/_ref_import.py, timed in its own harness regime (cmd/benchmark_cache.py:42-112: sample_batch_size 1,
warm-up sample of 10 steps, then `num_samples` x `num_steps`, wall clock, cache off then cache on) next to the
oracle's restatement of the same call sequence, plus a B = 32 score-step timing of both.  Result:
profiles/r04_reference_cpu_timing.json -- the CPU ratio `speedup = t_no_cache / t_cache` (benchmark_cache.py:182)
that bench.py quotes as data in its `harness_b1` block.

    python tools/time_reference_cpu.py [--samples 10] [--steps 100] [--reps 3] [--threads N]
"""
import argparse, json, os, statistics, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle._ref_import import import_reference, reference_available
from oracle import ffd_oracle as O
from fastfourierdiffusion_amd.utils import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=10)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--only-b32", action="store_true", help="just the B = 32 step timings (seconds instead of minutes)")
ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_reference_cpu_timing.json"))
args = ap.parse_args()
assert reference_available(), "the reference tree is only present in the build container"
torch.set_num_threads(args.threads)
ns = import_reference()

C_, L, d, NL, H = 1, 187, 72, 10, 12  # ECG, default transformer (BASELINE configs[1]/[2])
sd = {k: torch.from_numpy(v.copy()) for k, v in synthetic.transformer_state_dict(C_, L, d, NL, seed=42).items()}


def ref_model():
    sch = ns.VPScheduler(fourier_noise_scaling=True, beta_min=0.1, beta_max=20.0)
    sch.set_noise_scaling(L)
    m = ns.ScoreModule(n_channels=C_, max_len=L, noise_scheduler=sch, fourier_noise_scaling=True, d_model=d,
                       num_layers=NL, n_head=H)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    m.eval()
    with torch.no_grad():
        for _ in range(4):
            m.pos_encoder(torch.zeros(1, L, d))
    return m


def ref_harness(use_cache):
    """benchmark_sampling of the reference, verbatim call sequence (cmd/benchmark_cache.py:71-99)."""
    m = ref_model()
    sampler = ns.DiffusionSampler(score_model=m, sample_batch_size=1, use_cache=use_cache, cache_kwargs={} if use_cache else None)
    if use_cache and m.cache is not None:
        m.cache.reset()
    sampler.sample(num_samples=1, num_diffusion_steps=10)
    if use_cache and m.cache is not None:
        m.cache.reset()
    t0 = time.time()
    sampler.sample(num_samples=args.samples, num_diffusion_steps=args.steps)
    return time.time() - t0


def oracle_harness(use_cache):
    def run(n, steps, seed):
        noise = (torch.from_numpy(z) for z in synthetic.noise_stream((1, L, C_), n * (steps + 1), seed))
        return O.sample(sd, kind="transformer", n_channels=C_, max_len=L, num_layers=NL, n_head=H, sde="vp",
                        sde_kwargs={"beta_min": 0.1, "beta_max": 20.0}, fourier_noise_scaling=True, num_samples=n,
                        batch_size=1, num_steps=steps, noise=noise, use_cache=use_cache)
    run(1, 10, 1)
    t0 = time.time()
    run(args.samples, args.steps, 2)
    return time.time() - t0


def step_time_b32(which):
    """One score evaluation + VP step at B = 32 (bench.py's cpu_baseline shape), ms."""
    B = 32
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, L, C_, generator=g)
    if which == "reference":
        m = ref_model()
        sch = m.noise_scheduler
        sch.set_timesteps(1000)
        f = lambda x, tv: sch.step(m(ns.DiffusableBatch(X=x, y=None, timesteps=torch.full((B,), tv))), tv, x).prev_sample
    else:
        G = O.noise_scaling(L, True)
        ts, dt = O.timesteps(1000)
        fwd = O.score_forward_stock if which == "oracle_stock" else O.score_forward
        f = lambda x, tv: O.vp_step(x, fwd(x, torch.full((B,), tv), sd, NL, H), torch.randn(B, L, C_, generator=g), tv, G, dt)
    with torch.no_grad():
        for _ in range(2):
            f(x, 0.9)
        reps = []
        for _ in range(3):
            t0 = time.time()
            n = 10
            for i in range(n):
                x2 = f(x, 0.9 - 0.001 * i)
            reps.append((time.time() - t0) / n * 1e3)
        return statistics.median(reps)


res = {"what": "unmodified reference (imported via oracle/_ref_import.py) and the oracle restatement on this container's CPU",
       "config": "ECG L=187 C=1, transformer d72/H12/NL10/F2048, VP, random-init weights seed 42",
       "threads": args.threads, "torch": torch.__version__, "num_samples": args.samples, "num_diffusion_steps": args.steps,
       "reps": args.reps, "harness_b1": {}}
with torch.no_grad():
    for name, fn in (() if args.only_b32 else (("reference", ref_harness), ("oracle", oracle_harness))):
        offs, ons = [], []
        for r in range(args.reps):
            offs.append(fn(False))
            ons.append(fn(True))
            print(f"{name} rep {r}: off {offs[-1]:.2f} s  on {ons[-1]:.2f} s  ratio {offs[-1] / ons[-1]:.3f}", flush=True)
        ratios = [a / b for a, b in zip(offs, ons)]
        res["harness_b1"][name] = {
            "seconds_off": offs, "seconds_on": ons, "off_over_on_per_rep": ratios,
            "off_over_on_median": statistics.median(ratios),
            "ms_per_step_off_median": statistics.median(offs) / (args.samples * args.steps) * 1e3,
            "ms_per_step_on_median": statistics.median(ons) / (args.samples * args.steps) * 1e3}
    res["step_b32_ms"] = {"reference": step_time_b32("reference"), "oracle": step_time_b32("oracle"),
                          "oracle_stock": step_time_b32("oracle_stock")}
    res["step_b32_ms"]["reference_again"] = step_time_b32("reference")
    res["oracle_stock_over_reference_b32"] = res["step_b32_ms"]["oracle_stock"] / (
        0.5 * (res["step_b32_ms"]["reference"] + res["step_b32_ms"]["reference_again"]))
    res["step_b32_samples_per_s_at_1000_steps"] = {k: 32.0 / v for k, v in res["step_b32_ms"].items()}
print(json.dumps(res, indent=1))
json.dump(res, open(args.out, "w"), indent=1)
