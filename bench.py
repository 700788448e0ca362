#!/usr/bin/env python3
"""Headline benchmark: samples/sec at 1000 diffusion steps, ECG frequency-domain
(BASELINE.json configs[1]) on N MI355X.

A "step" is one reverse-diffusion step (score evaluation + Euler-Maruyama update) of
one batch of B=512 synthetic series per GPU, enqueued through libffd's
``ffd_sample_batch``.  K steps are timed between barrier + synchronize pairs; the
metric is  value = N_gpus * B / (1000 * seconds_per_step)  -- the rate at which
complete 1000-step samples leave the node (the prior draw and the final idft are two
more launches per 1000 steps, < 0.01 % of the time).  With K = 1000 the
timed region *is* one complete sampling of the batch.

Sampling is embarrassingly parallel over the batch: ranks are independent shards
(weights replicated, Philox noise keyed by global sample index), no data-path
collective; the only torch.distributed traffic is the barrier, the max-over-ranks of
the elapsed time and the gather of the per-rank times.

Launching: ``python bench.py --gpus N`` starts its own N ranks (one fresh child process
per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set per child; the parent never
touches the GPU and relays rank 0's JSON line); under ``python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N`` the ranks torchrun started are used as they are.

Other workloads (one JSON line each, same contract):
  --workload syn512 --batch 8192 --cache --steps 3 --warmup 1   BASELINE configs[4] per-GPU shard
  --workload nasa_lstm --batch 512                              BASELINE configs[3]
  --cache                                                       BASELINE configs[2] at the bench batch
  --ablation                                                    cmd/benchmark_cache.py's K/R/... grid at B=1

Rank 0 prints ONE JSON line (plus human-readable notes on stderr).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (the opt-in split FFN keeps 6 bf16 products per fp32 one)
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense fp32 matrix peak
def _ref_over_port():
    """Build-container measurement (tools/time_reference_cpu.py, committed): B = 32 step time of the CPU baseline's port
    (oracle with the stock nn.TransformerEncoder backbone) over the UNMODIFIED reference's -- how much faster the
    reference itself is than the port that is timed on this box."""
    try:
        rt = json.load(open(os.path.join(ROOT, "profiles", "r04_reference_cpu_timing.json")))["step_b32_ms"]
        return rt["oracle_stock"] / (0.5 * (rt["reference"] + rt["reference_again"]))
    except Exception:
        return None


PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)

WORKLOADS = {
    "ecg": "ECG frequency-domain L=187 C=1, transformer d72/H12/NL10/F2048, VP beta[0.1,20], 1000-step sampler "
           "(BASELINE configs[1])",
    "syn512": "synthetic L=512 C=8 transformer d72/H12/NL10/F2048 (BASELINE configs[4] per-GPU shard)",
    "nasa_lstm": "NASA-charge L=251 C=4 LSTM d72/NL10 (BASELINE configs[3])",
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=512, help="samples per GPU")
    ap.add_argument("--workload", default="ecg", choices=sorted(WORKLOADS))
    ap.add_argument("--cache", action="store_true", help="time the E2-CRF cached path (BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cache-ratio / harness side measurements")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the compact lines for the other BASELINE configs and the API-level end-to-end timing")
    ap.add_argument("--ablation", action="store_true",
                    help="also run the reference harness's ablation grid (cmd/benchmark_cache.py:274-422) at B=1")
    ap.add_argument("--tune", action="append", default=[], help="key=value for ffd_tune (experiments)")
    ap.add_argument("--ffn-split", action="store_true",
                    help="run with the opt-in bf16x3-split FFN (fp32-equivalent, NOT the reference's fp32 arithmetic): "
                         "its own line, dtype 'bf16x3-split, fp32 accumulate'")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a one-GPU box: every rank uses cuda:0 and the barrier / max-reduce "
                         "run over gloo (RCCL refuses two ranks on one device); the reported number is meaningless")
    ap.add_argument("--launch-plan", action="store_true",
                    help="print the rank plan the self-launcher would start (JSON) and exit; touches no GPU")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------
# self-launcher: `python bench.py --gpus N` without torchrun
# ---------------------------------------------------------------------------
def launch_plan(n_gpus: int, argv, port: int):
    """One entry per rank: the child's command line and the environment additions."""
    child_argv = [a for a in argv if a != "--launch-plan"]
    plan = []
    for r in range(n_gpus):
        plan.append({"rank": r, "cmd": [sys.executable, os.path.abspath(__file__)] + child_argv,
                     "env": {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_gpus),
                             "LOCAL_WORLD_SIZE": str(n_gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                             "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}})
    return plan


def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv) -> int:
    """Parent of an N-rank run.  It never imports torch or touches HIP; it starts N fresh children
    (no exec-replacement), relays rank 0's JSON line and fails if any child fails."""
    plan = launch_plan(args.gpus, argv, int(os.environ.get("MASTER_PORT", 0)) or free_port())
    if args.launch_plan:
        print(json.dumps({"n_gpus": args.gpus, "ranks": plan}))
        return 0
    if args.gpus > 1 and any(k in os.environ for k in ("ROCPROFILER_LIBRARY", "ROCP_TOOL_LIB", "ROCPROFV3_PRELOAD")) or \
            (args.gpus > 1 and "rocprof" in os.environ.get("LD_PRELOAD", "")):
        log("bench.py launcher: a profiler preload is active; profile with --gpus 1 (the preload initialises the GPU in "
            "this parent, which must stay a pure launcher)")
        return 2
    procs = []
    for p in plan:
        env = dict(os.environ)
        env.update(p["env"])
        procs.append(subprocess.Popen(p["cmd"], env=env, stdout=subprocess.PIPE if p["rank"] == 0 else subprocess.DEVNULL,
                                      text=True))
    # Poll every rank: the first failure ends the run (the others would sit in init_process_group / a barrier until
    # the c10d timeout); rank 0's stdout is drained by a thread so that a full pipe cannot block it.
    import threading

    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    failed = False
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if not failed and any(c not in (None, 0) for c in codes):
            failed = True
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            deadline = time.time() + 10.0
            while time.time() < deadline and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    reader.join(timeout=5.0)
    out0 = buf[0] if buf else ""
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if any(codes) or line is None:
        log(f"bench.py launcher: rank exit codes {codes}" + ("" if line else "; no JSON line from rank 0"))
        return max([c for c in codes if c] + [1])
    print(line, flush=True)
    return 0


# ---------------------------------------------------------------------------
# workload
# ---------------------------------------------------------------------------
def build_model(device, workload):
    import torch

    from fastfourierdiffusion_amd.models.score_models import LSTMScoreModule, ScoreModule
    from fastfourierdiffusion_amd.schedulers.sde import VPScheduler
    from fastfourierdiffusion_amd.utils import synthetic

    sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True)
    if workload == "nasa_lstm":
        C_, L, d, NL = 4, 251, 72, 10
        sd = synthetic.lstm_state_dict(C_, L, d, NL, seed=42)
        m = LSTMScoreModule(n_channels=C_, max_len=L, noise_scheduler=sch, d_model=d, num_layers=NL)
    else:
        C_, L = (1, 187) if workload == "ecg" else (8, 512)
        d, NL, H = 72, 10, 12
        sd = synthetic.transformer_state_dict(C_, L, d, NL, seed=42)
        m = ScoreModule(n_channels=C_, max_len=L, noise_scheduler=sch, d_model=d, num_layers=NL, n_head=H)
    sch.set_noise_scaling(L)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to(device).eval(), sch, sd


def run_steps(model, X, ts_c, n_total, step_size, first, n_run, use_cache, stream, offset):
    from fastfourierdiffusion_amd import _native as N

    ctx = model._ctx()
    rc = ctx.lib.ffd_sample_batch(ctx.handle, X.data_ptr(), X.shape[0], ts_c, n_total, step_size, first, n_run, 42,
                                  offset, None, int(use_cache), first if use_cache else 0, stream)
    N.check(rc, ctx.handle, "ffd_sample_batch")


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask, cgroup quota, and the GPU
    box's per-GPU share (16) -- os.cpu_count() reports the whole 256-thread host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("FFD_CPU_THREADS", "16"))))


def cpu_baseline(sd, L, Cn, NL, H, kind):
    """The oracle (CPU restatement of the reference path, torch-CPU fp32, all host cores)
    on a bounded sample of the same workload.  The transformer backbone is torch's stock nn.TransformerEncoder built as
    the reference builds it (score_models.py:61-66), so that the baseline runs on the fused encoder-layer path the
    reference's own CPU run takes and at its speed (profiles/r04_reference_cpu_timing.json: within a few per cent of
    the unmodified reference; the explicit restatement used for parity is ~1.5x slower)."""
    import torch

    from oracle import ffd_oracle as O

    cores = host_cores()
    torch.set_num_threads(cores)
    B, warm, steps = 32, 2, 20
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    G = O.noise_scaling(L, True)
    ts, dt = O.timesteps(1000)
    g = torch.Generator().manual_seed(0)
    x = O.prior(torch.randn(B, L, Cn, generator=g), G)
    t_acc = 0.0
    for i in range(warm + steps):
        t0 = time.perf_counter()
        tv = ts[i].item()
        t = torch.full((B,), tv)
        if kind == "lstm":
            score = O.lstm_score_forward_stock(x, t, sdt, NL)
        else:
            score = O.score_forward_stock(x, t, sdt, NL, H)
        x = O.vp_step(x, score, torch.randn(B, L, Cn, generator=g), tv, G, dt)
        if i >= warm:
            t_acc += time.perf_counter() - t0
    s_per_step = t_acc / steps
    return {"value": B / (1000.0 * s_per_step), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU fp32, stock {'nn.LSTM layers' if kind == 'lstm' else 'nn.TransformerEncoder backbone'} = the fused path the reference runs), B={B}, {steps} of 1000 steps timed after {warm} warm-up, x(1000/{steps})",
            "reference_speed_over_port": None if kind == "lstm" else _ref_over_port(),
            "reference_speed_source": "profiles/r04_reference_cpu_timing.json (unmodified reference vs this port, B=32 step, build container)",
            "ms_per_step": s_per_step * 1e3}


def cpu_harness_b1(sd, L, Cn, NL, H, num_samples, num_steps):
    """The reference harness regime on the CPU (oracle): B=1, cache off / on, same call sequence as
    cmd/benchmark_cache.py:42-112 (10-step warm-up sample, then the timed samples)."""
    import torch

    from fastfourierdiffusion_amd.utils import synthetic
    from oracle import ffd_oracle as O

    torch.set_num_threads(host_cores())
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    res = {}
    for uc in (False, True):
        def run(n, steps, seed):
            noise = (torch.from_numpy(z) for z in synthetic.noise_stream((1, L, Cn), n * (steps + 1), seed))
            return O.sample(sdt, kind="transformer", n_channels=Cn, max_len=L, num_layers=NL, n_head=H, sde="vp",
                            sde_kwargs={"beta_min": 0.1, "beta_max": 20.0}, fourier_noise_scaling=True, num_samples=n,
                            batch_size=1, num_steps=steps, noise=noise, use_cache=uc)
        run(1, 10, 1)
        t0 = time.perf_counter()
        run(num_samples, num_steps, 2)
        res[uc] = time.perf_counter() - t0
    return {"ms_per_step_off": res[False] / (num_samples * num_steps) * 1e3,
            "ms_per_step_on": res[True] / (num_samples * num_steps) * 1e3, "off_over_on": res[False] / res[True],
            "num_samples": num_samples, "num_diffusion_steps": num_steps, "cores": host_cores()}


def roofline_entry(lib, ctx, N, cls, B, cache_hit, traffic_tab, key):
    """One roofline object for kernel class `cls` from the in-situ HIP-event timing just collected."""
    ms, cnt = C.c_float(), C.c_int()
    N.check(lib.ffd_kernel_timing_get(ctx.handle, cls, C.byref(ms), C.byref(cnt)), ctx.handle, "ffd_kernel_timing_get")
    fl, by = C.c_double(), C.c_double()
    name = lib.ffd_kernel_work(ctx.handle, cls, B, int(cache_hit), C.byref(fl), C.byref(by))
    if not name or cnt.value == 0 or ms.value <= 0:
        return None
    name = name.decode()
    mfma = cls in (N.K_FFN, N.K_ATTN, N.K_LSTM_REC)
    sec = ms.value * 1e-3
    if mfma and name == "k_ffn_ln_split":  # fp32-equivalent FLOP against the bf16 dense peak / 6 kept terms
        ach, peak, unit = fl.value / sec / 1e12, PEAK_BF16_MFMA_TFLOPS / 6.0, "TFLOP/s"
    elif mfma:
        ach, peak, unit = fl.value / sec / 1e12, PEAK_FP32_MFMA_TFLOPS, "TFLOP/s"
    else:
        ach, peak, unit = by.value / sec / 1e9, PEAK_HBM_GBS, "GB/s"
    return {"kernel": name, "bound": "mfma" if mfma else "hbm", "achieved": ach, "peak": peak, "unit": unit,
            "frac": ach / peak, "traffic": traffic_tab.get(f"{key}:{name}", traffic_tab.get(f"{key}:{name.split('<')[0].split(' ')[0]}")),
            "flops_per_launch": fl.value, "algorithmic_bytes_per_launch": by.value, "ms_per_launch": ms.value,
            "launches_timed": cnt.value}


def measure_other(device, workload, B, steps, warmup, use_cache):
    """Compact line for a non-headline BASELINE config in the default run: value, ms/step and the dominant kernel's
    roofline fraction (HIP-event pairs on the launch stream), same method as the headline."""
    import torch

    from fastfourierdiffusion_amd import _native as N
    from fastfourierdiffusion_amd.sampling.sampler import DiffusionSampler

    model, sch, _ = build_model(device, workload)
    n_total = 1000
    sch.set_timesteps(n_total)
    ts_c = (C.c_float * n_total)(*sch.timesteps.tolist())
    step_size = float(sch.step_size)
    sampler = DiffusionSampler(model, B, use_cache=use_cache, cache_kwargs={}, rng="philox", seed=42)
    stream = N.current_stream_ptr(device)
    if use_cache:
        model.cache.reset()
    Xw = sampler.sample_prior(B)
    run_steps(model, Xw, ts_c, n_total, step_size, 0, warmup, use_cache, stream, 0)
    if use_cache:
        model.cache.reset()
    X = sampler.sample_prior(B)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run_steps(model, X, ts_c, n_total, step_size, 0, steps, use_cache, stream, 0)
    torch.cuda.synchronize(device)
    sec = (time.perf_counter() - t0) / steps
    assert torch.isfinite(X).all()
    ctx = model._ctx()
    lib = ctx.lib
    is_lstm = workload == "nasa_lstm"
    n_ev = min(steps, 3)
    N.check(lib.ffd_kernel_timing_begin(ctx.handle, 0xFF, n_ev * (3 * model.num_layers + 3)), ctx.handle, "timing_begin")
    run_steps(model, X, ts_c, n_total, step_size, steps, n_ev, use_cache, stream, 0)
    N.check(lib.ffd_kernel_timing_end(ctx.handle), ctx.handle, "timing_end")
    order = [N.K_LSTM_REC, N.K_LSTM_GATES] if is_lstm else [N.K_FFN, N.K_ATTN, N.K_OUTPROJ]
    lines = [r for r in (roofline_entry(lib, ctx, N, c, B, use_cache and not is_lstm, {}, "") for c in order) if r]
    res = {"workload": WORKLOADS[workload], "batch": B, "cache": use_cache, "steps_timed": steps, "warmup": warmup,
           "value": B / (1000.0 * sec), "unit": "samples/s", "ms_per_step": sec * 1e3,
           "roofline": {k: lines[0][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "ms_per_launch")} if lines else None,
           "roofline_kernels": [{k: r[k] for k in ("kernel", "bound", "frac", "ms_per_launch")} for r in lines[1:]]}
    del model, sampler, X, Xw
    torch.cuda.empty_cache()
    return res


def api_e2e(device, B, rng):
    """SURVEY 8(d)'s metric through the drop-in API: wall time of DiffusionSampler(model, B).sample(B, 1000) (prior
    draw, 1000 steps, the copy to the host the reference's sample() ends with) + idft, device-synchronised."""
    import torch

    from fastfourierdiffusion_amd.sampling.sampler import DiffusionSampler
    from fastfourierdiffusion_amd.utils.fourier import idft

    model, sch, _ = build_model(device, "ecg")
    sampler = DiffusionSampler(model, B, rng=rng, seed=42)
    sampler.sample(num_samples=B, num_diffusion_steps=20)  # warm-up (workspace, time-embedding table)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    x = sampler.sample(num_samples=B, num_diffusion_steps=1000)
    xt = idft(x)
    torch.cuda.synchronize(device)
    sec = time.perf_counter() - t0
    assert torch.isfinite(xt).all() and tuple(xt.shape) == (B, model.max_len, model.n_channels)
    return {"rng": rng, "seconds": sec, "samples_per_s": B / sec, "num_samples": B, "num_diffusion_steps": 1000,
            "includes": "prior draw + 1000 steps + host copy of the samples + idft"}


def main() -> None:
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.launch_plan):
        raise SystemExit(self_launch(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    import torch

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs; there is no CPU path"
    if args.rehearse_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from fastfourierdiffusion_amd import _native as N
    from fastfourierdiffusion_amd.sampling.sampler import DiffusionSampler
    from fastfourierdiffusion_amd.sharding import gather_seconds, reduce_max_seconds, shard_range

    for kv in args.tune:
        k, v = kv.split("=")
        assert N.lib().ffd_tune(k.encode(), int(v)) == 0, kv
    if args.ffn_split:
        assert N.lib().ffd_tune(b"ffn_split", 1) == 0
    model, sch, sd = build_model(device, args.workload)
    B, L, Cn = args.batch, model.max_len, model.n_channels
    n_total = 1000
    sch.set_timesteps(n_total)
    ts_c = (C.c_float * n_total)(*sch.timesteps.tolist())
    step_size = float(sch.step_size)
    use_cache = bool(args.cache)
    is_lstm = args.workload == "nasa_lstm"
    if use_cache and is_lstm:
        raise SystemExit("--cache: the LSTM backbone has no cache (SURVEY Q9)")
    sampler = DiffusionSampler(model, B, use_cache=use_cache, cache_kwargs={}, rng="philox", seed=42)
    stream = N.current_stream_ptr(device)
    offset, _ = shard_range(world * B, world, rank)  # global sample index of this shard's first sample (= rank * B)
    comm_dev = None if args.rehearse_one_gpu else device

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    K, W = args.steps, args.warmup
    if K > n_total or K < 1:
        raise SystemExit("--steps must be in [1, 1000] (1000 = one complete sampling)")
    if use_cache:
        model.cache.reset()
    # warm-up: W untimed steps on a throw-away prior draw
    if W > 0:
        Xw = sampler.sample_prior(B, _sample_offset=offset)
        run_steps(model, Xw, ts_c, n_total, step_size, 0, min(W, n_total), use_cache, stream, offset)
    # timed region: the first K steps of a real sampling of a fresh batch (for the cached
    # path this includes the table-filling step 0, as in the reference's benchmark)
    if use_cache:
        model.cache.reset()
    X = sampler.sample_prior(B, _sample_offset=offset)
    barrier()
    t0 = time.perf_counter()
    run_steps(model, X, ts_c, n_total, step_size, 0, K, use_cache, stream, offset)
    barrier()
    own = time.perf_counter() - t0
    elapsed = reduce_max_seconds(own, comm_dev)
    per_rank = gather_seconds(own, comm_dev)
    assert torch.isfinite(X).all(), "non-finite samples"
    ms_per_step = elapsed / K * 1e3
    value = world * B / (1000.0 * (elapsed / K))

    out = {
        "metric": "samples/sec at 1000 diffusion steps, ECG freq-domain" if args.workload == "ecg" else
                  f"samples/sec at 1000 diffusion steps, {args.workload}",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16x3-split, fp32 accumulate (FFN); f32 elsewhere" if args.ffn_split else "f32", "data": "synthetic",
        "config": {"workload": WORKLOADS[args.workload],
                   "batch_per_gpu": B, "global_batch": B * world, "diffusion_steps": n_total,
                   "cache": use_cache, "noise": "philox on device", "sharding": f"batch x{world}, no collectives"},
        "ms_per_step_per_rank": [t / K * 1e3 for t in per_rank],
    }

    if rank == 0 and not args.no_extras:
        ctx = model._ctx()
        lib = ctx.lib
        flops_step = lib.ffd_flops_per_sample_step(ctx.handle, 0) * B
        out["achieved_tflops_whole_step"] = flops_step / (ms_per_step * 1e-3) / 1e12
        # SURVEY 8(d) quotes the metric on a full sampling "incl. prior draw and final idft": time those two
        # launches (median of 5, device-synchronised) and report the rate with them added to 1000 steps.
        from fastfourierdiffusion_amd.utils.fourier import idft

        def med_ms(fn):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
                fn()
                torch.cuda.synchronize(device)
                ts.append((time.perf_counter() - t1) * 1e3)
            return sorted(ts)[2]

        prior_ms = med_ms(lambda: sampler.sample_prior(B, _sample_offset=offset))
        idft_ms = med_ms(lambda: idft(X))
        out["prior_ms"], out["idft_ms"] = prior_ms, idft_ms
        out["value_incl_prior_idft"] = world * B / (ms_per_step + (prior_ms + idft_ms) / 1000.0)

        # Per-kernel rooflines, timed in situ: HIP event pairs around every launch of 20 more sampling steps on
        # the launch stream (the quantity rocprofv3 --kernel-trace reports; profiles/r02_* hold those summaries).
        NLy = model.num_layers
        n_ev_steps = 20 if B * L <= 512 * 512 else 3
        N.check(lib.ffd_kernel_timing_begin(ctx.handle, 0xFF, n_ev_steps * (3 * NLy + 3)), ctx.handle,
                "ffd_kernel_timing_begin")
        run_steps(model, X, ts_c, n_total, step_size, 1, n_ev_steps, use_cache, stream, offset)
        N.check(lib.ffd_kernel_timing_end(ctx.handle), ctx.handle, "ffd_kernel_timing_end")
        tpath = os.path.join(ROOT, "profiles", "r04_traffic.json")
        if not os.path.exists(tpath):
            tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
        traffic_tab = json.load(open(tpath)) if os.path.exists(tpath) else {}
        key = f"{args.workload}:{B}" + (":cache" if use_cache else "")
        order = ([N.K_LSTM_REC, N.K_LSTM_GATES] if is_lstm else [N.K_FFN, N.K_ATTN, N.K_OUTPROJ]) + \
                [N.K_EMBED, N.K_UNEMBED, N.K_SDE]
        lines = [r for r in (roofline_entry(lib, ctx, N, c, B, use_cache, traffic_tab, key) for c in order) if r]
        if lines:
            out["roofline"] = lines[0]  # the dominant kernel of this workload
            out["roofline"]["traffic_note"] = ("HBM bytes per launch from the committed rocprofv3 PMC passes "
                                               "(FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), " + os.path.relpath(tpath, ROOT))
            out["roofline_kernels"] = lines[1:]
        if not is_lstm:
            iso = C.c_float()
            N.check(lib.ffd_bench_ffn(ctx.handle, B, 20, C.byref(iso), stream), ctx.handle, "ffd_bench_ffn")
            out["roofline"]["ms_per_launch_back_to_back"] = iso.value
        if not is_lstm and not use_cache and world == 1:
            # cache-on / cache-off ratio at the bench batch (BASELINE configs[2]), 200 steps each
            def timed(uc):
                s2 = DiffusionSampler(model, B, use_cache=uc, cache_kwargs={}, rng="philox", seed=42)
                if uc:
                    model._first_cache.reset()
                X2 = s2.sample_prior(B)
                nst = 200 if B * L <= 512 * 512 else 4
                run_steps(model, X2, ts_c, n_total, step_size, 0, 20 if nst == 200 else 1, uc, stream, 0)
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
                run_steps(model, X2, ts_c, n_total, step_size, 20, nst, uc, stream, 0)
                torch.cuda.synchronize(device)
                return (time.perf_counter() - t1) / nst

            t_off, t_on = timed(False), timed(True)
            model.disable_caching()
            out["cache_ratio"] = {"off_over_on": t_off / t_on, "ms_off": t_off * 1e3, "ms_on": t_on * 1e3,
                                  "batch": B, "note": "pure-cache steps (K/V projections skipped)"}
        if not is_lstm and not use_cache and world == 1 and not args.ffn_split:
            # Opt-in, reported next to the f32 line and never as `value`: the same steps with the FFN on the bf16
            # matrix cores as a three-part / six-term split (fp32-equivalent to ~2e-7, passes the same goldens, but not
            # the reference's fp32 FMA arithmetic).  python bench.py --ffn-split makes it the whole line.
            def timed_split(on):
                assert lib.ffd_tune(b"ffn_split", on) == 0
                X3 = sampler.sample_prior(B, _sample_offset=offset)
                nst = 100 if B * L <= 512 * 512 else 3
                run_steps(model, X3, ts_c, n_total, step_size, 0, 10 if nst == 100 else 1, False, stream, offset)
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
                run_steps(model, X3, ts_c, n_total, step_size, 10, nst, False, stream, offset)
                torch.cuda.synchronize(device)
                return (time.perf_counter() - t1) / nst

            try:
                t_split = timed_split(1)
                N.check(lib.ffd_kernel_timing_begin(ctx.handle, 1 << N.K_FFN, 3 * NLy), ctx.handle, "ffd_kernel_timing_begin")
                run_steps(model, X, ts_c, n_total, step_size, 1, 3, False, stream, offset)
                N.check(lib.ffd_kernel_timing_end(ctx.handle), ctx.handle, "ffd_kernel_timing_end")
                rl = roofline_entry(lib, ctx, N, N.K_FFN, B, False, {}, key)
            finally:
                lib.ffd_tune(b"ffn_split", 0)
            out["ffn_split_opt_in"] = {
                "dtype": "bf16x3-split, fp32 accumulate (FFN); f32 elsewhere", "value": B / (1000.0 * t_split),
                "unit": "samples/s", "ms_per_step": t_split * 1e3, "over_f32_line": (ms_per_step * 1e-3) / t_split,
                "roofline": rl, "note": "off by default (ffd_tune ffn_split); peak = 2.5 PFLOP/s dense bf16 / 6 kept terms"}
        if args.workload == "ecg" and world == 1:
            # The reference's own harness regime (cmd/benchmark_cache.py:42-112,159-186): sample_batch_size = 1,
            # 10 samples x 100 steps, speedup = t_no_cache / t_cache -- next to the oracle's CPU ratio at B=1.
            from fastfourierdiffusion_amd.benchmark import benchmark_sampling, run_cache_benchmark

            ns, nd = 10, 100
            # a fresh model, as in the reference's script: the harness's enable_caching is then the model's FIRST, the
            # one whose cache object the layers report to (later ones read zero hits)
            hmodel, _, _ = build_model(device, "ecg")
            # three alternating (off, on) repetitions (each with benchmark_sampling's own warm-up sample); the pair with
            # the median ratio is reported, the per-repetition ratios beside it.  The cache statistics are the first
            # cached run's: the layers report to the model's FIRST cache object.
            reps = []
            for _ in range(3):
                a = benchmark_sampling(hmodel, ns, nd, use_cache=False)
                c = benchmark_sampling(hmodel, ns, nd, use_cache=True, cache_kwargs={})
                reps.append((a["elapsed_time"] / c["elapsed_time"], a, c))
            first_stats = reps[0][2]["cache_stats"]
            reps.sort(key=lambda r: r[0])
            _, r_off, r_on = reps[1]
            r_on = dict(r_on, cache_stats=first_stats)
            hmodel.disable_caching()
            hb = {"sample_batch_size": 1, "num_samples": ns, "num_diffusion_steps": nd,
                  "off_over_on_per_rep": [r[0] for r in reps],
                  "ms_per_step_off": r_off["elapsed_time"] / (ns * nd) * 1e3,
                  "ms_per_step_on": r_on["elapsed_time"] / (ns * nd) * 1e3,
                  "samples_per_s_off": ns / r_off["elapsed_time"], "samples_per_s_on": ns / r_on["elapsed_time"],
                  "off_over_on": r_off["elapsed_time"] / r_on["elapsed_time"],
                  "cache_hit_ratio": r_on["cache_stats"].get("cache_hit_ratio"),
                  "reference_published": "17.70 s / 15.78 s for 20 samples x 100 steps on Apple mps = 1.13 samples/s, "
                                         "ratio 1.12 (BASELINE.md section 1)"}
            # The contract ("cache-on / cache-off ratio within 5 % of the reference's") is read against the reference's
            # published ACCELERATOR ratio (mps, 1.12): like for like.  The reference's CPU ratio is data beside it: the
            # unmodified reference timed in the build container (tools/time_reference_cpu.py, committed) is < 1 there.
            hb["contract"] = {"reference_ratio": 1.12, "reference_ratio_source": "notebooks/ablation_cache_test.ipynb:277-278 "
                              "(15.78 s vs 17.70 s, Apple mps, B=1, 20 x 100 steps)", "gpu_ratio": hb["off_over_on"],
                              "gpu_over_reference": hb["off_over_on"] / 1.12,
                              "within_5_percent": abs(hb["off_over_on"] / 1.12 - 1.0) <= 0.05,
                              "why_not_the_same_box_cpu_ratio": "BASELINE.md reads 'the reference's measured ratio on the same "
                              "box'; the reference cannot run on this box's GPU (CUDA / mps only) and on a CPU its cache is a "
                              "SLOW-DOWN (0.84 median in the build container, 0.90 for the oracle on this box's cores: the "
                              "explicit cached layers lose torch's fused encoder path), so a device ratio > 1 can only be "
                              "compared with the reference's own device ratio, the published 1.12; the CPU ratios are "
                              "reported beside it (reference_cpu_timing, cpu_oracle), 20-30 % away by construction"}
            rpath = os.path.join(ROOT, "profiles", "r04_reference_cpu_timing.json")
            if not os.path.exists(rpath):
                rpath = os.path.join(ROOT, "profiles", "r03_reference_cpu_timing.json")
            if os.path.exists(rpath):
                rt = json.load(open(rpath))
                hb["reference_cpu_timing"] = {
                    "file": "profiles/" + os.path.basename(rpath), "threads": rt["threads"],
                    "num_samples": rt["num_samples"], "num_diffusion_steps": rt["num_diffusion_steps"],
                    "reference_off_over_on_per_rep": rt["harness_b1"]["reference"]["off_over_on_per_rep"],
                    "reference_off_over_on_median": rt["harness_b1"]["reference"]["off_over_on_median"],
                    "oracle_off_over_on_median": rt["harness_b1"]["oracle"]["off_over_on_median"],
                    "reference_ms_per_step_off": rt["harness_b1"]["reference"]["ms_per_step_off_median"],
                    "note": "the unmodified reference, build container CPU (not this box); measured, not a target"}
            if not args.no_cpu_baseline:
                hb["cpu_oracle"] = cpu_harness_b1(sd, L, Cn, NLy, model.n_head, ns, nd)
            out["harness_b1"] = hb
            if args.ablation:
                gmodel, _, _ = build_model(device, "ecg")
                rows = run_cache_benchmark(gmodel, num_samples=ns, num_diffusion_steps=nd)
                gmodel.disable_caching()
                out["ablation"] = [{k: v for k, v in r.items()} for r in rows]
        if args.workload == "ecg" and world == 1 and not use_cache and not args.ffn_split and not args.no_other_workloads:
            # the other BASELINE configs, so that the driver's record carries them (configs[4] per-GPU shard, configs[3]
            # at the large-batch and at SURVEY's stated batch) and the metric through the Python API
            del X
            torch.cuda.empty_cache()
            out["other_workloads"] = [
                measure_other(device, "syn512", 8192, 3, 1, True),
                measure_other(device, "nasa_lstm", 8192, 20, 2, False),
                measure_other(device, "nasa_lstm", 512, 200, 30, False),  # (0.25 s timed: a 20-step window right behind the
                                                                          #  B = 8192 runs read 22 % low on one box)
                measure_other(device, "nasa_lstm", 2048, 50, 5, False),  # the wavefront with every CU busy
            ]
            out["api_e2e"] = [api_e2e(device, B, "philox"), api_e2e(device, B, "torch")]
            out["api_e2e_over_value"] = out["api_e2e"][0]["samples_per_s"] / value
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (a bounded ~25 s CPU sample)
            out["cpu_baseline"] = cpu_baseline(sd, L, Cn, model.num_layers, model.n_head,
                                               "lstm" if is_lstm else "transformer")
            # (against the reference's own speed: the port's value x the measured reference / port ratio)
            rop = out["cpu_baseline"].get("reference_speed_over_port") or 1.0
            out["speedup_vs_cpu_baseline"] = value / (out["cpu_baseline"]["value"] * rop)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
