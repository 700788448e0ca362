#!/usr/bin/env python3
"""Headline benchmark: samples/sec at 1000 diffusion steps, ECG frequency-domain
(BASELINE.json configs[1]) on N MI355X.

A "step" is one reverse-diffusion step (score evaluation + Euler-Maruyama update) of
one batch of B=512 synthetic series per GPU, enqueued through libffd's
``ffd_sample_batch``.  K steps are timed between barrier + synchronize pairs; the
metric is  value = N_gpus * B / (1000 * seconds_per_step)  -- the rate at which
complete 1000-step samples leave the node (the prior draw and the final idft are two
more launches per 1000 steps, < 0.01 % of the time).  With the default
K = 1000 the timed region *is* one complete sampling of the batch.

Sampling is embarrassingly parallel over the batch: ranks are independent shards
(weights replicated, Philox noise keyed by global sample index), no data-path
collective; the only torch.distributed traffic is the barrier and the max-over-ranks
of the elapsed time.

Rank 0 prints ONE JSON line (plus human-readable notes on stderr).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense fp32 matrix peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(device, workload):
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


def run_steps(model, sampler, X, ts_c, n_total, step_size, first, n_run, use_cache, stream, offset):
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
    on a bounded sample of the same workload."""
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
            score = O.lstm_score_forward(x, t, sdt, NL)
        else:
            score = O.score_forward(x, t, sdt, NL, H)
        x = O.vp_step(x, score, torch.randn(B, L, Cn, generator=g), tv, G, dt)
        if i >= warm:
            t_acc += time.perf_counter() - t0
    s_per_step = t_acc / steps
    return {"value": B / (1000.0 * s_per_step), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU fp32), B={B}, {steps} of 1000 steps timed after {warm} warm-up, x(1000/{steps})",
            "ms_per_step": s_per_step * 1e3}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=512, help="samples per GPU")
    ap.add_argument("--workload", default="ecg", choices=["ecg", "syn512", "nasa_lstm"])
    ap.add_argument("--cache", action="store_true", help="time the E2-CRF cached path (BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cache-ratio side measurements")
    ap.add_argument("--tune", action="append", default=[], help="key=value for ffd_tune (experiments)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a one-GPU box: every rank uses cuda:0 and the barrier / max-reduce "
                         "run over gloo (RCCL refuses two ranks on one device); the reported number is meaningless")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
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
    from fastfourierdiffusion_amd.sharding import reduce_max_seconds

    for kv in args.tune:
        k, v = kv.split("=")
        assert N.lib().ffd_tune(k.encode(), int(v)) == 0, kv
    model, sch, sd = build_model(device, args.workload)
    B, L, Cn = args.batch, model.max_len, model.n_channels
    n_total = 1000
    sch.set_timesteps(n_total)
    ts_c = (C.c_float * n_total)(*sch.timesteps.tolist())
    step_size = float(sch.step_size)
    use_cache = bool(args.cache)
    sampler = DiffusionSampler(model, B, use_cache=use_cache, cache_kwargs={}, rng="philox", seed=42)
    stream = N.current_stream_ptr(device)
    offset = rank * B  # global sample index of this shard's first sample

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
        run_steps(model, sampler, Xw, ts_c, n_total, step_size, 0, min(W, n_total), use_cache, stream, offset)
    # timed region: the first K steps of a real sampling of a fresh batch (for the cached
    # path this includes the table-filling step 0, as in the reference's benchmark)
    if use_cache:
        model.cache.reset()
    X = sampler.sample_prior(B, _sample_offset=offset)
    first = 0
    barrier()
    t0 = time.perf_counter()
    run_steps(model, sampler, X, ts_c, n_total, step_size, first, K, use_cache, stream, offset)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = reduce_max_seconds(elapsed, None if args.rehearse_one_gpu else device)
    assert torch.isfinite(X).all(), "non-finite samples"
    ms_per_step = elapsed / K * 1e3
    value = world * B / (1000.0 * (elapsed / K))

    out = {
        "metric": "samples/sec at 1000 diffusion steps, ECG freq-domain" if args.workload == "ecg" else
                  f"samples/sec at 1000 diffusion steps, {args.workload}",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": {"ecg": "ECG frequency-domain L=187 C=1, transformer d72/H12/NL10/F2048, VP beta[0.1,20], "
                                       "1000-step sampler (BASELINE configs[1])",
                                "syn512": "synthetic L=512 C=8 transformer (configs[4] per-GPU shard)",
                                "nasa_lstm": "NASA-charge L=251 C=4 LSTM d72/NL10 (configs[3])"}[args.workload],
                   "batch_per_gpu": B, "global_batch": B * world, "diffusion_steps": n_total,
                   "cache": use_cache, "noise": "philox on device", "sharding": f"batch x{world}, no collectives"},
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
        if args.workload != "nasa_lstm":
            # dominant kernel, timed in situ: HIP event pairs around every k_ffn_ln launch of 20 more
            # sampling steps on the launch stream (same quantity rocprofv3 --kernel-trace reports)
            NLy = model.num_layers
            N.check(lib.ffd_ffn_timing_begin(ctx.handle, 20 * NLy), ctx.handle, "ffd_ffn_timing_begin")
            run_steps(model, sampler, X, ts_c, n_total, step_size, 1, 20, use_cache, stream, offset)
            ms, cnt = C.c_float(), C.c_int()
            N.check(lib.ffd_ffn_timing_end(ctx.handle, C.byref(ms), C.byref(cnt)), ctx.handle, "ffd_ffn_timing_end")
            fl = lib.ffd_ffn_flops_per_launch(ctx.handle, B)
            ach = fl / (ms.value * 1e-3) / 1e12
            iso = C.c_float()
            N.check(lib.ffd_bench_ffn(ctx.handle, B, 50, C.byref(iso), stream), ctx.handle, "ffd_bench_ffn")
            traffic = None  # HBM bytes per launch from the committed PMC passes (cannot be collected in-process)
            tpath = os.path.join(ROOT, "profiles", "r01_ffn_traffic.json")
            if args.workload == "ecg" and B == 512 and os.path.exists(tpath):
                traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
            out["roofline"] = {"kernel": "k_ffn_ln<72,4> (fused FFN: linear1 + relu + linear2 + residual + LayerNorm2)",
                               "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                               "traffic_note": "bytes per launch, rocprofv3 FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, "
                                               "profiles/r01_ffn_traffic.json",
                               "flops_per_launch": fl, "ms_per_launch": ms.value, "launches_timed": cnt.value,
                               "ms_per_launch_back_to_back": iso.value}
            if not use_cache and world == 1:
                # cache-on / cache-off ratio at the same batch (BASELINE configs[2]), 200 steps each
                def timed(uc):
                    s2 = DiffusionSampler(model, B, use_cache=uc, cache_kwargs={}, rng="philox", seed=42)
                    if uc:
                        model._first_cache.reset()
                    X2 = s2.sample_prior(B)
                    run_steps(model, s2, X2, ts_c, n_total, step_size, 0, 20, uc, stream, 0)
                    torch.cuda.synchronize(device)
                    t1 = time.perf_counter()
                    run_steps(model, s2, X2, ts_c, n_total, step_size, 20, 200, uc, stream, 0)
                    torch.cuda.synchronize(device)
                    return (time.perf_counter() - t1) / 200

                t_off, t_on = timed(False), timed(True)
                model.disable_caching()
                out["cache_ratio"] = {"off_over_on": t_off / t_on, "ms_off": t_off * 1e3, "ms_on": t_on * 1e3,
                                      "batch": B, "note": "pure-cache steps (K/V projections skipped), 200 steps"}
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (a bounded ~25 s CPU sample)
            out["cpu_baseline"] = cpu_baseline(sd, L, Cn, model.num_layers, model.n_head,
                                               "lstm" if args.workload == "nasa_lstm" else "transformer")
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
