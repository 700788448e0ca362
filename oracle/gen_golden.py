"""Generate tests/golden/*.npz by running the UNMODIFIED reference on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container (where /root/reference
is mounted):   python oracle/gen_golden.py
Nothing from the reference is copied: this script imports it from where it
lies (oracle/_ref_import.py) and stores *data* -- expected outputs for seeded
inputs that the tests regenerate from the same seeds
(fastfourierdiffusion_amd.utils.synthetic).

N(0,1) draws are injected: torch.randn / torch.randn_like are patched during
reference sampling to pop from a recorded numpy stream, because torch's CPU
generator stream cannot be reproduced on the device.
"""
from __future__ import annotations

import math
import os
import sys
from contextlib import contextmanager

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle._ref_import import import_reference  # noqa: E402
from oracle import cases  # noqa: E402
from fastfourierdiffusion_amd.utils import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def to_t(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


@contextmanager
def injected_noise(stream):
    """Patch torch.randn / randn_like to read from ``stream`` (iterator of np arrays)."""
    orig_randn, orig_like = torch.randn, torch.randn_like

    def randn(*shape, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        z = next(stream)
        assert tuple(z.shape) == tuple(shape), (z.shape, shape)
        return torch.from_numpy(z.copy())

    def randn_like(t, **kw):
        z = next(stream)
        assert tuple(z.shape) == tuple(t.shape), (z.shape, t.shape)
        return torch.from_numpy(z.copy()).to(t.device)

    torch.randn, torch.randn_like = randn, randn_like
    try:
        yield
    finally:
        torch.randn, torch.randn_like = orig_randn, orig_like


def make_scheduler(ns, c):
    if c["sde"] == "vp":
        s = ns.VPScheduler(fourier_noise_scaling=c["fourier"], **c["sde_kwargs"])
    else:
        s = ns.VEScheduler(fourier_noise_scaling=c["fourier"], **c["sde_kwargs"])
    s.set_noise_scaling(c["L"])
    return s


def make_model(ns, c):
    """Reference model with build-generated weights, warmed to the Q7 fixed point."""
    sch = make_scheduler(ns, c)
    if c["kind"] == "lstm":
        m = ns.LSTMScoreModule(n_channels=c["C"], max_len=c["L"], noise_scheduler=sch,
                               fourier_noise_scaling=c["fourier"], d_model=c["d"], num_layers=c["NL"])
        sd = synthetic.lstm_state_dict(c["C"], c["L"], c["d"], c["NL"], seed=c["wseed"])
    else:
        m = ns.ScoreModule(n_channels=c["C"], max_len=c["L"], noise_scheduler=sch,
                           fourier_noise_scaling=c["fourier"], d_model=c["d"], num_layers=c["NL"],
                           n_head=c["H"])
        sd = synthetic.transformer_state_dict(c["C"], c["L"], c["d"], c["NL"], seed=c["wseed"])
    missing, unexpected = m.load_state_dict(to_t(sd), strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("cached_backbone") for k in missing), missing
    m.eval()
    if c["kind"] != "lstm":
        with torch.no_grad():
            for _ in range(4):  # Q7: embedding max_norm renorm reaches its fixed point
                m.pos_encoder(torch.zeros(1, c["L"], c["d"]))
    return m, sch


def gen_freqca(ns) -> None:
    """G10: FreqCa helpers, spectral density and the cache's FreqCa state after sampling."""
    from fdiff.utils import fourier as rf

    g = {}
    for (name, B, L, D, seed, ratio) in cases.DECOMP_CASES:
        shape = (L, D) if B == 0 else (B, L, D)
        x = torch.from_numpy(next(synthetic.noise_stream(shape, 1, seed)))
        lo, hi = rf.frequency_decompose_fft(x, ratio)
        lo2, hi2 = rf.frequency_decompose_dct(x, ratio)
        assert torch.equal(lo, lo2) and torch.equal(hi, hi2)
        g[f"decomp_{name}_low"], g[f"decomp_{name}_high"] = lo.numpy(), hi.numpy()
    for (name, K, shape, order, ts, target, seed) in cases.HERMITE_CASES:
        hist = [torch.from_numpy(a) for a in synthetic.noise_stream(shape, K, seed)]
        g[f"hermite_{name}"] = rf.predict_hermite(hist, list(ts), target, order).numpy()
    for (L, C, B, seed, apply) in cases.DENSITY_CASES:
        x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, seed)))
        g[f"density_L{L}_C{C}_{int(apply)}"] = rf.spectral_density(x, apply_dft=apply).numpy()
    for c in cases.FREQCA_TRAJ_CASES:
        m, sch = make_model(ns, c)
        B, L, C, N = c["B"], c["L"], c["C"], c["N"]
        nb = max(1, c["num_samples"] // B)
        stream = synthetic.noise_stream((B, L, C), nb * (N + 1), c["zseed"])
        sampler = ns.DiffusionSampler(score_model=m, sample_batch_size=B, use_cache=True,
                                      cache_kwargs=dict(c["cache_kwargs"]))
        with injected_noise(stream):
            out = sampler.sample(num_samples=c["num_samples"], num_diffusion_steps=N)
        cache = m.cache
        name = c["name"]
        g[f"{name}_out"] = out.numpy()
        g[f"{name}_ts"] = sch.timesteps.numpy().copy()  # the grid this host's torch.linspace produced
        g[f"{name}_crf_cache"] = cache.crf_cache.numpy()
        if cache.use_freqca:
            g[f"{name}_low"] = cache.crf_low_cache.numpy()
            g[f"{name}_high_hist"] = torch.stack(cache.crf_high_history, 0).numpy()
            g[f"{name}_t_hist"] = np.array(cache.crf_timestep_history, dtype=np.float64)
            st = cache.get_cache_stats()
            g[f"{name}_stats"] = np.array([st["freq_decomp_count"], st["freq_decomp_skipped"], st["current_step"]],
                                          dtype=np.int64)
            g[f"{name}_pred"] = cache.predict_crf_freqca(c["t_pred"]).numpy()
        print(name, out.shape, len(cache.crf_high_history))
    np.savez_compressed(os.path.join(OUT, "g10_freqca.npz"), **g)


def gen_extra_traj(ns) -> None:
    """G11: further sampler combinations (VE / time domain x cache x FreSca), each with the timestep grid used."""
    g = {}
    for c in cases.EXTRA_TRAJ_CASES:
        m, sch = make_model(ns, c)
        B, L, C, N = c["B"], c["L"], c["C"], c["N"]
        nb = max(1, c["num_samples"] // B)
        stream = synthetic.noise_stream((B, L, C), nb * (N + 1), c["zseed"])
        fk = c.get("fresca")
        fres = {} if fk is None else dict(use_fresca=True, fresca_low_scale=fk["low_scale"],
                                          fresca_high_scale=fk["high_scale"], fresca_cutoff_ratio=fk["cutoff_ratio"],
                                          fresca_cutoff_strategy=fk["cutoff_strategy"])
        sampler = ns.DiffusionSampler(score_model=m, sample_batch_size=B, use_cache=c["use_cache"],
                                      cache_kwargs=dict(c.get("cache_kwargs", {})), **fres)
        with injected_noise(stream):
            out = sampler.sample(num_samples=c["num_samples"], num_diffusion_steps=N)
        g[c["name"]] = out.numpy()
        g[c["name"] + "_ts"] = sch.timesteps.numpy().copy()
        print(c["name"], out.shape, float(out.abs().max()))
    np.savez_compressed(os.path.join(OUT, "g11_extra_traj.npz"), **g)


def gen_round2(ns) -> None:
    """G12: per-sample timesteps, a config-5-shaped cached trajectory, two samplers on one model with different
    gate parameters, and the (un)standardise + (i)dft wrappers of cmd/sample.py / datamodules.py."""
    g = {}
    for c in cases.MIXED_T_CASES:
        m, sch = make_model(ns, c)
        B, L, C = c["B"], c["L"], c["C"]
        x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, c["xseed"])))
        t = torch.tensor(c["t"], dtype=torch.int64 if c["int_t"] else torch.float32)
        with torch.no_grad():
            g[c["name"] + "_score"] = m(ns.DiffusableBatch(X=x, y=None, timesteps=t)).numpy()
            if c.get("recompute"):
                m.enable_caching()
                m.eval()
                m.cache.reset()
                for j, rec in enumerate(c["recompute"]):
                    xj = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, c["xseed"] + 100 + j)))
                    sc, crf = m(ns.DiffusableBatch(X=xj, y=None, timesteps=t), recompute_tokens=set(rec), step=j,
                                return_crf=True)
                    g[f"{c['name']}_cseq{j}_score"] = sc.numpy()
                    g[f"{c['name']}_cseq{j}_crf"] = crf.numpy().copy()
                m.disable_caching()
        print(c["name"])
    for c in cases.SYN_TRAJ_CASES:
        m, sch = make_model(ns, c)
        B, L, C, N = c["B"], c["L"], c["C"], c["N"]
        stream = synthetic.noise_stream((B, L, C), max(1, c["num_samples"] // B) * (N + 1), c["zseed"])
        sampler = ns.DiffusionSampler(score_model=m, sample_batch_size=B, use_cache=True, cache_kwargs=dict(c["cache_kwargs"]))
        with injected_noise(stream):
            out = sampler.sample(num_samples=c["num_samples"], num_diffusion_steps=N)
        g[c["name"]] = out.numpy()
        g[c["name"] + "_ts"] = sch.timesteps.numpy().copy()
        print(c["name"], out.shape, float(out.abs().max()))
    c = cases.TWO_SAMPLER_CASE
    m, sch = make_model(ns, c)
    B, L, C, N = c["B"], c["L"], c["C"], c["N"]
    for tag, kw, zs in (("first", c["first_kwargs"], c["zseed1"]), ("second", c["second_kwargs"], c["zseed2"])):
        stream = synthetic.noise_stream((B, L, C), max(1, c["num_samples"] // B) * (N + 1), zs)
        sampler = ns.DiffusionSampler(score_model=m, sample_batch_size=B, use_cache=True, cache_kwargs=dict(kw))
        with injected_noise(stream):
            out = sampler.sample(num_samples=c["num_samples"], num_diffusion_steps=N)
        g[f"{c['name']}_{tag}"] = out.numpy()
        st = m.cache.get_cache_stats()
        g[f"{c['name']}_{tag}_stats"] = np.array([st["recompute_count"], st["cache_hit_count"]], dtype=np.int64)
    g[c["name"] + "_ts"] = sch.timesteps.numpy().copy()
    print(c["name"])
    for (L, C, B, seed) in cases.AFFINE_FFT_CASES:
        x, mean, std = (torch.from_numpy(a) for a in synthetic.noise_stream((B, L, C), 3, seed))
        mean, std = mean[0], std[0].abs() + 0.5
        g[f"unstd_idft_L{L}_C{C}"] = ns.idft(x * std + mean).numpy()   # cmd/sample.py:107-113
        g[f"dft_std_L{L}_C{C}"] = ((ns.dft(x) - mean) / std).numpy()    # datamodules.py:42-43,61-62
    np.savez_compressed(os.path.join(OUT, "g12_round2.npz"), **g)


def gen_round4(ns) -> None:
    """G13: the full-length LSTM trajectory of BASELINE configs[3], the reference's class-default transformer
    (d_model 60) at the ECG length, and a batch of the reference's default sample_batch_size (50) with the cache on."""
    g = {}
    for c in cases.ROUND4_TRAJ_CASES:
        m, sch = make_model(ns, c)
        B, L, C, N = c["B"], c["L"], c["C"], c["N"]
        nb = max(1, c["num_samples"] // B)
        stream = synthetic.noise_stream((B, L, C), nb * (N + 1), c["zseed"])
        sampler = ns.DiffusionSampler(score_model=m, sample_batch_size=B, use_cache=c["use_cache"],
                                      cache_kwargs=dict(c.get("cache_kwargs", {})))
        with injected_noise(stream):
            out = sampler.sample(num_samples=c["num_samples"], num_diffusion_steps=N)
        g[c["name"]] = out.numpy()
        g[c["name"] + "_ts"] = sch.timesteps.numpy().copy()
        print(c["name"], out.shape, float(out.abs().max()), flush=True)
    np.savez_compressed(os.path.join(OUT, "g13_round4.npz"), **g)


def gen_fresca2d(ns) -> None:
    """G14: the 4-D (batch, H, W, channels) branch of frequency_scale (fresca.py:184-213)."""
    from fdiff.utils.fresca import frequency_scale
    g = {}
    for (name, H, W, C, B, seed, lo, hi, ratio, strat) in cases.FRESCA2D_CASES:
        x = torch.from_numpy(next(synthetic.noise_stream((B, H * W, C), 1, seed))).reshape(B, H, W, C)
        y = frequency_scale(x, low_scale=lo, high_scale=hi, cutoff_ratio=ratio, cutoff_strategy=strat)
        g[name] = y.numpy().copy()
        print(name, tuple(y.shape), float(y.abs().max()), flush=True)
    np.savez_compressed(os.path.join(OUT, "g14_fresca2d.npz"), **g)


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    ns = import_reference()
    torch.set_num_threads(8)
    if "--only-fresca2d" in sys.argv:  # just G14
        gen_fresca2d(ns)
        return
    if "--only-round4" in sys.argv:  # just G13
        gen_round4(ns)
        return
    if "--only-round2" in sys.argv:  # just G12
        gen_round2(ns)
        return
    if "--only-freqca" in sys.argv:  # regenerate just G10 (the 1000-step trajectories take minutes)
        gen_freqca(ns)
        return
    if "--only-extra" in sys.argv:  # just G11
        gen_extra_traj(ns)
        return
    meta = {"torch": torch.__version__}

    # ---- G1: dft / idft -------------------------------------------------
    g = {}
    for (L, C, B, seed) in cases.FFT_CASES:
        x = next(synthetic.noise_stream((B, L, C), 1, seed))
        xt = torch.from_numpy(x)
        g[f"dft_L{L}_C{C}"] = ns.dft(xt).numpy()
        g[f"idft_L{L}_C{C}"] = ns.idft(xt).numpy()
    np.savez_compressed(os.path.join(OUT, "g1_fft.npz"), **g)

    # ---- G2: scheduler tables -------------------------------------------
    g = {}
    for L in cases.TABLE_LENS:
        for fourier in (False, True):
            s = ns.VPScheduler(fourier_noise_scaling=fourier)
            s.set_noise_scaling(L)
            g[f"G_L{L}_f{int(fourier)}"] = s.G.numpy()
    for N in cases.TABLE_STEPS:
        s = ns.VPScheduler()
        s.set_timesteps(N)
        g[f"ts_N{N}"] = s.timesteps.numpy()
        g[f"dt_N{N}"] = s.step_size.numpy()
    np.savez_compressed(os.path.join(OUT, "g2_tables.npz"), **g)

    # ---- G3: VP / VE reverse step with injected z -----------------------
    g = {}
    for c in cases.STEP_CASES:
        sch = make_scheduler(ns, c)
        sch.set_timesteps(c["N"])
        B, L, C = c["B"], c["L"], c["C"]
        st = synthetic.noise_stream((B, L, C), 3, c["seed"])
        x, s, z = (torch.from_numpy(a) for a in st)
        for i in c["idx"]:
            t = sch.timesteps[i].item()
            with injected_noise(iter([z.numpy()])):
                out = sch.step(model_output=s, timestep=t, sample=x).prev_sample
            g[f"{c['name']}_i{i}"] = out.numpy()
        with injected_noise(iter([z.numpy()])):
            g[f"{c['name']}_prior"] = sch.prior_sampling((B, L, C)).numpy()
    np.savez_compressed(os.path.join(OUT, "g3_steps.npz"), **g)

    # ---- G4..G8: models --------------------------------------------------
    g = {}
    for c in cases.MODEL_CASES:
        m, sch = make_model(ns, c)
        B, L, C = c["B"], c["L"], c["C"]
        name = c["name"]
        if c["kind"] != "lstm":
            g[f"{name}_pos_fixed"] = m.pos_encoder.embedding.weight.detach().numpy().copy()
        x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, c["xseed"])))
        with torch.no_grad():
            for tv in c["t_values"]:
                t = torch.full((B,), tv, dtype=torch.float32)
                # G4: time encoder known answer (added to zeros)
                te = m.time_encoder(torch.zeros(B, 1, c["d"]), t)
                g[f"{name}_temb_t{tv}"] = te[:, 0, :].numpy()
                batch = ns.DiffusableBatch(X=x, y=None, timesteps=t)
                g[f"{name}_score_t{tv}"] = m(batch).numpy()  # G5 / G8
            if c.get("cache_seq"):
                # G6: cached forward sequence through the reference's own modes
                m.enable_caching(**c.get("cache_kwargs", {}))
                m.eval()
                m.cache.reset()
                tv = c["t_values"][0]
                t = torch.full((B,), tv, dtype=torch.float32)
                for j, rec in enumerate(c["cache_seq"]):
                    xj = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, c["xseed"] + 100 + j)))
                    batch = ns.DiffusableBatch(X=xj, y=None, timesteps=t)
                    sc, crf = m(batch, recompute_tokens=set(rec), step=j, return_crf=True)
                    g[f"{name}_cseq{j}_score"] = sc.numpy()
                    if c.get("dump_table"):
                        g[f"{name}_cseq{j}_k"] = m.cache.k_cache_tensor.numpy().copy()
                        g[f"{name}_cseq{j}_v"] = m.cache.v_cache_tensor.numpy().copy()
                        g[f"{name}_cseq{j}_crf"] = crf.numpy().copy()
                    else:
                        g[f"{name}_cseq{j}_k_l0h0"] = m.cache.k_cache_tensor[0, 0].numpy().copy()
                        g[f"{name}_cseq{j}_v_lNhN"] = m.cache.v_cache_tensor[-1, -1].numpy().copy()
                st = m.cache.get_cache_stats()
                g[f"{name}_cstats"] = np.array([st["recompute_count"], st["cache_hit_count"]], dtype=np.int64)
                m.disable_caching()
    np.savez_compressed(os.path.join(OUT, "g5_models.npz"), **g)

    # ---- G7: trajectories with injected noise ---------------------------
    g = {}
    for c in cases.TRAJ_CASES:
        m, sch = make_model(ns, c)
        B, L, C, N = c["B"], c["L"], c["C"], c["N"]
        nb = max(1, c["num_samples"] // B)
        stream = synthetic.noise_stream((B, L, C), nb * (N + 1), c["zseed"])
        fk = c.get("fresca")
        fres = {} if fk is None else dict(use_fresca=True, fresca_low_scale=fk["low_scale"],
                                          fresca_high_scale=fk["high_scale"], fresca_cutoff_ratio=fk["cutoff_ratio"],
                                          fresca_cutoff_strategy=fk["cutoff_strategy"])
        sampler = ns.DiffusionSampler(score_model=m, sample_batch_size=B, use_cache=c["use_cache"],
                                      cache_kwargs=dict(c.get("cache_kwargs", {})), **fres)
        with injected_noise(stream):
            out = sampler.sample(num_samples=c["num_samples"], num_diffusion_steps=N)
        g[c["name"]] = out.numpy()
        print(c["name"], out.shape, float(out.abs().max()))
    np.savez_compressed(os.path.join(OUT, "g7_traj.npz"), **g)

    # ---- G6b: FreSca on raw score tensors ---------------------------------
    from fdiff.utils.fresca import apply_fresca_to_score
    g = {}
    for (name, L, C, B, seed, lo, hi, ratio, strat, tstep, nsteps) in cases.FRESCA_CASES:
        x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, seed)))
        y = apply_fresca_to_score(x, low_scale=lo, high_scale=hi, cutoff_ratio=ratio, cutoff_strategy=strat,
                                  timestep=tstep, num_steps=nsteps)
        g[name] = y.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g6_fresca.npz"), **g)
    gen_fresca2d(ns)

    # ---- G9: gate schedule ------------------------------------------------
    g = {}
    for (K, R, L, steps) in cases.GATE_CASES:
        cache = ns.E2CRFCache(num_layers=1, max_len=L, device=torch.device("cpu"), K=K, R=R)
        sizes = np.array([len(cache.determine_recompute_set(None, 0.1, s)) for s in steps], dtype=np.int64)
        first = np.array([sorted(cache.determine_recompute_set(None, 0.1, s))[:1] or [-1] for s in steps],
                         dtype=np.int64).ravel()
        g[f"gate_K{K}_R{R}_L{L}_sizes"] = sizes
        g[f"gate_K{K}_R{R}_L{L}_first"] = first
    np.savez_compressed(os.path.join(OUT, "g9_gate.npz"), **g)

    gen_freqca(ns)
    gen_extra_traj(ns)
    gen_round2(ns)
    gen_round4(ns)

    with open(os.path.join(OUT, "META.txt"), "w") as f:
        f.write("generated by oracle/gen_golden.py from the unmodified reference at /root/reference\n")
        for k, v in meta.items():
            f.write(f"{k}: {v}\n")
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
