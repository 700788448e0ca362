"""Pin the CPU oracle (oracle/ffd_oracle.py) against the reference's outputs.

The .npz files under tests/golden were produced by oracle/gen_golden.py, which
runs the unmodified reference on the seeded inputs regenerated here.  Also
restates the two numeric invariants the reference's own tests hold for this
path: the dft/idft round trip (tests/test_utils.py:36-51, atol 1e-5) and the
encoder known answers (tests/test_transformer.py:18-82, atol 1e-5).
"""
import math

import numpy as np
import pytest
import torch

from conftest import rel_err
from fastfourierdiffusion_amd.utils import synthetic
from oracle import cases
from oracle import ffd_oracle as O


def to_t(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def make_sd(c):
    if c["kind"] == "lstm":
        return to_t(synthetic.lstm_state_dict(c["C"], c["L"], c["d"], c["NL"], seed=c["wseed"]))
    return to_t(synthetic.transformer_state_dict(c["C"], c["L"], c["d"], c["NL"], seed=c["wseed"]))


# single-kernel tolerance: <= 1e-6 relative to the output max-norm
TOL_KERNEL = 2e-6
# trajectory tolerance: <= 1e-5 relative max-norm at up to 1000 steps
TOL_TRAJ = 1e-5


@pytest.mark.parametrize("case", cases.FFT_CASES, ids=lambda c: f"L{c[0]}C{c[1]}")
def test_fft_golden(golden, case):
    L, C, B, seed = case
    x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, seed)))
    g = golden["g1_fft"]
    assert rel_err(O.dft(x), g[f"dft_L{L}_C{C}"]) < TOL_KERNEL
    assert rel_err(O.idft(x), g[f"idft_L{L}_C{C}"]) < TOL_KERNEL


def test_fft_roundtrip_reference_invariant():
    # tests/test_utils.py:36-51 (B=100, C=3, L=100/101, atol=1e-5)
    rng = np.random.default_rng(0)
    for L in (100, 101):
        x = torch.from_numpy(rng.standard_normal((100, L, 3)).astype(np.float32))
        assert torch.allclose(x, O.idft(O.dft(x)), atol=1e-5)
        assert torch.allclose(x, O.dft(O.idft(x)), atol=1e-5)


def test_tables_golden(golden):
    g = golden["g2_tables"]
    for L in cases.TABLE_LENS:
        for f in (False, True):
            np.testing.assert_array_equal(O.noise_scaling(L, f).numpy(), g[f"G_L{L}_f{int(f)}"])
    for N in cases.TABLE_STEPS:
        ts, dt = O.timesteps(N)
        np.testing.assert_array_equal(ts.numpy(), g[f"ts_N{N}"])
        np.testing.assert_array_equal(dt.numpy(), g[f"dt_N{N}"])


@pytest.mark.parametrize("c", cases.STEP_CASES, ids=lambda c: c["name"])
def test_step_golden(golden, c):
    g = golden["g3_steps"]
    B, L, C = c["B"], c["L"], c["C"]
    x, s, z = (torch.from_numpy(a) for a in synthetic.noise_stream((B, L, C), 3, c["seed"]))
    G = O.noise_scaling(L, c["fourier"])
    ts, dt = O.timesteps(c["N"])
    for i in c["idx"]:
        t = ts[i].item()
        out = (O.vp_step if c["sde"] == "vp" else O.ve_step)(x, s, z, t, G, dt, **c["sde_kwargs"])
        # Q8: elementwise restatement vs the reference's diag-matmul form: <= a few ulp
        assert rel_err(out, g[f"{c['name']}_i{i}"]) < 5e-7
    pr = O.prior(z, G, c["sde_kwargs"].get("sigma_max") if c["sde"] == "ve" else None)
    assert rel_err(pr, g[f"{c['name']}_prior"]) < 5e-7


@pytest.mark.parametrize("c", cases.MODEL_CASES, ids=lambda c: c["name"])
def test_model_golden(golden, c):
    g = golden["g5_models"]
    sd = make_sd(c)
    B, L, C, d = c["B"], c["L"], c["C"], c["d"]
    name = c["name"]
    if c["kind"] != "lstm":
        pos = O.renorm_embedding(sd["pos_encoder.embedding.weight"], math.sqrt(d))
        np.testing.assert_allclose(pos.numpy(), g[f"{name}_pos_fixed"], rtol=0, atol=1e-6)
        # tests/test_transformer.py:29 max_norm invariant
        assert float((pos ** 2).sum(-1).max()) <= d + 1e-5
    x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, c["xseed"])))
    for tv in c["t_values"]:
        t = torch.full((B,), tv, dtype=torch.float32)
        temb = O.time_embedding(t, sd["time_encoder.W"], sd["time_encoder.dense.weight"],
                                sd["time_encoder.dense.bias"], d)
        np.testing.assert_allclose(temb.numpy(), g[f"{name}_temb_t{tv}"], rtol=0, atol=1e-5)
        if c["kind"] == "lstm":
            sc = O.lstm_score_forward(x, t, sd, c["NL"])
        else:
            sc = O.score_forward(x, t, sd, c["NL"], c["H"])
        assert rel_err(sc, g[f"{name}_score_t{tv}"]) < TOL_KERNEL * 5
    if c.get("cache_seq"):
        table = O.KVTable(c["NL"], L)
        t = torch.full((B,), c["t_values"][0], dtype=torch.float32)
        for j, rec in enumerate(c["cache_seq"]):
            xj = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, c["xseed"] + 100 + j)))
            sc, crf = O.score_forward(xj, t, sd, c["NL"], c["H"], table, rec, return_crf=True)
            assert rel_err(sc, g[f"{name}_cseq{j}_score"]) < TOL_KERNEL * 5, j
            if c.get("dump_table"):
                assert rel_err(table.k, g[f"{name}_cseq{j}_k"]) < TOL_KERNEL * 5
                assert rel_err(table.v, g[f"{name}_cseq{j}_v"]) < TOL_KERNEL * 5
                assert rel_err(crf, g[f"{name}_cseq{j}_crf"]) < TOL_KERNEL * 5
            else:
                assert rel_err(table.k[0, 0], g[f"{name}_cseq{j}_k_l0h0"]) < TOL_KERNEL * 5
                assert rel_err(table.v[-1, -1], g[f"{name}_cseq{j}_v_lNhN"]) < TOL_KERNEL * 5
        np.testing.assert_array_equal(
            np.array([table.recompute_count, table.cache_hit_count]), g[f"{name}_cstats"])


@pytest.mark.parametrize("case", cases.FRESCA_CASES, ids=lambda c: c[0])
def test_fresca_golden(golden, case):
    name, L, C, B, seed, lo, hi, ratio, strat, tstep, nsteps = case
    x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, seed)))
    y = O.fresca(x, lo, hi, ratio, strat, tstep, nsteps)
    assert rel_err(y, golden["g6_fresca"][name]) < TOL_KERNEL


@pytest.mark.parametrize("case", cases.FRESCA2D_CASES, ids=lambda c: c[0])
def test_fresca2d_golden(golden, case):
    """The 4-D (batch, H, W, channels) branch of frequency_scale (fresca.py:184-213) against the unmodified reference."""
    name, H, W, C, B, seed, lo, hi, ratio, strat = case
    x = torch.from_numpy(next(synthetic.noise_stream((B, H * W, C), 1, seed))).reshape(B, H, W, C)
    y = O.fresca2d(x, lo, hi, ratio, strat)
    assert rel_err(y, golden["g14_fresca2d"][name]) < TOL_KERNEL


_FAST_TRAJ = [c for c in cases.TRAJ_CASES if c["N"] * (c["d"] // 24) <= 400]
_SLOW_TRAJ = [c for c in cases.TRAJ_CASES if c not in _FAST_TRAJ]


def _run_traj(golden, c):
    sd = make_sd(c)
    B, L, C, N = c["B"], c["L"], c["C"], c["N"]
    nb = max(1, c["num_samples"] // B)
    noise = (torch.from_numpy(z) for z in synthetic.noise_stream((B, L, C), nb * (N + 1), c["zseed"]))
    ck = c.get("cache_kwargs", {})
    out = O.sample(sd, kind=c["kind"], n_channels=C, max_len=L, num_layers=c["NL"], n_head=c["H"],
                   sde=c["sde"], sde_kwargs=c["sde_kwargs"], fourier_noise_scaling=c["fourier"],
                   num_samples=c["num_samples"], batch_size=B, num_steps=N, noise=noise,
                   use_cache=c["use_cache"], K=ck.get("K", 5), R=ck.get("R", 10), fresca_kwargs=c.get("fresca"))
    ref = golden["g7_traj"][c["name"]]
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL_TRAJ, rel_err(out, ref)


@pytest.mark.parametrize("c", _FAST_TRAJ, ids=lambda c: c["name"])
def test_traj_golden(golden, c):
    _run_traj(golden, c)


@pytest.mark.slow
@pytest.mark.parametrize("c", _SLOW_TRAJ, ids=lambda c: c["name"])
def test_traj_golden_1000(golden, c):
    torch.set_num_threads(8)
    _run_traj(golden, c)


@pytest.mark.parametrize("case", cases.GATE_CASES, ids=lambda c: f"K{c[0]}R{c[1]}L{c[2]}")
def test_gate_golden(golden, case):
    K, R, L, steps = case
    g = golden["g9_gate"]
    sizes = [len(O.gate(s, L, K, R)) for s in steps]
    first = [(O.gate(s, L, K, R)[:1] or [-1])[0] for s in steps]
    np.testing.assert_array_equal(sizes, g[f"gate_K{K}_R{R}_L{L}_sizes"])
    np.testing.assert_array_equal(first, g[f"gate_K{K}_R{R}_L{L}_first"])


# ---- FreqCa helpers, spectral density, FreqCa cache state (scope row (f)2/(f)3) ----
# Hermite prediction solves ridge-regularised normal equations in fp32 inside the reference
# (torch.linalg.inv); its result is conditioning-limited, hence the looser tolerance.
TOL_HERMITE = 2e-3


@pytest.mark.parametrize("case", cases.DECOMP_CASES, ids=lambda c: c[0])
def test_freq_decompose_golden(golden, case):
    name, B, L, D, seed, ratio = case
    shape = (L, D) if B == 0 else (B, L, D)
    x = torch.from_numpy(next(synthetic.noise_stream(shape, 1, seed)))
    lo, hi = O.frequency_decompose(x, ratio)
    g = golden["g10_freqca"]
    assert lo.shape == g[f"decomp_{name}_low"].shape
    assert rel_err(lo, g[f"decomp_{name}_low"]) < TOL_KERNEL
    assert rel_err(hi, g[f"decomp_{name}_high"]) < TOL_KERNEL
    assert rel_err(lo + hi, x) < 4 * TOL_KERNEL  # the two parts partition the spectrum


@pytest.mark.parametrize("case", cases.HERMITE_CASES, ids=lambda c: c[0])
def test_hermite_golden(golden, case):
    name, K, shape, order, ts, target, seed = case
    hist = [torch.from_numpy(a) for a in synthetic.noise_stream(shape, K, seed)]
    y = O.predict_hermite(hist, list(ts), target, order)
    assert rel_err(y, golden["g10_freqca"][f"hermite_{name}"]) < TOL_HERMITE


@pytest.mark.parametrize("case", cases.DENSITY_CASES, ids=lambda c: f"L{c[0]}C{c[1]}")
def test_spectral_density_golden(golden, case):
    L, C, B, seed, apply = case
    x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, seed)))
    assert rel_err(O.spectral_density(x, apply), golden["g10_freqca"][f"density_L{L}_C{C}_{int(apply)}"]) < 2 * TOL_KERNEL


@pytest.mark.parametrize("c", cases.FREQCA_TRAJ_CASES, ids=lambda c: c["name"])
def test_freqca_state_golden(golden, c):
    sd = make_sd(c)
    B, L, C, N = c["B"], c["L"], c["C"], c["N"]
    nb = max(1, c["num_samples"] // B)
    noise = (torch.from_numpy(z) for z in synthetic.noise_stream((B, L, C), nb * (N + 1), c["zseed"]))
    ck = c["cache_kwargs"]
    st = O.FreqCaState(R=ck.get("R", 10), low_freq_ratio=ck.get("low_freq_ratio", 0.3),
                       max_history=ck.get("max_history", 10), interval=ck.get("freq_decomp_interval", 10),
                       use_freqca=ck.get("use_freqca", False))
    out = O.sample(sd, kind=c["kind"], n_channels=C, max_len=L, num_layers=c["NL"], n_head=c["H"], sde=c["sde"],
                   sde_kwargs=c["sde_kwargs"], fourier_noise_scaling=c["fourier"], num_samples=c["num_samples"],
                   batch_size=B, num_steps=N, noise=noise, use_cache=True, K=ck.get("K", 5), R=ck.get("R", 10),
                   freqca=st)
    g, name = golden["g10_freqca"], c["name"]
    assert rel_err(out, g[f"{name}_out"]) < TOL_TRAJ
    assert rel_err(st.crf_cache, g[f"{name}_crf_cache"]) < TOL_TRAJ
    if st.use_freqca:
        assert rel_err(st.low, g[f"{name}_low"]) < TOL_TRAJ
        assert rel_err(torch.stack(st.high_history, 0), g[f"{name}_high_hist"]) < TOL_TRAJ
        np.testing.assert_allclose(np.array(st.t_history), g[f"{name}_t_hist"], rtol=0, atol=0)
        assert len(st.high_history) == int(g[f"{name}_stats"][0])
        pred = st.low + O.predict_hermite(st.high_history, st.t_history, c["t_pred"], ck.get("hermite_order", 3))
        assert rel_err(pred, g[f"{name}_pred"]) < TOL_HERMITE


# ---- G11: further sampler combinations pinned against the reference ----
@pytest.mark.parametrize("c", cases.EXTRA_TRAJ_CASES, ids=lambda c: c["name"])
def test_extra_traj_golden(golden, c):
    sd = make_sd(c)
    B, L, C, N = c["B"], c["L"], c["C"], c["N"]
    nb = max(1, c["num_samples"] // B)
    noise = (torch.from_numpy(z) for z in synthetic.noise_stream((B, L, C), nb * (N + 1), c["zseed"]))
    ck = c.get("cache_kwargs", {})
    out = O.sample(sd, kind=c["kind"], n_channels=C, max_len=L, num_layers=c["NL"], n_head=c["H"], sde=c["sde"],
                   sde_kwargs=c["sde_kwargs"], fourier_noise_scaling=c["fourier"], num_samples=c["num_samples"],
                   batch_size=B, num_steps=N, noise=noise, use_cache=c["use_cache"], K=ck.get("K", 5), R=ck.get("R", 10),
                   fresca_kwargs=c.get("fresca"))
    ref = golden["g11_extra_traj"][c["name"]]
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL_TRAJ, rel_err(out, ref)


# ---- G12 (round 2): per-sample timesteps, config-5-shaped cached trajectory, two samplers, affine (i)dft ----
@pytest.mark.parametrize("c", cases.MIXED_T_CASES, ids=lambda c: c["name"])
def test_mixed_timesteps_golden(golden, c):
    g = golden["g12_round2"]
    sd = make_sd(c)
    B, L, C = c["B"], c["L"], c["C"]
    x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, c["xseed"])))
    t = torch.tensor(c["t"], dtype=torch.float32)  # integer timesteps promote to fp32 in timesteps * W (transformer.py:80)
    sc = O.lstm_score_forward(x, t, sd, c["NL"]) if c["kind"] == "lstm" else O.score_forward(x, t, sd, c["NL"], c["H"])
    assert rel_err(sc, g[c["name"] + "_score"]) < TOL_KERNEL * 5
    if c.get("recompute"):
        table = O.KVTable(c["NL"], L)
        for j, rec in enumerate(c["recompute"]):
            xj = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, c["xseed"] + 100 + j)))
            sc, crf = O.score_forward(xj, t, sd, c["NL"], c["H"], table, rec, return_crf=True)
            assert rel_err(sc, g[f"{c['name']}_cseq{j}_score"]) < TOL_KERNEL * 5, j
            assert rel_err(crf, g[f"{c['name']}_cseq{j}_crf"]) < TOL_KERNEL * 5, j


def _oracle_sample(c, zseed, ck):
    sd = make_sd(c)
    B, L, C, N = c["B"], c["L"], c["C"], c["N"]
    nb = max(1, c["num_samples"] // B)
    noise = (torch.from_numpy(z) for z in synthetic.noise_stream((B, L, C), nb * (N + 1), zseed))
    return O.sample(sd, kind=c["kind"], n_channels=C, max_len=L, num_layers=c["NL"], n_head=c["H"], sde=c["sde"],
                    sde_kwargs=c["sde_kwargs"], fourier_noise_scaling=c["fourier"], num_samples=c["num_samples"],
                    batch_size=B, num_steps=N, noise=noise, use_cache=True, K=ck.get("K", 5), R=ck.get("R", 10))


@pytest.mark.parametrize("c", cases.SYN_TRAJ_CASES, ids=lambda c: c["name"])
def test_syn_traj_golden(golden, c):
    out = _oracle_sample(c, c["zseed"], c["cache_kwargs"])
    assert rel_err(out, golden["g12_round2"][c["name"]]) < TOL_TRAJ


def test_two_samplers_golden(golden):
    """Each sampler on a shared model behaves like a fresh run with ITS cache's K / R: step 0 rewrites the tables
    the layers stay bound to (Q5), so the first sampler's configuration leaves no trace in the second's samples."""
    c = cases.TWO_SAMPLER_CASE
    g = golden["g12_round2"]
    assert rel_err(_oracle_sample(c, c["zseed1"], c["first_kwargs"]), g["two_samplers_first"]) < TOL_TRAJ
    assert rel_err(_oracle_sample(c, c["zseed2"], c["second_kwargs"]), g["two_samplers_second"]) < TOL_TRAJ
    # the second sampler's own cache object never sees a hit or a recompute (Q5)
    np.testing.assert_array_equal(g["two_samplers_second_stats"], [0, 0])


@pytest.mark.parametrize("case", cases.AFFINE_FFT_CASES, ids=lambda c: f"L{c[0]}C{c[1]}")
def test_affine_fft_golden(golden, case):
    L, C, B, seed = case
    x, mean, std = (torch.from_numpy(a) for a in synthetic.noise_stream((B, L, C), 3, seed))
    mean, std = mean[0], std[0].abs() + 0.5
    g = golden["g12_round2"]
    assert rel_err(O.unstandardize_idft(x, mean, std), g[f"unstd_idft_L{L}_C{C}"]) < TOL_KERNEL
    assert rel_err(O.dft_standardize(x, mean, std), g[f"dft_std_L{L}_C{C}"]) < TOL_KERNEL


# ---- G13 (round 4): BASELINE configs[3] at its full 1000 steps, the class-default d_model 60, sample_batch_size 50 ----
@pytest.mark.slow
@pytest.mark.parametrize("c", cases.ROUND4_TRAJ_CASES, ids=lambda c: c["name"])
def test_round4_traj_golden(golden, c):
    torch.set_num_threads(8)
    sd = make_sd(c)
    B, L, C, N = c["B"], c["L"], c["C"], c["N"]
    nb = max(1, c["num_samples"] // B)
    noise = (torch.from_numpy(z) for z in synthetic.noise_stream((B, L, C), nb * (N + 1), c["zseed"]))
    ck = c.get("cache_kwargs", {})
    out = O.sample(sd, kind=c["kind"], n_channels=C, max_len=L, num_layers=c["NL"], n_head=c["H"], sde=c["sde"],
                   sde_kwargs=c["sde_kwargs"], fourier_noise_scaling=c["fourier"], num_samples=c["num_samples"],
                   batch_size=B, num_steps=N, noise=noise, use_cache=c["use_cache"], K=ck.get("K", 5), R=ck.get("R", 10),
                   stock_modules=c["kind"] == "lstm" and N >= 1000)  # (2.5 M explicit cell steps would take minutes)
    ref = golden["g13_round4"][c["name"]]
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL_TRAJ, rel_err(out, ref)


def test_stock_lstm_form_equals_the_explicit_restatement(golden):
    """The 1000-step LSTM trajectory check runs the oracle on stock nn.LSTM layers (torch's fused CPU LSTM): the same
    function as the explicit cell loop, and it meets the same goldens."""
    for name in ("nasa_lstm", "small_lstm"):
        c = next(c for c in cases.MODEL_CASES if c["name"] == name)
        sd = make_sd(c)
        B, L, C = c["B"], c["L"], c["C"]
        x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, c["xseed"])))
        for tv in c["t_values"]:
            t = torch.full((B,), tv, dtype=torch.float32)
            a = O.lstm_score_forward_stock(x, t, sd, c["NL"])
            assert rel_err(a, O.lstm_score_forward(x, t, sd, c["NL"])) < TOL_KERNEL
            assert rel_err(a, golden["g5_models"][f"{name}_score_t{tv}"]) < TOL_KERNEL * 5


def test_stock_module_form_equals_the_explicit_restatement(golden):
    """bench.py's CPU baseline times the oracle with torch's stock nn.TransformerEncoder as backbone (the construction
    of score_models.py:61-66, hence torch's fused encoder-layer path the reference runs on): it must be the same
    function as the explicit restatement, and meet the same goldens."""
    for name in ("ecg", "reftest"):
        c = next(c for c in cases.MODEL_CASES if c["name"] == name)
        sd = make_sd(c)
        B, L, C = c["B"], c["L"], c["C"]
        x = torch.from_numpy(next(synthetic.noise_stream((B, L, C), 1, c["xseed"])))
        for tv in c["t_values"]:
            t = torch.full((B,), tv, dtype=torch.float32)
            a = O.score_forward_stock(x, t, sd, c["NL"], c["H"])
            assert rel_err(a, O.score_forward(x, t, sd, c["NL"], c["H"])) < TOL_KERNEL
            assert rel_err(a, golden["g5_models"][f"{name}_score_t{tv}"]) < TOL_KERNEL * 5
