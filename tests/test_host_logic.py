"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol
include/ffd.h declares, host-only entry points agree with the golden vectors, the
fdiff-compatible Python surface keeps the reference's names / state_dict keys / error
behaviour, and the multi-GPU shard plan is exercised with world_size-2 gloo."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from fastfourierdiffusion_amd import _native

    if not os.path.exists(_native.LIB_PATH):
        from fastfourierdiffusion_amd.build import build

        build()
    return _native.lib()


def test_library_exports_every_declared_symbol(lib):
    from fastfourierdiffusion_amd import _native

    header = open(os.path.join(ROOT, "include", "ffd.h")).read()
    declared = set(re.findall(r"\b(ffd_[a-z0-9_]+)\s*\(", header))
    declared -= {"ffd_ctx"}
    assert declared, "no declarations parsed"
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    nm = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ffd_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    assert b"gfx950" in lib.ffd_version()


def test_no_device_fails_loudly(lib):
    """Without a GPU (this container) context creation reports an error, never a CPU path."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fastfourierdiffusion_amd import _native as N

    desc = N.ModelDesc(0, 1, 187, 72, 12, 10, 2048, 0, 0.1, 20.0, 1, 1e-5)
    h = C.c_void_p()
    rc = lib.ffd_create(C.byref(h), C.byref(desc), 0)
    assert rc == -4 and h.value
    assert b"no HIP device" in lib.ffd_last_error(h) or b"failed" in lib.ffd_last_error(h)
    lib.ffd_destroy(h)
    # unsupported shapes are rejected before any device work
    for bad in (N.ModelDesc(0, 1, 187, 36, 12, 10, 2048, 0, 0.1, 20.0, 1, 1e-5),   # d_model w/o kernel
                N.ModelDesc(0, 1, 600, 72, 12, 10, 2048, 0, 0.1, 20.0, 1, 1e-5),   # L > 512
                N.ModelDesc(0, 1, 187, 72, 12, 10, 2000, 0, 0.1, 20.0, 1, 1e-5)):  # F % 64
        rc = lib.ffd_create(C.byref(h), C.byref(bad), 0)
        assert rc == -2, lib.ffd_last_error(h)
        lib.ffd_destroy(h)


def test_host_noise_scaling_matches_golden(lib, golden):
    g = golden["g2_tables"]
    for L in cases.TABLE_LENS:
        for f in (0, 1):
            out = (C.c_float * L)()
            assert lib.ffd_host_noise_scaling(L, f, out) == 0
            np.testing.assert_array_equal(np.array(out[:], dtype=np.float32), g[f"G_L{L}_f{f}"])


def test_host_timesteps_within_one_ulp_of_torch(lib, golden):
    """torch.linspace's last ulp depends on the host's vector width, so the library takes
    the grid from the caller; its own scalar linspace must agree to <= 1 ulp."""
    g = golden["g2_tables"]
    for n in cases.TABLE_STEPS:
        ts = (C.c_float * n)()
        dt = C.c_float()
        assert lib.ffd_host_timesteps(n, 1e-5, ts, C.byref(dt)) == 0
        a = np.array(ts[:], dtype=np.float32)
        ref = g[f"ts_N{n}"]
        assert np.all(np.abs(a - ref) <= np.spacing(np.maximum(np.abs(ref), 1e-5).astype(np.float32)))
        assert a[0] == 1.0 and a[-1] == np.float32(1e-5)
    assert lib.ffd_host_timesteps(1, 1e-5, ts, C.byref(dt)) == -1  # reference: timesteps[1] IndexError


@pytest.mark.parametrize("case", cases.GATE_CASES, ids=lambda c: f"K{c[0]}R{c[1]}L{c[2]}")
def test_host_gate_matches_golden(lib, golden, case):
    from fastfourierdiffusion_amd.utils.caching import E2CRFCache

    K, R, L, steps = case
    g = golden["g9_gate"]
    sizes = [lib.ffd_host_gate(s, L, K, R) for s in steps]
    np.testing.assert_array_equal(sizes, g[f"gate_K{K}_R{R}_L{L}_sizes"])
    cache = E2CRFCache(num_layers=1, max_len=L, device=torch.device("cpu"), K=K, R=R)
    for s, n in zip(steps, sizes):
        assert cache.determine_recompute_set(None, 0.1, s) == set(range(n))  # always a prefix


def test_fdiff_surface_and_state_dict_keys():
    import fastfourierdiffusion_amd as pkg

    pkg.install_as_fdiff(force=True)
    from fdiff.models.score_models import LSTMScoreModule, ScoreModule
    from fdiff.sampling.sampler import DiffusionSampler
    from fdiff.schedulers.sde import SDE, VEScheduler, VPScheduler
    from fdiff.utils.caching import E2CRFCache
    from fdiff.utils.dataclasses import DiffusableBatch, collate_batch
    from fdiff.utils.fourier import dft, idft  # noqa: F401
    from fastfourierdiffusion_amd.utils import synthetic

    sch = VPScheduler(beta_min=0.1, beta_max=20, fourier_noise_scaling=True)
    assert isinstance(sch, SDE) and sch.T == 1.0 and sch.eps == 1e-5
    sch.set_noise_scaling(187)
    sch.set_timesteps(1000)
    assert sch.G.shape == (187,) and sch.G_matrix.shape == (187, 187) and sch.timesteps.shape == (1000,)
    assert float(sch.step_size) == pytest.approx(0.0010010004, rel=1e-6)
    m = ScoreModule(n_channels=1, max_len=187, noise_scheduler=sch, d_model=72, num_layers=10, n_head=12)
    ref_keys = set(synthetic.transformer_state_dict(1, 187, 72, 10).keys())
    assert set(m.state_dict().keys()) == ref_keys
    assert sum(p.numel() for p in m.parameters()) == 3_202_413  # SURVEY: confirmed by import of the reference
    for a in ("max_len", "n_channels", "noise_scheduler", "num_training_steps", "d_model", "scale_noise", "cache",
              "use_cache", "cached_backbone", "device"):
        assert hasattr(m, a), a
    lm = LSTMScoreModule(n_channels=4, max_len=251, noise_scheduler=VPScheduler(), d_model=72, num_layers=10)
    assert set(lm.state_dict().keys()) == set(synthetic.lstm_state_dict(4, 251, 72, 10).keys())
    assert sum(p.numel() for p in lm.parameters()) == 426_424
    with pytest.raises(AttributeError):  # Q9: LSTM cannot enable caching
        lm.enable_caching()
    with pytest.raises(NotImplementedError):
        ScoreModule(n_channels=1, max_len=8, noise_scheduler=object())
    s = DiffusionSampler(score_model=m, sample_batch_size=50)
    for a in ("score_model", "noise_scheduler", "sample_batch_size", "n_channels", "max_len", "use_cache"):
        assert hasattr(s, a), a
    s2 = DiffusionSampler(score_model=m, sample_batch_size=1, use_cache=True, cache_kwargs={"K": 3, "R": 150})
    assert isinstance(m.cache, E2CRFCache) and m.cache.K == 3 and m.cache.R == 150 and m.use_cache
    first = m.cache
    DiffusionSampler(score_model=m, sample_batch_size=1, use_cache=True, cache_kwargs={})
    assert m.cache is not first and m._first_cache is first  # Q5
    assert set(m.cache.get_cache_stats()) == {"cache_hit_ratio", "cache_ratio", "recompute_count", "cache_hit_count",
                                              "current_step"}
    m.disable_caching()
    assert m.cache is None and not m.use_cache
    b = collate_batch([{"X": torch.zeros(5, 2), "timestep": torch.tensor(0.5)} for _ in range(3)])
    assert isinstance(b, DiffusableBatch) and len(b) == 3 and b.device.type == "cpu"
    ve = VEScheduler(sigma_min=0.01, sigma_max=2)
    assert ve.sigma_max == 2 and ve.noise_scaling is False
    del s2


def test_same_seed_same_init_as_torch_reference_modules():
    """Equal torch seeds give the weights the reference would draw (same module
    construction order, score_models.py:55-66): checked against stock torch modules."""
    import math

    import torch.nn as nn

    from fastfourierdiffusion_amd.models.score_models import ScoreModule
    from fastfourierdiffusion_amd.schedulers.sde import VPScheduler

    torch.manual_seed(123)
    m = ScoreModule(n_channels=3, max_len=20, noise_scheduler=VPScheduler(), d_model=24, num_layers=2, n_head=4)
    torch.manual_seed(123)
    emb = nn.Embedding(20, 24, max_norm=math.sqrt(24))
    W = torch.randn(12) * 30.0
    dense = nn.Linear(24, 24)
    embedder = nn.Linear(3, 24)
    assert torch.equal(m.pos_encoder.embedding.weight, emb.weight)
    assert torch.equal(m.time_encoder.W, W)
    assert torch.equal(m.time_encoder.dense.weight, dense.weight)
    assert torch.equal(m.embedder.weight, embedder.weight)


def test_load_from_checkpoint_roundtrip(tmp_path):
    """cmd/sample.py:68-75 path: a Lightning-style checkpoint (hyper_parameters + state_dict, with the
    scheduler object pickled under the reference's module path) loads through the mirror.  No real
    checkpoint exists in this environment, so the file is synthesised here (parity unpinned)."""
    import fastfourierdiffusion_amd as pkg

    pkg.install_as_fdiff(force=True)
    from fdiff.models.score_models import LSTMScoreModule, ScoreModule
    from fdiff.schedulers.sde import VPScheduler

    sch = VPScheduler(beta_min=0.1, beta_max=20, fourier_noise_scaling=True)
    src = ScoreModule(n_channels=2, max_len=12, noise_scheduler=sch, d_model=24, num_layers=2, n_head=4)
    sd = dict(src.state_dict())
    sd["cached_backbone.0.linear1.weight"] = torch.zeros(3)  # left behind by enable_caching during training
    ckpt = {"state_dict": sd, "pytorch-lightning_version": "2.1.0", "epoch": 3,
            "hyper_parameters": dict(n_channels=2, max_len=12, noise_scheduler=sch, fourier_noise_scaling=True,
                                     d_model=24, num_layers=2, n_head=4, num_training_steps=1000, lr_max=1e-3,
                                     likelihood_weighting=False)}
    path = tmp_path / "epoch=3.ckpt"
    torch.save(ckpt, path)
    m = ScoreModule.load_from_checkpoint(checkpoint_path=str(path), weights_only=False)
    assert isinstance(m.noise_scheduler, VPScheduler) and m.noise_scheduler.beta_1 == 20 and m.max_len == 12
    for k, v in src.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    l_src = LSTMScoreModule(n_channels=2, max_len=12, noise_scheduler=sch, d_model=8, num_layers=2)
    torch.save({"state_dict": l_src.state_dict(),
                "hyper_parameters": dict(n_channels=2, max_len=12, noise_scheduler=sch, d_model=8, num_layers=2)},
               tmp_path / "l.ckpt")
    lm = LSTMScoreModule.load_from_checkpoint(tmp_path / "l.ckpt")
    assert torch.equal(lm.state_dict()["backbone.1.weight_hh_l0"], l_src.state_dict()["backbone.1.weight_hh_l0"])


def test_cpu_tensors_are_refused():
    from fastfourierdiffusion_amd._native import FFDError
    from fastfourierdiffusion_amd.models.score_models import ScoreModule
    from fastfourierdiffusion_amd.schedulers.sde import VPScheduler
    from fastfourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    from fastfourierdiffusion_amd.utils.fourier import dft

    sch = VPScheduler()
    sch.set_noise_scaling(20)
    m = ScoreModule(n_channels=3, max_len=20, noise_scheduler=sch, d_model=24, num_layers=2, n_head=4)
    with pytest.raises(FFDError):
        m(DiffusableBatch(X=torch.zeros(2, 20, 3), timesteps=torch.ones(2)))
    with pytest.raises(AssertionError):
        m(DiffusableBatch(X=torch.zeros(2, 21, 3), timesteps=torch.ones(2)))
    if not torch.cuda.is_available():
        with pytest.raises(FFDError):
            dft(torch.zeros(2, 20, 3))
        with pytest.raises(FFDError):
            sch.step(torch.zeros(2, 20, 3), 0.5, torch.zeros(2, 20, 3))


_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from fastfourierdiffusion_amd.sharding import shard_range, reduce_max_seconds
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
off, cnt = shard_range(1001, 2, rank)
t = reduce_max_seconds(0.5 + rank, None)
gathered = [None, None]
dist.all_gather_object(gathered, (off, cnt))
if rank == 0:
    print(json.dumps({{"ranges": gathered, "tmax": t}}))
dist.destroy_process_group()
"""


def test_shard_plan_two_ranks_gloo(tmp_path):
    from fastfourierdiffusion_amd.sharding import shard_range

    # pure function: contiguous, disjoint, covering, balanced within 1
    for n, w in ((1001, 2), (65536, 8), (7, 8), (512, 1)):
        rs = [shard_range(n, w, r) for r in range(w)]
        assert rs[0][0] == 0 and sum(c for _, c in rs) == n
        assert all(rs[i][0] + rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        assert max(c for _, c in rs) - min(c for _, c in rs) <= 1
    script = tmp_path / "w.py"
    port = 29000 + (os.getpid() % 2000)
    script.write_text(_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    import json

    res = json.loads(outs[0].strip().splitlines()[-1])
    assert res["ranges"] == [[0, 501], [501, 500]] and res["tmax"] == 1.5


def test_create_frequency_masks_golden():
    """fresca.py:13-108: pure host helper of the fdiff.utils.fresca mirror, pinned against the reference (g11)."""
    import numpy as np
    import torch

    from conftest import load_golden
    from fastfourierdiffusion_amd.utils import synthetic
    from fastfourierdiffusion_amd.utils.fresca import create_frequency_masks
    from oracle import cases

    g = load_golden("g11_extra_traj.npz")
    for (name, shape, ratio, strat, seed) in cases.MASK_CASES:
        spec = None if seed is None else torch.from_numpy(np.abs(next(synthetic.noise_stream(shape, 1, seed))))
        lo, hi = create_frequency_masks(shape, ratio, strat, spec)
        np.testing.assert_array_equal(lo.numpy(), g[name + "_low"])
        np.testing.assert_array_equal(hi.numpy(), g[name + "_high"])
    with pytest.raises(ValueError):
        create_frequency_masks((8,), 0.5, "energy", None)
    with pytest.raises(ValueError):
        create_frequency_masks((2, 2, 2), 0.5)


def test_hermite_polynomials_known_answers():
    """fourier.py:341-394: H0 = 1, H1 = 2s, H2 = 4s^2 - 2, H3 = 8s^3 - 12s, then H_{n+1} = 2s H_n - 2n H_{n-1}."""
    from fastfourierdiffusion_amd.utils.fourier import hermite_polynomials

    s = torch.tensor([-1.0, -0.5, 0.0, 0.25, 1.0])
    H = hermite_polynomials(s, 4)
    assert H.shape == (5, 5)
    torch.testing.assert_close(H[0], torch.ones(5))
    torch.testing.assert_close(H[1], 2 * s)
    torch.testing.assert_close(H[2], 4 * s ** 2 - 2)
    torch.testing.assert_close(H[3], 8 * s ** 3 - 12 * s)
    torch.testing.assert_close(H[4], 16 * s ** 4 - 48 * s ** 2 + 12)
    assert hermite_polynomials(torch.zeros(2, 3), 2).shape == (3, 2, 3)
