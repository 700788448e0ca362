"""Case tables shared by oracle/gen_golden.py and tests/ (TEST INFRASTRUCTURE).

Pure data: shapes, seeds and hyper-parameters of every golden vector.  Inputs
are regenerated from the seeds (fastfourierdiffusion_amd.utils.synthetic); the
.npz files under tests/golden hold only the reference's outputs.
"""
from __future__ import annotations

VP = {"beta_min": 0.1, "beta_max": 20.0}          # cmd/conf/score_model/noise_scheduler/vpsde.yaml
VE = {"sigma_min": 0.01, "sigma_max": 2.0}        # .../vesde.yaml

# (L, C, B, seed)  -- L=100/101 are the reference's own test sizes (tests/test_utils.py:7-9)
FFT_CASES = [(100, 3, 4, 11), (101, 3, 4, 12), (187, 1, 4, 13), (251, 4, 2, 14), (512, 8, 2, 15),
             (20, 3, 2, 16), (50, 3, 2, 17)]

TABLE_LENS = [20, 50, 100, 101, 187, 251, 512]
TABLE_STEPS = [6, 8, 10, 20, 50, 100, 120, 1000]

STEP_CASES = [
    dict(name="vp_ecg", sde="vp", sde_kwargs=VP, fourier=True, L=187, C=1, B=4, N=1000, idx=[0, 500, 999], seed=21),
    dict(name="ve_ecg", sde="ve", sde_kwargs=VE, fourier=True, L=187, C=1, B=4, N=1000, idx=[0, 500, 999], seed=22),
    dict(name="vp_nasa", sde="vp", sde_kwargs=VP, fourier=True, L=251, C=4, B=2, N=1000, idx=[0, 999], seed=23),
    dict(name="vp_syn", sde="vp", sde_kwargs=VP, fourier=True, L=512, C=8, B=2, N=50, idx=[0, 25, 49], seed=24),
    dict(name="vp_time", sde="vp", sde_kwargs=VP, fourier=False, L=187, C=1, B=4, N=50, idx=[0, 49], seed=25),
    dict(name="ve_even", sde="ve", sde_kwargs=VE, fourier=True, L=100, C=3, B=2, N=10, idx=[0, 9], seed=26),
]

_SMALL = dict(kind="transformer", d=24, H=4, NL=2, L=20, C=3)          # hd = 6
_REFUNIT = dict(kind="transformer", d=8, H=4, NL=2, L=20, C=3)          # tests/test_score_models.py:13-19 (hd = 2)
_REFUNIT_LSTM = dict(kind="lstm", d=8, H=1, NL=2, L=20, C=3)
_REFTEST = dict(kind="transformer", d=60, H=12, NL=3, L=50, C=3)       # tests/test_sampling.py:7-11 (hd = 5)
_ECG = dict(kind="transformer", d=72, H=12, NL=10, L=187, C=1)         # cmd/conf/score_model/default.yaml
_SYN = dict(kind="transformer", d=72, H=12, NL=10, L=512, C=8)         # BASELINE configs[4] shape
_NASA_LSTM = dict(kind="lstm", d=72, H=1, NL=10, L=251, C=4)           # cmd/conf/score_model/lstm.yaml
_SMALL_LSTM = dict(kind="lstm", d=24, H=1, NL=2, L=20, C=3)

MODEL_CASES = [
    dict(name="small", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=3, wseed=42, xseed=31,
         t_values=[1.0, 0.5], cache_seq=[list(range(20)), [], list(range(10)), [], list(range(17))],
         dump_table=True),
    dict(name="reftest", **_REFTEST, sde="vp", sde_kwargs=VP, fourier=True, B=2, wseed=43, xseed=32,
         t_values=[0.7], cache_seq=[list(range(50)), [], list(range(10)), []], dump_table=False),
    dict(name="ecg", **_ECG, sde="vp", sde_kwargs=VP, fourier=True, B=2, wseed=42, xseed=33,
         t_values=[1.0, 0.25], cache_seq=[list(range(187)), [], list(range(10)), []], dump_table=False),
    dict(name="syn", **_SYN, sde="vp", sde_kwargs=VP, fourier=True, B=1, wseed=44, xseed=34,
         t_values=[0.9], cache_seq=[list(range(512)), [], list(range(10))], dump_table=False),
    dict(name="refunit", **_REFUNIT, sde="vp", sde_kwargs=VP, fourier=True, B=5, wseed=47, xseed=37,
         t_values=[0.5], cache_seq=[list(range(20)), [], list(range(10))], dump_table=True),
    dict(name="refunit_lstm", **_REFUNIT_LSTM, sde="vp", sde_kwargs=VP, fourier=True, B=5, wseed=48, xseed=38,
         t_values=[0.5]),
    dict(name="nasa_lstm", **_NASA_LSTM, sde="vp", sde_kwargs=VP, fourier=True, B=2, wseed=45, xseed=35,
         t_values=[1.0, 0.3]),
    dict(name="small_lstm", **_SMALL_LSTM, sde="vp", sde_kwargs=VP, fourier=True, B=3, wseed=46, xseed=36,
         t_values=[0.6]),
]

TRAJ_CASES = [
    # small model: every combination, incl. multi-batch cache semantics (Q3) and an in-trajectory refresh (R=100)
    dict(name="traj_small_vp", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=3, num_samples=3, N=8,
         use_cache=False, wseed=42, zseed=51),
    dict(name="traj_small_ve", **_SMALL, sde="ve", sde_kwargs=VE, fourier=True, B=3, num_samples=3, N=8,
         use_cache=False, wseed=42, zseed=52),
    dict(name="traj_small_vp_cache", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=3, num_samples=3, N=8,
         use_cache=True, cache_kwargs={}, wseed=42, zseed=53),
    dict(name="traj_small_vp_cache_2batches", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=2,
         num_samples=5, N=6, use_cache=True, cache_kwargs={}, wseed=42, zseed=54),
    dict(name="traj_small_vp_cache_R100", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=2,
         num_samples=4, N=120, use_cache=True, cache_kwargs={"K": 3, "R": 100}, wseed=42, zseed=55),
    dict(name="traj_small_fresca", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=3, num_samples=3, N=8,
         use_cache=False, wseed=42, zseed=62, fresca=dict(low_scale=1.0, high_scale=1.5, cutoff_ratio=0.5,
                                                          cutoff_strategy="energy")),
    dict(name="traj_small_fresca_cache_spatial", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=2,
         num_samples=2, N=8, use_cache=True, cache_kwargs={}, wseed=42, zseed=63,
         fresca=dict(low_scale=0.9, high_scale=1.2, cutoff_ratio=0.4, cutoff_strategy="spatial")),
    dict(name="traj_refunit_ve", **_REFUNIT, sde="ve", sde_kwargs=VE, fourier=True, B=5, num_samples=5, N=10,
         use_cache=False, wseed=47, zseed=64),
    dict(name="traj_small_time", **_SMALL, sde="vp", sde_kwargs=VP, fourier=False, B=2, num_samples=2, N=8,
         use_cache=False, wseed=42, zseed=56),
    dict(name="traj_reftest_vp", **_REFTEST, sde="vp", sde_kwargs=VP, fourier=True, B=2, num_samples=2, N=10,
         use_cache=False, wseed=43, zseed=57),
    dict(name="traj_small_lstm", **_SMALL_LSTM, sde="vp", sde_kwargs=VP, fourier=True, B=2, num_samples=2,
         N=8, use_cache=False, wseed=46, zseed=58),
    # BASELINE configs[0]-like: ECG time-domain, 50 steps (B reduced to 2 to keep the fixture small)
    dict(name="traj_ecg_time_50", **_ECG, sde="vp", sde_kwargs=VP, fourier=False, B=2, num_samples=2, N=50,
         use_cache=False, wseed=42, zseed=59),
    # BASELINE configs[1]/[2]: ECG freq-domain, the full 1000 steps, cache off / on
    dict(name="traj_ecg_1000", **_ECG, sde="vp", sde_kwargs=VP, fourier=True, B=2, num_samples=2, N=1000,
         use_cache=False, wseed=42, zseed=60),
    dict(name="traj_ecg_1000_cache", **_ECG, sde="vp", sde_kwargs=VP, fourier=True, B=2, num_samples=2,
         N=1000, use_cache=True, cache_kwargs={}, wseed=42, zseed=60),
    dict(name="traj_nasa_lstm_20", **_NASA_LSTM, sde="vp", sde_kwargs=VP, fourier=True, B=2, num_samples=2,
         N=20, use_cache=False, wseed=45, zseed=61),
]

# FreSca on a raw score tensor: (name, L, C, B, seed, low, high, cutoff_ratio, strategy, timestep, num_steps)
FRESCA_CASES = [
    ("ecg_energy", 187, 1, 6, 71, 1.0, 1.5, 0.5, "energy", None, None),
    ("ecg_energy_dyn", 187, 1, 6, 72, 1.0, 1.5, 0.5, "energy", 0.73, 100),
    ("ecg_spatial", 187, 1, 4, 73, 0.8, 1.3, 0.25, "spatial", None, None),
    ("nasa_energy", 251, 4, 3, 74, 0.9, 2.0, 0.7, "energy", 0.2, 1000),
    ("even_energy", 100, 3, 5, 75, 1.2, 0.7, 0.3, "energy", None, None),
    ("syn_spatial", 512, 8, 2, 76, 1.0, 1.5, 0.5, "spatial", 1.0, 1000),
    ("syn_energy", 512, 8, 2, 77, 1.1, 1.4, 0.9, "energy", None, None),
    ("identity", 50, 3, 2, 78, 1.0, 1.0, 0.5, "energy", None, None),
]
# the 4-D branch of frequency_scale (fresca.py:184-213): (name, H, W, C, B, seed, low, high, ratio, strategy)
FRESCA2D_CASES = [
    ("img_energy", 16, 16, 3, 4, 91, 1.0, 1.5, 0.5, "energy"),
    ("img_spatial", 16, 12, 2, 3, 92, 0.8, 1.3, 0.5, "spatial"),
    ("odd_energy", 9, 15, 1, 2, 93, 1.2, 0.7, 0.3, "energy"),
    ("wide_spatial", 8, 64, 4, 2, 94, 1.0, 1.5, 0.25, "spatial"),
    ("tall_energy", 48, 10, 2, 2, 95, 0.9, 2.0, 0.8, "energy"),
    ("full_energy_never", 6, 6, 2, 2, 96, 0.5, 1.5, 1.0, "energy"),  # no disc holds all the energy: Rc stays 0
    ("max_size", 64, 64, 1, 1, 97, 1.1, 1.4, 0.6, "energy"),
    ("identity2d", 8, 8, 2, 2, 98, 1.0, 1.0, 0.5, "energy"),
]
FRESCA_DEFAULT = dict(low_scale=1.0, high_scale=1.5, cutoff_ratio=0.5, cutoff_strategy="energy")  # benchmark_cache.py:63-68

# (K, R, L, steps)
_STEPS = [0, 1, 2, 5, 10, 99, 100, 150, 200, 300, 499, 500, 501, 999, 1000, 1500]
GATE_CASES = [(5, 10, 187, _STEPS), (0, 10, 187, _STEPS), (3, 100, 187, _STEPS), (1, 150, 187, _STEPS),
              (5, 500, 20, _STEPS), (200, 10, 187, _STEPS), (80, 10, 187, _STEPS), (5, 10, 8, _STEPS)]

# FreqCa helpers (next row (f)2).  (name, B, L, D, seed, low_freq_ratio); B == 0 -> 2-D (L, D) input
DECOMP_CASES = [
    ("crf_ecg", 3, 187, 72, 81, 0.3),
    ("crf_small", 2, 20, 24, 82, 0.3),
    ("even", 3, 100, 16, 83, 0.5),
    ("nlow1", 2, 64, 8, 84, 0.0),
    ("two_d", 0, 50, 8, 85, 0.3),
    ("syn", 2, 512, 72, 86, 0.25),
]
# (name, K, shape, order, timesteps, target, seed)
HERMITE_CASES = [
    ("h_order3", 6, (4, 20, 8), 3, [1.0, 0.9, 0.8, 0.7, 0.6, 0.5], 0.45, 91),
    ("h_order2", 4, (20, 8), 2, [0.8, 0.6, 0.4, 0.2], 0.1, 92),
    ("h_inside", 10, (2, 187, 72), 3, [1.0 - 0.01 * i for i in range(10)], 0.955, 93),
    ("h_single", 1, (20, 8), 3, [0.5], 0.4, 94),
    ("h_same_t", 3, (20, 8), 3, [0.5, 0.5, 0.5], 0.4, 95),
]
# (L, C, B, seed, apply_dft)
DENSITY_CASES = [(187, 1, 4, 96, True), (100, 3, 2, 97, True), (251, 4, 2, 98, False), (512, 8, 2, 99, True)]
# sampler runs with E2CRFCache(use_freqca=True): the cache object's FreqCa state after sample()
FREQCA_TRAJ_CASES = [
    dict(name="freqca_small", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=2, num_samples=4, N=23,
         use_cache=True, cache_kwargs=dict(use_freqca=True, freq_decomp="fft", freq_decomp_interval=4, max_history=3,
                                           hermite_order=2, low_freq_ratio=0.3), wseed=42, zseed=65, t_pred=0.05),
    dict(name="freqca_small_dct", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=3, num_samples=3, N=50,
         use_cache=True, cache_kwargs=dict(use_freqca=True), wseed=42, zseed=66, t_pred=0.5),
    dict(name="nofreqca_small_crf", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=3, num_samples=3, N=25,
         use_cache=True, cache_kwargs={}, wseed=42, zseed=67, t_pred=None),
]

# MLPScoreModule (scope row (f)3).  PARITY UNPINNED against the reference: torchvision.ops.MLP is not importable here,
# so no golden exists; the product is checked against the oracle restatement (oracle.mlp_score_forward) only.
MLP_CASES = [
    dict(name="refunit_mlp", kind="mlp", d=8, d_mlp=512, NL=2, L=20, C=3, H=1, B=5, wseed=49, xseed=39),    # tests/test_score_models.py:13-19,36
    dict(name="ecg_mlp", kind="mlp", d=72, d_mlp=1024, NL=10, L=187, C=1, H=1, B=4, wseed=50, xseed=40),       # cmd/conf/score_model/mlp.yaml
    dict(name="nasa_mlp", kind="mlp", d=72, d_mlp=512, NL=3, L=251, C=4, H=1, B=7, wseed=51, xseed=41),      # L*C % 4 == 0
    dict(name="odd_mlp", kind="mlp", d=13, d_mlp=70, NL=1, L=9, C=2, H=1, B=33, wseed=52, xseed=42),         # ragged everything
]

# More sampler combinations pinned against the reference (g11): VE / time-domain with the cache and FreSca together
EXTRA_TRAJ_CASES = [
    dict(name="x_ve_cache_fresca", **_SMALL, sde="ve", sde_kwargs=VE, fourier=True, B=2, num_samples=4, N=104,
         use_cache=True, cache_kwargs={"K": 3, "R": 100}, wseed=42, zseed=71,
         fresca=dict(low_scale=0.9, high_scale=1.4, cutoff_ratio=0.5, cutoff_strategy="energy")),
    dict(name="x_vp_time_cache", **_SMALL, sde="vp", sde_kwargs=VP, fourier=False, B=2, num_samples=4, N=104,
         use_cache=True, cache_kwargs={"K": 3, "R": 100}, wseed=42, zseed=72),
    dict(name="x_ve_time_fresca_spatial", **_SMALL, sde="ve", sde_kwargs=VE, fourier=False, B=3, num_samples=3, N=20,
         use_cache=False, wseed=42, zseed=73,
         fresca=dict(low_scale=1.1, high_scale=0.8, cutoff_ratio=0.3, cutoff_strategy="spatial")),
    dict(name="x_reftest_vp_cache", **_REFTEST, sde="vp", sde_kwargs=VP, fourier=True, B=2, num_samples=2, N=12,
         use_cache=True, cache_kwargs={}, wseed=43, zseed=74),
]


# ---- round 2 (g12) -----------------------------------------------------------------------------------
# ScoreModule.forward with PER-SAMPLE timesteps (time_encoder(X, timesteps), score_models.py:102); "int" = the
# reference's own unit test draws integer timesteps with torch.randint(0, n_diffusion_steps) (tests/test_score_models.py:70)
MIXED_T_CASES = [
    dict(name="mt_refunit", **_REFUNIT, sde="vp", sde_kwargs=VP, fourier=True, B=5, wseed=47, xseed=121, t=[3, 0, 7, 9, 1], int_t=True),
    dict(name="mt_refunit_lstm", **_REFUNIT_LSTM, sde="vp", sde_kwargs=VP, fourier=True, B=5, wseed=48, xseed=122, t=[3, 0, 7, 9, 1], int_t=True),
    dict(name="mt_small", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=4, wseed=42, xseed=123, t=[1.0, 0.61, 0.25, 1e-5], int_t=False,
         recompute=[list(range(20)), [], list(range(10))]),
    dict(name="mt_ecg", **_ECG, sde="vp", sde_kwargs=VP, fourier=True, B=3, wseed=42, xseed=124, t=[0.9, 0.5, 0.1], int_t=False),
    dict(name="mt_nasa_lstm", **_NASA_LSTM, sde="vp", sde_kwargs=VP, fourier=True, B=2, wseed=45, xseed=125, t=[0.8, 0.2], int_t=False),
]
# BASELINE configs[4] shape, short cached trajectory
SYN_TRAJ_CASES = [
    dict(name="traj_syn_cache", **_SYN, sde="vp", sde_kwargs=VP, fourier=True, B=2, num_samples=2, N=8, use_cache=True,
         cache_kwargs={}, wseed=44, zseed=131),
]
# two samplers on ONE model with different gate parameters (cmd/benchmark_cache.py:274-311 re-uses the model like this):
# the second sampler must run ITS cache's K / R although the layers stay bound to the first cache (Q5)
TWO_SAMPLER_CASE = dict(name="two_samplers", **_SMALL, sde="vp", sde_kwargs=VP, fourier=True, B=2, num_samples=4, N=104,
                        first_kwargs={}, second_kwargs={"K": 10, "R": 100}, wseed=42, zseed1=132, zseed2=133)
# cmd/sample.py:107-113 (X * std + mean, then idft) and its ingest twin datamodules.py:42-62 ((dft(X) - mean) / std).
# (L, C, B, seed)
AFFINE_FFT_CASES = [(187, 1, 4, 141), (251, 4, 3, 142), (100, 3, 2, 143), (512, 8, 2, 144)]


# ---- round 4 (g13) -----------------------------------------------------------------------------------
# BASELINE configs[3] at its full length (NASA charge, LSTM backbone, 1000 steps), the reference's CLASS-default
# transformer (score_models.py:31-33: d_model 60, 3 layers, 12 heads) at the ECG length, and the reference's default
# sample_batch_size = 50 (cmd/conf/sampler/default.yaml:3) with the cache on.
_D60_ECG = dict(kind="transformer", d=60, H=12, NL=3, L=187, C=1)
ROUND4_TRAJ_CASES = [
    dict(name="traj_nasa_lstm_1000", **_NASA_LSTM, sde="vp", sde_kwargs=VP, fourier=True, B=2, num_samples=2, N=1000,
         use_cache=False, wseed=45, zseed=151),
    dict(name="traj_d60_ecg_100", **_D60_ECG, sde="vp", sde_kwargs=VP, fourier=True, B=3, num_samples=3, N=100,
         use_cache=False, wseed=53, zseed=152),
    dict(name="traj_d60_ecg_cache_30", **_D60_ECG, sde="ve", sde_kwargs=VE, fourier=True, B=2, num_samples=4, N=30,
         use_cache=True, cache_kwargs={}, wseed=53, zseed=153),
    dict(name="traj_ecg_b50_cache_12", **_ECG, sde="vp", sde_kwargs=VP, fourier=True, B=50, num_samples=50, N=12,
         use_cache=True, cache_kwargs={}, wseed=42, zseed=154),
]
