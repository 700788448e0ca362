/*
 * libffd -- C ABI of the MI355X-native frequency-domain diffusion sampler.
 *
 * This is the drop-in boundary for the sampling hot path of
 * NoakLiu/FastFourierDiffusion (`fdiff`).  The reference has no FFI layer of its
 * own (it is pure Python on stock PyTorch ops), so each entry point below names
 * the reference Python interface it replaces (file:line relative to the
 * reference repository root).  Signatures use only plain C types: pointers are
 * raw device (HBM) or host addresses, sizes are explicit, streams are passed as
 * `void*` (a `hipStream_t`; NULL = the default stream).  No torch types.
 *
 * Conventions
 *   - every function returns 0 on success and a negative ffd_status on failure;
 *     it never throws and never calls exit(); `ffd_last_error(ctx)` returns a
 *     human-readable message for the most recent failure on that context;
 *   - all device work is stream-ordered on the caller's stream: the functions
 *     enqueue kernels and return without synchronising (exceptions are noted);
 *   - I/O buffers are caller-owned and only borrowed for the stream-ordered
 *     call; weights are copied (and re-packed) into context-owned HBM;
 *   - one context per device; a context is not thread-safe, distinct contexts
 *     are independent: the only process-wide state is a per-device cache of FFT
 *     twiddle tables (the experiment knobs of ffd_tune -- kernel choice / tiling,
 *     never results -- are per calling thread);
 *   - tensors are dense row-major fp32: series X/score (B, L, C) with C
 *     innermost and L the Fourier / attention axis (score_models.py:87-90,
 *     fourier.py:12,24).
 */
#ifndef FFD_H
#define FFD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ffd_ctx ffd_ctx;

typedef enum {
  FFD_OK = 0,
  FFD_ERR_INVALID = -1,     /* bad argument / shape (the reference's `assert`s, score_models.py:87-93) */
  FFD_ERR_UNSUPPORTED = -2, /* configuration this build has no kernel for (reference: NotImplementedError) */
  FFD_ERR_STATE = -3,       /* call order (e.g. forward before weights are finalised) */
  FFD_ERR_HIP = -4,         /* a HIP runtime call failed; message carries hipGetErrorString */
  FFD_ERR_NOMEM = -5
} ffd_status;

enum { FFD_MODEL_TRANSFORMER = 0, FFD_MODEL_LSTM = 1, FFD_MODEL_MLP = 2 };
enum { FFD_SDE_VP = 0, FFD_SDE_VE = 1 };

/* Model + scheduler hyper-parameters.
 * Replaces the constructor arguments of ScoreModule / LSTMScoreModule / MLPScoreModule
 * (src/fdiff/models/score_models.py:25-37, 444-455, 364-376) and of VPScheduler /
 * VEScheduler (src/fdiff/schedulers/sde.py:91-97, 169-175). */
typedef struct {
  int32_t kind;            /* FFD_MODEL_* */
  int32_t n_channels;      /* C */
  int32_t max_len;         /* L */
  int32_t d_model;         /* d   (8, 16, 24, 32, 48, 60, 64 or 72 in this build; d / n_head in {2,3,4,5,6,8}; any d for mlp) */
  int32_t n_head;          /* H   (ignored for lstm) */
  int32_t num_layers;      /* NL */
  int32_t dim_feedforward; /* F   (PyTorch default 2048, score_models.py:61-63); multiple of 64.  mlp: d_mlp (any >= 1) */
  int32_t sde;             /* FFD_SDE_* */
  double sde_a;            /* VP: beta_min   | VE: sigma_min */
  double sde_b;            /* VP: beta_max   | VE: sigma_max */
  int32_t fourier_noise_scaling; /* SDE.noise_scaling (sde.py:16,49) */
  double eps;              /* SDE.eps (sde.py:16) */
} ffd_model_desc;

/* E2-CRF cache configuration: the E2CRFCache kwargs the gate actually reads
 * (src/fdiff/utils/caching.py:28-47,131-181: K and R; tau_0, tau_warn, ... are
 * stored but never read by the reference). */
typedef struct {
  int32_t K;
  int32_t R;
} ffd_cache_cfg;

/* E2CRFCache.stats / get_cache_stats counters (caching.py:107-111, 599-653). */
typedef struct {
  int64_t recompute_count;
  int64_t cache_hit_count;
  int64_t current_step;
  int32_t table_allocated; /* k_cache_tensor is not None */
  int32_t reserved;
} ffd_cache_stats;

/* ---- lifetime ---------------------------------------------------------- */

/* Build a context on HIP device `device`.  Replaces module construction +
 * `.cuda()` (cmd/sample.py:68-77).  Fails with FFD_ERR_HIP if no gfx950 device
 * is usable: there is no CPU fallback. */
int ffd_create(ffd_ctx** out, const ffd_model_desc* desc, int device);
void ffd_destroy(ffd_ctx* ctx);
const char* ffd_last_error(const ffd_ctx* ctx);
/* Static description of the build ("libffd <ver> gfx950 ..."), usable without a device. */
const char* ffd_version(void);

/* ---- weights ----------------------------------------------------------- */

/* Copy one parameter (host or device pointer, `n` floats) into the context.
 * `name` is the reference state_dict key:
 *   pos_encoder.embedding.weight (L,d)   time_encoder.W ((d+1)/2)
 *   time_encoder.dense.{weight (d,d),bias}  embedder.{weight (d,C),bias}
 *   unembedder.{weight (C,d),bias}
 *   backbone.layers.{i}.self_attn.{in_proj_weight (3d,d),in_proj_bias,out_proj.weight,out_proj.bias}
 *   backbone.layers.{i}.{linear1,linear2}.{weight,bias}  backbone.layers.{i}.{norm1,norm2}.{weight,bias}
 *   backbone.{i}.{weight_ih_l0 (4d,d),weight_hh_l0 (4d,d),bias_ih_l0,bias_hh_l0}   (lstm)
 * Replaces nn.Module.load_state_dict / load_from_checkpoint (cmd/sample.py:68-75). Synchronous. */
int ffd_load_weight(ffd_ctx* ctx, const char* name, const float* data, size_t n);
/* Validate that every parameter was supplied, apply the nn.Embedding(max_norm)
 * renormalisation to its fixed point (transformer.py:13-15) and
 * re-pack GEMM weights into MFMA fragment order.  Synchronous. */
int ffd_finalize_weights(ffd_ctx* ctx);

/* ---- scheduler tables (host only; no device needed) --------------------- */

/* SDE.set_noise_scaling (sde.py:42-60): writes G[0..L). */
int ffd_host_noise_scaling(int max_len, int fourier_noise_scaling, float* G_out);
/* SDE.set_timesteps (sde.py:62-64): fp32 linspace(1, eps, n) and step_size = t[0]-t[1]. */
int ffd_host_timesteps(int n, double eps, float* ts_out, float* step_size_out);
/* E2CRFCache.determine_recompute_set (caching.py:131-181): the recompute set is
 * always the prefix [0, n); returns n (>=0) for `step`. */
int ffd_host_gate(int step, int max_len, int K, int R);

/* ---- single operators (device pointers, stream ordered) ---------------- */

/* ScoreModule.forward / LSTMScoreModule.forward (score_models.py:79-119, 486-511)
 * without cache: score_out (B,L,C) <- model(x (B,L,C), t). All samples share `t`
 * (sampler.py:59-60). */
int ffd_score_forward(ffd_ctx* ctx, const float* x, float t, float* score_out, int B, void* stream);

/* ScoreModule.forward(batch, recompute_tokens, step, return_crf=True)
 * (score_models.py:79-194 + CachedTransformerEncoderLayer.forward,
 * cached_transformer.py:106-329).  `n_recompute` is |recompute_tokens| (always
 * the prefix [0,n), see ffd_host_gate).  Maintains the context's KV table
 * (caching.py:302-396) and counters.  crf_out (NL,L,d) may be NULL. */
int ffd_score_forward_cached(ffd_ctx* ctx, const float* x, float t, float* score_out, float* crf_out,
                             int B, int n_recompute, void* stream);

/* The same two forwards with PER-SAMPLE diffusion times: the reference evaluates
 * time_encoder(X, timesteps) per sample (score_models.py:102, transformer.py:77-91) and its own
 * unit test calls forward with mixed timesteps (tests/test_score_models.py:70).  `timesteps` is
 * a (B) fp32 device array; no host synchronisation.  n_recompute < 0: no cache (ffd_score_forward);
 * otherwise the cached forward for |recompute_tokens| = n_recompute (crf_out may be NULL). */
int ffd_score_forward_ts(ffd_ctx* ctx, const float* x, const float* timesteps, float* score_out, float* crf_out,
                         int B, int n_recompute, void* stream);

/* Scheduler hyper-parameters for the context-free scheduler operators below
 * (VPScheduler / VEScheduler constructor arguments, sde.py:91-97, 169-175). */
typedef struct {
  int32_t sde;      /* FFD_SDE_* */
  int32_t reserved;
  double a;         /* VP: beta_min  | VE: sigma_min */
  double b;         /* VP: beta_max  | VE: sigma_max */
} ffd_sde_desc;

/* VPScheduler.step / VEScheduler.step (sde.py:129-165, 215-246), in place on x (B,L,C).
 * Context-free, like the reference scheduler objects: G (L floats, device) is the
 * scheduler's noise scaling (ffd_host_noise_scaling uploaded by the caller).
 * `t` is timesteps[i] widened to double (sampler.py:96-98).  z (B,L,C) supplies the
 * N(0,1) draw; if NULL it is generated on device by Philox4x32-10 keyed by
 * (seed, step) and counted by the global element index (sample_offset*L*C + i), so
 * results do not depend on how a batch is sharded across GPUs. */
int ffd_sde_step(const ffd_sde_desc* sde, float* x, const float* score, const float* G, double t,
                 float step_size, const float* z, uint64_t seed, uint64_t sample_offset, int step, int B,
                 int L, int C, void* stream);

/* SDE.prior_sampling (sde.py:79-87, 125-127): x <- G (.) z  (VE: * sigma_max);
 * z == NULL draws on device (Philox stream tag 0xFFFFFFFF). */
int ffd_prior(const ffd_sde_desc* sde, float* x, const float* z, const float* G, uint64_t seed,
              uint64_t sample_offset, int B, int L, int C, void* stream);

/* fdiff.utils.fourier.dft / idft (fourier.py:8-52, 55-94): packed ortho rFFT and
 * its inverse along dim 1 of (B,L,C).  `in` and `out` may not alias. Context-free. */
int ffd_dft(const float* in, float* out, int B, int L, int C, void* stream);
int ffd_idft(const float* in, float* out, int B, int L, int C, void* stream);
/* The runner-side wrappers around them, fused into the same kernel (no extra pass over the data):
 *   ffd_unstandardize_idft  cmd/sample.py:107-113   out = idft(in * std + mean)   (sampled spectra -> time series)
 *   ffd_dft_standardize     src/fdiff/dataloaders/datamodules.py:42-43,61-62   out = (dft(in) - mean) / std
 * mean / std: (L, C) fp32 on the device (DiffusionDataset.feature_mean / feature_std), broadcast over B. */
int ffd_unstandardize_idft(const float* in, float* out, const float* mean, const float* std, int B, int L, int C,
                           void* stream);
int ffd_dft_standardize(const float* in, float* out, const float* mean, const float* std, int B, int L, int C,
                        void* stream);

/* PositionalEncoding.forward (transformer.py:17-29): out = x + weight[arange(L)] for x (B,L,D); like
 * nn.Embedding(max_norm) the looked-up rows of `weight` (L,D, device) are renormalised IN PLACE first
 * (rows with ||w|| > max_norm scaled by max_norm/(||w||+1e-7); max_norm <= 0 disables it). */
int ffd_positional_encoding(const float* x, float* weight, float* out, int B, int L, int D, float max_norm,
                            void* stream);
/* GaussianFourierProjection.forward (transformer.py:77-91), use_time_axis=True:
 * out[b,l,:] = x[b,l,:] + dense([sin(2 pi t_b W), cos(2 pi t_b W)][:D]); timesteps (B) fp32 on device,
 * temb_work = B*D floats of scratch.  L = 1 gives the use_time_axis=False form. */
int ffd_time_encoding(const float* x, const float* timesteps, const float* W, const float* dense_w,
                      const float* dense_b, float* temb_work, float* out, int B, int L, int D, void* stream);

/* FreSca spectral scaling (fdiff.utils.fresca.frequency_scale / apply_fresca_to_score,
 * fresca.py:111-268, 3-D case): out = irfft((low*[k<=Rc] + high*[k>Rc]) (.) rfft(in)) along dim 1.
 * strategy 0 "spatial": Rc = cutoff_ratio * (L/2+1); 1 "energy": Rc = first k whose cumulative
 * batch-mean |X_k| reaches cutoff_ratio * total (fresca.py:46-58; a batch-wide statistic, reduced
 * on the device in a fixed order -- no host sync).  `work` = B*C*(L/2+1) + 4 floats of scratch.
 * Context-free; in != out.  low == high == 1 is the caller's early exit (fresca.py:137-138). */
enum { FFD_FRESCA_SPATIAL = 0, FFD_FRESCA_ENERGY = 1 };
int ffd_fresca(const float* in, float* out, float* work, int B, int L, int C, float low_scale, float high_scale,
               double cutoff_ratio, int strategy, void* stream);

/* The 4-D branch of frequency_scale (fresca.py:184-213): in / out (B, H, W, C), rfft2 / irfft2 over (H, W), ortho;
 * low-pass mask [sqrt(kh^2 + kw^2) <= Rc] on the raw bin indices of the (H, W/2+1) half spectrum (fresca.py:73-81);
 * "spatial": Rc = cutoff_ratio * min(H/2, (W/2+1)/2); "energy": the first integer radius R in 0..int(min(H, W/2+1)/2)
 * whose disc holds cutoff_ratio of the batch-and-channel mean |X| (fresca.py:89-101), 0 if none does.  Not on the
 * sampling path (scores are 3-D); H, W <= 256 and H*W <= 4096, else FFD_ERR_UNSUPPORTED.
 * `work` = 3*B*C*H*(W/2+1) + 4 floats of scratch (8-byte aligned).  Context-free; in != out. */
int ffd_fresca2d(const float* in, float* out, float* work, int B, int H, int W, int C, float low_scale,
                 float high_scale, double cutoff_ratio, int strategy, void* stream);

/* DiffusionSampler(use_fresca=True, ...) (sampler.py:21-26,79-93): apply FreSca to every score
 * inside ffd_sample_batch, with the reference's time-dependent high scale
 * h(t) = (1 - t/num_steps)*(h-1) + 1 for h > 1 (fresca.py:247-257; num_steps <= 0: static h). */
typedef struct {
  float low_scale, high_scale;
  double cutoff_ratio;
  int32_t strategy;   /* FFD_FRESCA_* */
  int32_t num_steps;  /* the sampler's _num_diffusion_steps, or 0 when unset */
} ffd_fresca_cfg;
int ffd_fresca_enable(ffd_ctx* ctx, const ffd_fresca_cfg* cfg);
int ffd_fresca_disable(ffd_ctx* ctx);

/* ---- FreqCa helpers (E2CRFCache(use_freqca=True), caching.py:486-522,561-597) ---- */

/* frequency_decompose_fft / frequency_decompose_dct (fourier.py:219-286; the dct variant returns the
 * fft result, fourier.py:303): low = irfft(rfft(x)[k < n_low]), high = irfft(rfft(x)[k >= n_low]) along
 * dim 1 of x (B, L, D), n_low = max(1, int((L/2+1) * low_freq_ratio)), ortho norm.  low/high must not
 * alias x. */
int ffd_freq_decompose(const float* x, float* low, float* high, int B, int L, int D, double low_freq_ratio,
                       void* stream);

/* predict_hermite (fourier.py:397-497): least-squares fit of Hermite polynomials H_0..H_order over the K
 * history points (timesteps normalised to [-1,1], ridge 1e-6) evaluated at `target`.
 * history: device (K, n) stacked tensors; timesteps: host K doubles; out: device n floats.
 * K < 2 or equal timesteps return history[K-1] (fourier.py:416-428).  K <= 32, order <= 8. */
int ffd_hermite_predict(const float* history, const double* timesteps, double target, int order, float* out, int K,
                        size_t n, void* stream);

/* spectral_density (fourier.py:97-131) of a PACKED spectrum xf (B, L, C) -> out (B, L/2+1, C);
 * callers with time-domain input run ffd_dft first (apply_dft=True). */
int ffd_spectral_density(const float* xf, float* out, int B, int L, int C, void* stream);

/* CRF capture inside ffd_sample_batch (the reference's cache.update_crf call, sampler.py:70-73):
 * on cached steps whose global step g satisfies g % every == 0 the (NL, L, d) CRF (score_models.py:181-194)
 * is written to ring slot (g / every) % n_slots; `last` receives the CRF of the last step of each
 * ffd_sample_batch call with g % last_every == 0 (E2CRFCache.crf_cache, caching.py:474-484).  Either pointer may
 * be NULL; cfg == NULL switches capture off.  Buffers are caller-owned device memory. */
typedef struct {
  float* ring;
  int32_t n_slots;
  int32_t every;
  float* last;
  int32_t last_every;
  int32_t reserved;
} ffd_crf_capture_cfg;
int ffd_cache_crf_capture(ffd_ctx* ctx, const ffd_crf_capture_cfg* cfg);

/* ---- the sampling loop -------------------------------------------------- */

/* E2CRFCache lifecycle used by DiffusionSampler (sampler.py:37-39,151-153):
 * enable/disable (score_models.py:202-289), reset (caching.py:115-129). */
int ffd_cache_enable(ffd_ctx* ctx, const ffd_cache_cfg* cfg);
int ffd_cache_disable(ffd_ctx* ctx);
int ffd_cache_reset(ffd_ctx* ctx);
/* K and R the gate of ffd_sample_batch uses from now on, WITHOUT touching tables or counters: the
 * reference evaluates determine_recompute_set on the sampler's current score_model.cache
 * (sampler.py:179-200), which a later enable_caching(**kwargs) replaces while the layers stay bound to
 * the first cache's tables (score_models.py:232). */
int ffd_cache_configure(ffd_ctx* ctx, const ffd_cache_cfg* cfg);
int ffd_cache_stats_get(const ffd_ctx* ctx, ffd_cache_stats* out);
/* Copy the (NL,H,L,hd) K and V tables (caching.py:88-91; zeros until step 0 ran) into
 * caller-owned device buffers; stream ordered.  Requires ffd_cache_enable. */
int ffd_cache_tables_read(ffd_ctx* ctx, float* k_out, float* v_out, void* stream);

/* One batch of DiffusionSampler.sample's inner loop (sampler.py:156-210): runs steps
 * [first_step, first_step + n_run) of an n_steps-step reverse diffusion on x (B,L,C) in
 * place, enqueuing every kernel on `stream` with no host synchronisation (one sync only
 * when the timestep grid differs from the previous call's).
 *   timesteps host array (n_steps) = noise_scheduler.timesteps (sde.py:62-64); the caller
 *             supplies it because torch.linspace's last-ulp rounding depends on the host's
 *             vector width; step_size = timesteps[0]-timesteps[1] in fp32.
 *   z_inject  NULL, or (n_run,B,L,C) injected N(0,1) draws for the steps of this call;
 *             NULL = Philox on device keyed (seed, step index) as in ffd_sde_step.
 *   use_cache 0/1 (requires ffd_cache_enable); `global_step0` is the reference's
 *             global_step (sampler.py:130,210) at `first_step`.
 * x must already hold the prior sample (ffd_prior) or the state after step first_step-1. */
int ffd_sample_batch(ffd_ctx* ctx, float* x, int B, const float* timesteps, int n_steps, float step_size,
                     int first_step, int n_run, uint64_t seed, uint64_t sample_offset, const float* z_inject,
                     int use_cache, int global_step0, void* stream);

/* Tuning knobs for experiments and for the test suite's kernel variants (results stay within the parity tolerance,
 * only the kernel choice / tiling changes).  PER CALLING THREAD since round 4 (thread_local): a knob set on one thread
 * selects kernels for the launches THAT thread makes and is invisible to every other thread, so two samplers on two
 * threads cannot disturb each other; a thread starts from the defaults.  ffd_tune_get reads the calling thread's value.
 *   "ffn_mb" = 0 (heuristic) | 1 | 2 | 3 | 4   rows/16 per workgroup of k_ffn_ln;
 *   "ffn_height" = 1 | 2 | 0                   where the 16-row tiles are 0.55 - 1 or 1.25 - 3 per CU (ECG: B = 12 ... 21 and
 *                                              28 ... 65, the reference's default sample_batch_size of 50 among them):
 *                                              k_ffn_ln at 16 / 32 / 48 rows per workgroup, one tile per CU, with the
 *                                              out-projection + LN1 inside (1) or behind a k_linear_res_ln launch (2) |
 *                                              0: the small-batch pair / sliced forms;
 *   "reset" (value ignored)                    every knob below back to its default;
 *   "ffn_rows" = 1 | 0 | 2                     FFN at large M (d_model 72 / 64 / 60 / 48): row-owning waves + CU-shared LDS weight ring
 *                                              (k_ffn_rows, ffd_ffn_rows.hip) or the F-split workgroup (k_ffn_ln); 2 = at
 *                                              every M where the small- / mid-batch forms are off (test suite);
 *   "rows_slices_fuse" = 0 (by estimate) | 1 | 2 sliced form of k_ffn_rows: the out-projection + LN1 inside every unit (1) or as
 *                                              one k_linear_res_ln launch in front of slices without that slot (2);
 *   "ffn_rows_nw" = 0 (heuristic) | 4 | 8 | 12 waves per workgroup of k_ffn_rows (a tile is 32 rows per wave);
 *   "ffn_rows_cps" = 0 | 2 | 1                 32-unit chunks per ring slot (= per barrier) of k_ffn_rows (default 2);
 *   "ffn_rows_fuse" = 1 | 0                    out-projection + residual + LN1 inside k_ffn_rows (one more ring slot per
 *                                              tile; two-chunk slots only) or k_linear_res_ln in front of it;
 *   "rows_slices" = 0 (heuristic) | -1 | 2..32 mid-size M: the fused kernel over tiles x slices of the hidden dimension
 *                                              (one unit per CU) + a reduce / LN2 launch; -1 = never, n = n slices forced;
 *   "ffn_persist" = 1 | 0 | n                  k_ffn_ln at large M: persistent grid (n x resident workgroups) or one workgroup per tile;
 *   "small_path" = 1 | 0                       small M: out-proj + LN1 + FFN + LN2 with F split over up to 16 workgroups per
 *                                              16-row tile and a reduce + LN2 launch (ffd_small.hip); "small_wgs" = n: most
 *                                              workgroups the split form is used for (0 = heuristic);
 *   "mid_path" = 1 | 0 | 2 | 4 | 8            mid-size M: the 64-row FFN main loop over F slices + the reduce launch
 *                                              (1 = where the round model says it pays, n = force n slices);
 *   "ffn_split" = 0 | 1                        OPT-IN, off by default: the FFN on the bf16 matrix cores, every fp32 operand cut
 *                                              into three bf16 parts and the six largest cross terms kept (fp32-equivalent to
 *                                              ~2e-7, passes the same goldens; NOT the reference's fp32 FMA arithmetic);
 *                                              needs d_model <= 96, dim_feedforward % 128 == 0 (ffd_ffn_split.hip);
 *   "embed_threads" = n                        threads the embedding kernel's grid aims at (default 262144); "embed_ldsx" = 1 | 0:
 *                                              a wave's shared x rows through LDS or per-lane loads;
 *   "attn_small" = 1 | 0 | 2 | 4               small batches: several workgroups per (sample, head), the key range of a
 *                                              q-tile cut into 2 / 4 pieces over the waves (1 = by batch size, 0 = never);
 *   "ffn_rem" = 1 | 0                          d%16 remainder rows of GEMM2 on the 4x4x1 MFMA;
 *   "lstm_wave" = 1 | 0                        LSTM: all layers as a wavefront of (16-sample tile, layer) workgroups
 *                                              (k_lstm_wave, every batch), or the per-layer kernels (0: the test
 *                                              suite's cross-check; 2 is accepted and means 1);
 *   "lstm_wave_fault" = n, "lstm_wave_spin_ms" = ms   tests: unit n - 1 of k_lstm_wave withholds its progress word / the
 *                                              time limit of one wait on such a word (see ffd_async_status);
 *   "fail_alloc_after" = n                     tests: the n-th device allocation from now fails with FFD_ERR_NOMEM;
 *   "lstm_wave_persist" = 1 | 0                one launch whose resident workgroups run (chunk, layer, tile) units in
 *                                              index order, or one launch per group of `per` layers;
 *                                              "lstm_wave_per" = n: at most n layers in flight (0 = as many as the CUs
 *                                              hold; tests); "lstm_wave_chunk" = 0 | 1 | even n: cell steps per
 *                                              unit where the (tile, layer) pairs outnumber the CUs (0: chosen so that
 *                                              the rounds come out whole, 1: layers walked whole, n: forced);
 *   "fuse_tail" = 1 | 0                        unembedding inside the SDE-step kernel of ffd_sample_batch (no FreSca);
 *   "attn_fused" = 1 | 0                       in-projection + attention in one kernel (k_qkv_attention*);
 *   "attn_kvq" = 1 | 0                         small-batch split attention: tile 0 = k | v, tile 1 = q, q projected for a
 *                                              workgroup's own q-tiles only (head_dim 6 / 8) | the whole head;
 *   "attn_hpw" = 0 (heuristic) | 1 | 2         heads per workgroup of that kernel;
 *   "attn_qg" = 0 (heuristic) | 1 | 2 | 3      query tiles per wave;
 */
int ffd_tune(const char* key, int value);
int ffd_tune_get(const char* key, int* value);

/* ---- introspection for benchmarks --------------------------------------- */

/* Algorithmic FLOPs of one score evaluation per sample (SURVEY 8(d) formula) and of
 * the fused FFN kernel per launch at batch B. */
double ffd_flops_per_sample_step(const ffd_ctx* ctx, int cache_hit);
double ffd_ffn_flops_per_launch(const ffd_ctx* ctx, int B);
/* Kernel classes of the sampling path, for in-situ timing and roofline accounting. */
enum {
  FFD_K_FFN = 0,         /* k_ffn_ln: linear1 + relu + linear2 + residual + LayerNorm2 */
  FFD_K_ATTN = 1,        /* k_qkv_attention*: in-projection + attention (or the two-kernel fallback) */
  FFD_K_OUTPROJ = 2,     /* k_linear_res_ln: out-projection + residual + LayerNorm1 */
  FFD_K_LSTM_REC = 3,    /* k_lstm_*: the L-step recurrence of one residual LSTM layer */
  FFD_K_LSTM_GATES = 4,  /* k_linear_rm: input-gate GEMM of one LSTM layer */
  FFD_K_SDE = 5,         /* k_sde_step (or the fused unembed + SDE step) */
  FFD_K_EMBED = 6,
  FFD_K_UNEMBED = 7,
  FFD_K_COUNT = 8
};
/* Diagnostics of the LSTM layer wavefront (k_lstm_wave): with host_out == NULL, arm a trace of `capacity_units` units --
 * the k_lstm_wave launches of the next forward passes (first sub-batch each) then write four 64-bit words per work unit
 * (chunk, layer, tile), unit index = chunk * layers * tiles + layer * tiles + tile: 100 MHz real-time ticks at the unit's
 * start, after its start-up (weights, state, first rows in), at its end, and [47:0] ticks spent waiting on progress words
 * | workgroup << 48.  With host_out != NULL: synchronise, copy the records out (n_units_out = how many), disarm.  The
 * capacity must cover chunks * layers * ceil(B / 16) units of the traced launch. */
int ffd_lstm_trace(ffd_ctx* ctx, unsigned long long* host_out, int capacity_units, int* n_units_out);

/* In-situ timing: between _begin and _end every launch of a kernel whose class bit is set in
 * `class_mask`, issued by forward / sample calls on this context, is bracketed by a HIP event pair
 * on the launch stream (up to max_launches pairs in total); _end synchronises; _get returns the mean
 * duration and launch count of one class.  For bench.py's roofline lines. */
int ffd_kernel_timing_begin(ffd_ctx* ctx, uint32_t class_mask, int max_launches);
int ffd_kernel_timing_end(ffd_ctx* ctx);
int ffd_kernel_timing_get(const ffd_ctx* ctx, int kernel_class, float* avg_ms_out, int* launches_out);
/* Algorithmic work of ONE launch of a kernel class at batch B (SURVEY 8(d) figures: FLOPs for the
 * MFMA-bound classes, HBM bytes for all); cache_hit = 1 for a pure-cache step.  Returns the kernel's
 * name (static string) or NULL for a class this model does not launch. */
const char* ffd_kernel_work(const ffd_ctx* ctx, int kernel_class, int B, int cache_hit, double* flops_out,
                            double* bytes_out);
/* Time `iters` launches of the dominant kernel (fused FFN+LN2 of layer 0) at batch B
 * on `stream` with HIP events; returns average milliseconds per launch in *ms_out.
 * Synchronous (benchmark helper only). */
int ffd_bench_ffn(ffd_ctx* ctx, int B, int iters, float* ms_out, void* stream);

/* Diagnostic: the shader clock the chip holds under the dominant kernel, and a timeline of one launch.  Launches the
 * fused FFN back to back for `warm_seconds` on random data, then once more with in-kernel stamps (s_memtime /
 * s_memrealtime, written to a scratch buffer nothing else reads); *ghz_out = median over workgroups of shader cycles
 * per 10 ns tick x 0.1 inside the main loop, *loop_us_out (may be NULL) = median time a workgroup spent in its main
 * loops.  raw_out (may be NULL): up to raw_capacity records of 8 x u64 per workgroup -- [0] main-loop shader cycles,
 * [1] main-loop 10 ns ticks, then chip-wide 100 MHz timestamps [2] entry, [3] / [4] first main loop begin / end,
 * [5] first epilogue end, [6] exit, and [7] tiles processed; *nwg_out = workgroups that ran.  Synchronous. */
int ffd_probe_ffn_clock(ffd_ctx* ctx, int B, double warm_seconds, double* ghz_out, double* loop_us_out,
                        unsigned long long* raw_out, int raw_capacity, int* nwg_out, void* stream);

/* Status of the stream-ordered work this context has enqueued, for the one failure a kernel can only report after
 * the fact: the LSTM backbone (LSTMScoreModule.forward, score_models.py:486-511) runs its layers as a wavefront of
 * workgroups that wait on each other's progress words inside ONE launch (k_lstm_wave).  That protocol needs the
 * launch's workgroups co-resident, i.e. THE DEVICE'S COMPUTE UNITS TO ITSELF: with another stream or process holding
 * compute units a wait can outlast its time limit (2 s per wait; ffd_tune "lstm_wave_spin_ms").  Such a wait is an
 * error, never a silent fall-through: the kernel records it, the launch drains, and the results of that launch are
 * invalid.  Returns FFD_ERR_STATE (message via ffd_last_error) once for such a launch, FFD_OK otherwise; call it after
 * synchronising the stream.  Every compute entry point makes the same check on entry, so the error also surfaces at
 * the next call.  (The reference has no counterpart: its nn.LSTM layers are separate stream-ordered kernels.) */
int ffd_async_status(ffd_ctx* ctx);

/* Diagnostic + benchmark helper for the fused in-projection + attention launch of layer 0 (ffd_qkvattn.hip) at batch B
 * on random rows (replaces nothing in the reference: measurement scaffolding for cached_transformer.py:228-311's
 * kernel).  n_recompute < 0: plain layer; otherwise the E2-CRF mode of that recompute-set size (needs ffd_cache_enable
 * and one FULL step for the tables).  Launches it back to back for `warm_seconds`, times `iters` launches with HIP
 * events (*ms_out = milliseconds per launch), then -- when raw_out != NULL -- once more as its stamped twin (d_model 72,
 * head_dim 6 only): up to raw_capacity records of 16 x u64 per WAVE: [0] 100 MHz real time at entry, shader-clock
 * (s_memtime) stamps [1] entry, [2] projection begin, [3] projection end, [4] attention begin, [5] attention end,
 * [6] exit, shader cycles summed over the wave's key tiles [7] K fragments + QK^T, [8] mask + softmax, [9] P.V,
 * [10] key tiles walked, [11] HW_ID, [12] 100 MHz real time at exit; *nrec_out = records written.  Synchronous. */
int ffd_probe_attn(ffd_ctx* ctx, int B, int n_recompute, double warm_seconds, int iters, float* ms_out,
                   unsigned long long* raw_out, int raw_capacity, int* nrec_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FFD_H */
