// libffd C ABI (include/ffd.h): context, weights, forward orchestration, E2-CRF cache
// state machine and the sampling loop.  Host code only launches kernels; there is no
// CPU compute path -- every entry point that needs the device fails loudly without one.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/ffd.h"
#include "ffd_internal.h"

using namespace ffd;

namespace {

struct DevBuf {
  float* p = nullptr;
  size_t n = 0;
};

struct LstmLayer {
  const float *wih, *whh, *bih, *bhh;
  float *wih_p, *bsum;
  float *ih_wpk = nullptr, *hh_wpk = nullptr, *b_wpk = nullptr;  // k_lstm_wave's fragment-ordered packs (d_model % 4 == 0, >= 16)
};

struct LayerPacked {
  float *in_wp, *q_wp, *kv_wp, *out_wp, *w1p, *w2p, *w2r;
  float* ring = nullptr;  // weight ring pack of the row-owning FFN
  float* ring_op = nullptr;  // + its out-projection slot (fused form)
  float *w1s = nullptr, *w2s = nullptr;  // bf16x3 packs of the opt-in split FFN, made on first use
  float *aw_full = nullptr, *aw_q = nullptr;    // per-head packs of the fused in-projection + attention kernel
  float *aw_full2 = nullptr, *aw_q2 = nullptr;  // same, per pair of heads (two-head workgroups)
  float* aw_kvq = nullptr;                      // per head, tile 0 = k | v, tile 1 = q (the split small-batch form)
};

__global__ void k_add_vec(const float* a, const float* b, float* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}

__global__ void k_fill_hash(float* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 16, h *= 0x85ebca6bu, h ^= h >> 13, h *= 0xc2b2ae35u, h ^= h >> 16;
    p[i] = ((float)(h >> 8) * (1.0f / 8388608.0f)) - 1.0f;  // uniform [-1, 1)
  }
}

}  // namespace

struct ffd_ctx {
  ffd_model_desc desc{};
  int device = 0;
  std::string err;
  std::map<std::string, DevBuf> raw;
  std::vector<void*> owned;  // packed / table / workspace allocations
  bool finalized = false;
  std::vector<LayerWeights> layers;
  std::vector<LayerPacked> packed;
  std::vector<LstmLayer> lstm;
  float* G_dev = nullptr;
  std::vector<float> G_host;
  // workspace
  int ws_B = 0;
  float *h0 = nullptr, *h1 = nullptr, *qkv = nullptr, *attn = nullptr, *score = nullptr;
  size_t qkv_floats = 0;
  float *temb1 = nullptr, *temb_tab = nullptr, *ts_dev = nullptr;
  int temb_cap = 0;
  std::vector<float> ts_host;
  long weight_epoch = 0, temb_epoch = -1;
  // in-situ kernel timing (ffd_kernel_timing_*)
  uint32_t time_mask = 0;
  std::vector<hipEvent_t> ev;  // pairs (start, stop)
  std::vector<int> ev_cls;     // kernel class of each used pair
  size_t ev_used = 0;
  float tm_ms[FFD_K_COUNT] = {0};
  int tm_n[FFD_K_COUNT] = {0};
  float* temb_b = nullptr;  // (B, d) per-sample time embeddings (ffd_score_forward_ts)
  int* lstm_prog = nullptr;  // progress words of the LSTM layer wavefront
  size_t lstm_prog_ints = 0;
  int* async_err = nullptr;  // host-mapped word a kernel's timed-out wait writes (k_lstm_wave); read by check_async
  float* lstm_state = nullptr;  // (tile, layer) state blocks of the time-chunked wavefront
  size_t lstm_state_floats = 0;
  float* ffn_part = nullptr;  // partial Y tiles of the small-M split FFN
  unsigned long long* lstm_trace = nullptr;  // ffd_lstm_trace: per-unit records of the next k_lstm_wave launch
  int lstm_trace_units = 0;
  size_t ffn_part_floats = 0;
  // FreSca (sampler-level)
  bool fresca_on = false;
  bool crf_cap_on = false;
  ffd_crf_capture_cfg crf_cap{};
  ffd_fresca_cfg fcfg{};
  float *score2 = nullptr, *fwork = nullptr;
  int fwork_B = 0;
  // cache
  bool cache_enabled = false;
  ffd_cache_cfg ccfg{5, 10};
  float *kt = nullptr, *vt = nullptr;
  bool table_allocated = false;
  ffd_cache_stats stats{};

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
  int d() const { return desc.d_model; }
  int hd() const { return desc.d_model / desc.n_head; }
  size_t table_floats() const { return (size_t)desc.num_layers * desc.n_head * desc.max_len * hd(); }
};

#define HIPCHECK(expr)                                                                          \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess)                                                                      \
      return ctx->fail(FFD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

static thread_local int g_fail_alloc_after = 0;  // tests: the n-th device allocation from now fails (ffd_tune "fail_alloc_after")

static int dev_alloc(ffd_ctx* ctx, float** p, size_t nfloats) {
  *p = nullptr;
  if (g_fail_alloc_after > 0 && --g_fail_alloc_after == 0)
    return ctx->fail(FFD_ERR_NOMEM, "hipMalloc(%zu floats) failed: injected by ffd_tune(\"fail_alloc_after\")", nfloats);
  hipError_t e = hipMalloc((void**)p, nfloats * sizeof(float) + 256);
  if (e != hipSuccess) return ctx->fail(FFD_ERR_NOMEM, "hipMalloc(%zu floats) failed: %s", nfloats, hipGetErrorString(e));
  ctx->owned.push_back(*p);
  return FFD_OK;
}

// Replace a workspace buffer by a larger one: the old allocation is released after the stream has drained (workspaces
// grow a handful of times; without this every growth kept the old buffer until ffd_destroy -- 3.6 GB for the q/k/v
// regions at B = 8192, L = 512).  `cap` is the capacity that guards the buffer: it reads 0 from the moment the old
// buffer is gone until the new one exists, so a failed growth (out of memory) leaves "nothing allocated", never a
// capacity that vouches for a freed or null pointer.  Capacities shared by several buffers (ws_B, fwork_B, temb_cap)
// are zeroed by the caller before the first regrow and set after the last.
template <typename CapT>
static int dev_regrow(ffd_ctx* ctx, float** p, size_t nfloats, CapT* cap, CapT cap_value) {
  float* old = *p;
  if (cap) *cap = 0;
  *p = nullptr;
  if (old) {
    (void)hipDeviceSynchronize();
    auto it = std::find(ctx->owned.begin(), ctx->owned.end(), (void*)old);
    if (it != ctx->owned.end()) ctx->owned.erase(it);
    (void)hipFree(old);
  }
  int rc = dev_alloc(ctx, p, nfloats);
  if (rc == FFD_OK && cap) *cap = cap_value;
  return rc;
}
static int dev_regrow(ffd_ctx* ctx, float** p, size_t nfloats) { return dev_regrow<int>(ctx, p, nfloats, nullptr, 0); }

static thread_local int g_fuse_tail = 1;
static thread_local int g_attn_kvq = 1;  // small-batch split attention on the kv | q pack (ffd_tune "attn_kvq")

// HIP event pair around a launch of kernel class `cls` while ffd_kernel_timing_begin has its bit set
struct Timed {
  ffd_ctx* ctx;
  hipStream_t s;
  bool on;
  Timed(ffd_ctx* c, int cls, hipStream_t st) : ctx(c), s(st) {
    on = ((c->time_mask >> cls) & 1u) && c->ev_used + 2 <= c->ev.size();
    if (on) {
      (void)hipEventRecord(c->ev[c->ev_used], s);
      c->ev_cls.push_back(cls);
    }
  }
  ~Timed() {
    if (on) {
      (void)hipEventRecord(ctx->ev[ctx->ev_used + 1], s);
      ctx->ev_used += 2;
    }
  }
};
#define TIMED(cls, expr)      \
  do {                        \
    Timed tm__(ctx, cls, s);  \
    HIPCHECK(expr);           \
  } while (0)

static bool d_supported(int d) {
#define X(v) if (d == v) return true;
  FFD_D_LIST(X)
#undef X
  return false;
}
static bool hd_supported(int hd) {
#define X(v) if (hd == v) return true;
  FFD_HD_LIST(X)
#undef X
  return false;
}

extern "C" {

int ffd_tune(const char* key, int value) {
  if (!key) return FFD_ERR_INVALID;
  if (!strcmp(key, "reset")) {  // every knob back to its default (the test suite calls this after each test)
    g_ffn_mb_override = 0, g_ffn_height = 1, g_ffn_persist = 1, g_ffn_rem = 1, g_ffn_split = 0, g_ffn_rows = 1, g_ffn_rows_nw = 0,
    g_ffn_rows_cps = 0, g_ffn_rows_fuse = 1, g_rows_slices = 0, g_rows_slices_fuse = 0, g_mid_path = 1, g_small_path = 1, g_small_wgs = 0, g_attn_small = 1, g_attn_fused = 1,
    g_attn_hpw = 0, g_attn_qg = 0, g_embed_ldsx = 1, g_embed_threads = 262144,
    g_lstm_wave = 1, g_lstm_wave_persist = 1, g_lstm_wave_per = 0, g_lstm_wave_chunk = 0, g_fuse_tail = 1, g_attn_kvq = 1, g_fail_alloc_after = 0, g_lstm_wave_fault = 0, g_lstm_wave_spin_ms = 2000;
    return FFD_OK;
  }
  if (!strcmp(key, "ffn_mb")) {
    if (value < 0 || value > 4) return FFD_ERR_INVALID;
    g_ffn_mb_override = value;
    return FFD_OK;
  }
  if (!strcmp(key, "ffn_height")) {  // 32- / 48-row k_ffn_ln tiles where the 16-row tiles are 1.4 - 3 per CU: 0 off | 1 one launch | 2 behind k_linear_res_ln
    if (value < 0 || value > 2) return FFD_ERR_INVALID;
    g_ffn_height = value;
    return FFD_OK;
  }
  if (!strcmp(key, "ffn_persist")) {  // 0: one workgroup per tile; n >= 1: persistent grid of n x the resident workgroups
    if (value < 0 || value > 8) return FFD_ERR_INVALID;
    g_ffn_persist = value;
    return FFD_OK;
  }
  if (!strcmp(key, "attn_small")) {  // small-batch attention: 0 never, 1 by batch size, 2 / 4 force the key pieces
    if (value != 0 && value != 1 && value != 2 && value != 4) return FFD_ERR_INVALID;
    g_attn_small = value;
    return FFD_OK;
  }
  if (!strcmp(key, "embed_ldsx")) {  // embedding kernel: the wave's x rows through LDS (1) or per-lane loads (0)
    g_embed_ldsx = value ? 1 : 0;
    return FFD_OK;
  }
  if (!strcmp(key, "embed_threads")) {
    if (value < 256) return FFD_ERR_INVALID;
    g_embed_threads = value;
    return FFD_OK;
  }
  if (!strcmp(key, "ffn_split")) {  // opt-in bf16x3-split FFN (not the reference's fp32 arithmetic; ffd_ffn_split.hip)
    g_ffn_split = value ? 1 : 0;
    return FFD_OK;
  }
  if (!strcmp(key, "mid_path")) {  // 64-row FFN over F slices for mid-size M: 0 off, 1 heuristic, 2 / 4 / 8 forced
    if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return FFD_ERR_INVALID;
    g_mid_path = value;
    return FFD_OK;
  }
  if (!strcmp(key, "small_wgs")) {  // most workgroups (row tiles x F splits) the small-M pair is used for
    if (value < 0) return FFD_ERR_INVALID;
    g_small_wgs = value;
    return FFD_OK;
  }
  if (!strcmp(key, "small_path")) {  // split out-proj + FFN pair for small M (0 = always the large-M kernels)
    g_small_path = value ? 1 : 0;
    return FFD_OK;
  }
  if (!strcmp(key, "ffn_rows")) {  // large-M FFN: row-owning waves + CU-shared weight ring (1, default) or k_ffn_ln (0)
    if (value < 0 || value > 2) return FFD_ERR_INVALID;
    g_ffn_rows = value;  // 2: at every M (the test suite runs the goldens through it)
    return FFD_OK;
  }
  if (!strcmp(key, "ffn_rows_cps")) {
    if (value < 0 || value > 2) return FFD_ERR_INVALID;
    g_ffn_rows_cps = value;
    return FFD_OK;
  }
  if (!strcmp(key, "ffn_rows_fuse")) {  // out-projection + LN1 inside k_ffn_rows (1, default) or k_linear_res_ln before it (0)
    g_ffn_rows_fuse = value ? 1 : 0;
    return FFD_OK;
  }
  if (!strcmp(key, "rows_slices")) {  // sliced form of the fused kernel at mid-size M: 0 heuristic, -1 off, 2..32 slices forced
    if (value < -1 || value == 1 || value > 32) return FFD_ERR_INVALID;
    g_rows_slices = value;
    return FFD_OK;
  }
  if (!strcmp(key, "rows_slices_fuse")) {  // sliced form: 0 by estimate | 1 out-projection inside every unit | 2 k_linear_res_ln once in front
    if (value < 0 || value > 2) return FFD_ERR_INVALID;
    g_rows_slices_fuse = value;
    return FFD_OK;
  }
  if (!strcmp(key, "ffn_rows_nw")) {
    if (value != 0 && value != 4 && value != 8 && value != 12) return FFD_ERR_INVALID;
    g_ffn_rows_nw = value;
    return FFD_OK;
  }
  if (!strcmp(key, "ffn_rem")) {
    g_ffn_rem = value ? 1 : 0;
    return FFD_OK;
  }
  if (!strcmp(key, "lstm_wave")) {  // LSTM layers as a wavefront (1, default; 2 = the same) | 0: the per-layer kernels
    if (value < 0 || value > 2) return FFD_ERR_INVALID;
    g_lstm_wave = value;
    return FFD_OK;
  }
  if (!strcmp(key, "lstm_wave_persist")) {  // k_lstm_wave workgroups walk their tile's layers (1) | one launch per layer group (0)
    g_lstm_wave_persist = value ? 1 : 0;
    return FFD_OK;
  }
  if (!strcmp(key, "lstm_wave_chunk")) {  // cell steps per unit of the time-shared wavefront: 0 = by the pass count, 1 = never, even n = forced
    if (value < 0 || (value > 1 && (value & 1)) || value > 1024) return FFD_ERR_INVALID;
    g_lstm_wave_chunk = value;
    return FFD_OK;
  }
  if (!strcmp(key, "lstm_wave_per")) {  // at most this many layers in flight (0 = as many as the CUs hold)
    if (value < 0 || value > 16) return FFD_ERR_INVALID;
    g_lstm_wave_per = value;
    return FFD_OK;
  }
  if (!strcmp(key, "lstm_wave_fault")) {  // tests: unit (value - 1) of k_lstm_wave never publishes its progress
    if (value < 0) return FFD_ERR_INVALID;
    g_lstm_wave_fault = value;
    return FFD_OK;
  }
  if (!strcmp(key, "lstm_wave_spin_ms")) {  // time limit of one wait on a progress word
    if (value < 1 || value > 20000) return FFD_ERR_INVALID;
    g_lstm_wave_spin_ms = value;
    return FFD_OK;
  }
  if (!strcmp(key, "fuse_tail")) {  // unembed inside the SDE-step kernel of ffd_sample_batch
    g_fuse_tail = value ? 1 : 0;
    return FFD_OK;
  }
  if (!strcmp(key, "attn_qg")) {
    if (value < 0 || value > 3) return FFD_ERR_INVALID;
    g_attn_qg = value;
    return FFD_OK;
  }
  if (!strcmp(key, "attn_hpw")) {
    if (value < 0 || value > 2) return FFD_ERR_INVALID;
    g_attn_hpw = value;
    return FFD_OK;
  }
  if (!strcmp(key, "fail_alloc_after")) {  // tests: the n-th device allocation from now fails with FFD_ERR_NOMEM (0 = off)
    if (value < 0) return FFD_ERR_INVALID;
    g_fail_alloc_after = value;
    return FFD_OK;
  }
  if (!strcmp(key, "attn_kvq")) {  // small-batch split attention: q projected for own q-tiles only (kv | q pack) | 0: whole head
    g_attn_kvq = value ? 1 : 0;
    return FFD_OK;
  }
  if (!strcmp(key, "attn_fused")) {
    if (value < 0 || value > 1) return FFD_ERR_INVALID;
    g_attn_fused = value;
    return FFD_OK;
  }
  return FFD_ERR_INVALID;
}

int ffd_tune_get(const char* key, int* value) {
  if (!key || !value) return FFD_ERR_INVALID;
  static const struct { const char* name; int* (*ptr)(); } tab[] = {
#define K(name, var) {name, []() -> int* { return &var; }}
      K("ffn_mb", g_ffn_mb_override), K("ffn_height", g_ffn_height), K("ffn_persist", g_ffn_persist), K("attn_small", g_attn_small),
      K("embed_ldsx", g_embed_ldsx), K("embed_threads", g_embed_threads), K("ffn_split", g_ffn_split),
      K("mid_path", g_mid_path), K("small_wgs", g_small_wgs), K("small_path", g_small_path), K("ffn_rows", g_ffn_rows),
      K("ffn_rows_cps", g_ffn_rows_cps), K("ffn_rows_fuse", g_ffn_rows_fuse), K("rows_slices", g_rows_slices), K("rows_slices_fuse", g_rows_slices_fuse),
      K("ffn_rows_nw", g_ffn_rows_nw), K("ffn_rem", g_ffn_rem), K("lstm_wave", g_lstm_wave),
      K("lstm_wave_persist", g_lstm_wave_persist), K("lstm_wave_chunk", g_lstm_wave_chunk),
      K("lstm_wave_per", g_lstm_wave_per), K("lstm_wave_fault", g_lstm_wave_fault),
      K("lstm_wave_spin_ms", g_lstm_wave_spin_ms), K("fuse_tail", g_fuse_tail), K("attn_qg", g_attn_qg),
      K("attn_hpw", g_attn_hpw), K("attn_kvq", g_attn_kvq), K("fail_alloc_after", g_fail_alloc_after), K("attn_fused", g_attn_fused),
#undef K
  };
  for (const auto& e : tab)
    if (!strcmp(key, e.name)) {
      *value = *e.ptr();  // (the calling thread's copy: the knobs are thread_local)
      return FFD_OK;
    }
  return FFD_ERR_INVALID;
}

const char* ffd_version(void) { return "libffd 0.1 (gfx950, fp32 MFMA 16x16x4, wave64)"; }

const char* ffd_last_error(const ffd_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int ffd_host_noise_scaling(int max_len, int fourier_noise_scaling, float* G) {
  if (max_len < 1 || !G) return FFD_ERR_INVALID;
  // sde.py:49-58 in fp32: ones * (1/sqrt2); G[0] *= sqrt2; (even L) G[L/2] *= sqrt2
  for (int i = 0; i < max_len; ++i) G[i] = 1.0f;
  if (fourier_noise_scaling) {
    const float c = (float)(1.0 / sqrt(2.0));
    const float s2 = (float)sqrt(2.0);
    for (int i = 0; i < max_len; ++i) G[i] = c * G[i];
    G[0] = G[0] * s2;
    if (max_len % 2 == 0) G[max_len / 2] = G[max_len / 2] * s2;
  }
  return FFD_OK;
}

int ffd_host_timesteps(int n, double eps, float* ts, float* step_size) {
  if (n < 2 || !ts) return FFD_ERR_INVALID;
  // torch.linspace(1.0, eps, n) fp32 (scalar form of ATen's linspace kernel):
  // step = (end - start)/(n-1); idx < n/2 ? start + step*idx : end - step*(n-idx-1)
  const float start = 1.0f, end = (float)eps;
  const float step = (end - start) / (float)(n - 1);
  const int half = n / 2;
  for (int i = 0; i < n; ++i) {
    volatile float prod = (i < half) ? step * (float)i : step * (float)(n - i - 1);
    ts[i] = (i < half) ? start + prod : end - prod;
  }
  if (step_size) *step_size = ts[0] - ts[1];
  return FFD_OK;
}

int ffd_host_gate(int step, int max_len, int K, int R) {
  // caching.py:131-181
  if (step == 0) return max_len;
  const int interval = (R < 100) ? 500 : R;
  const int k_tokens = K < max_len ? K : max_len;
  if (step % interval == 0) {
    int n = 2 * k_tokens;
    if (n > max_len) n = max_len;
    return n < 0 ? 0 : n;
  }
  return 0;
}

int ffd_create(ffd_ctx** out, const ffd_model_desc* desc, int device) {
  if (!out || !desc) return FFD_ERR_INVALID;
  *out = nullptr;
  ffd_ctx* ctx = new ffd_ctx();
  ctx->desc = *desc;
  ctx->device = device;
  *out = ctx;  // returned even on failure so the caller can read the message
  const ffd_model_desc& m = ctx->desc;
  if (m.n_channels < 1 || m.max_len < 1 || m.num_layers < 1)
    return ctx->fail(FFD_ERR_INVALID, "bad shape: C=%d L=%d NL=%d", m.n_channels, m.max_len, m.num_layers);
  if (m.kind == FFD_MODEL_MLP) {
    if (m.d_model < 1 || m.dim_feedforward < 1)
      return ctx->fail(FFD_ERR_INVALID, "mlp: d_model=%d d_mlp=%d", m.d_model, m.dim_feedforward);
  } else if (!d_supported(m.d_model)) {
    return ctx->fail(FFD_ERR_UNSUPPORTED, "d_model=%d: this build has kernels for d_model in {8, 16, 24, 32, 48, 60, 64, 72}", m.d_model);
  }
  if (m.kind == FFD_MODEL_TRANSFORMER) {
    if (m.n_head < 1 || m.d_model % m.n_head != 0)
      return ctx->fail(FFD_ERR_INVALID, "d_model=%d not divisible by n_head=%d", m.d_model, m.n_head);
    if (!hd_supported(m.d_model / m.n_head))
      return ctx->fail(FFD_ERR_UNSUPPORTED, "head_dim=%d: supported head dims are 2, 3, 4, 5, 6, 8", m.d_model / m.n_head);
    if (m.dim_feedforward < 64 || m.dim_feedforward % 64 != 0)
      return ctx->fail(FFD_ERR_UNSUPPORTED, "dim_feedforward=%d must be a positive multiple of 64", m.dim_feedforward);
    if (m.max_len > 512) return ctx->fail(FFD_ERR_UNSUPPORTED, "max_len=%d > 512 (attention kernel limit)", m.max_len);
  } else if (m.kind != FFD_MODEL_LSTM && m.kind != FFD_MODEL_MLP) {
    return ctx->fail(FFD_ERR_UNSUPPORTED, "model kind %d", m.kind);
  }
  if (m.sde != FFD_SDE_VP && m.sde != FFD_SDE_VE) return ctx->fail(FFD_ERR_UNSUPPORTED, "sde kind %d", m.sde);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return ctx->fail(FFD_ERR_HIP, "no HIP device available (%s): libffd has no CPU path",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev) return ctx->fail(FFD_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
  HIPCHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return ctx->fail(FFD_ERR_UNSUPPORTED, "device %d is %s; libffd is built for gfx950 only", device, prop.gcnArchName);
  ctx->G_host.resize(m.max_len);
  ffd_host_noise_scaling(m.max_len, m.fourier_noise_scaling, ctx->G_host.data());
  int rc = dev_alloc(ctx, &ctx->G_dev, m.max_len);
  if (rc) return rc;
  HIPCHECK(hipMemcpy(ctx->G_dev, ctx->G_host.data(), sizeof(float) * m.max_len, hipMemcpyHostToDevice));
  rc = dev_alloc(ctx, &ctx->temb1, m.d_model);
  if (rc) return rc;
  // (host memory the device writes through: readable by the host without a copy once the stream has drained)
  HIPCHECK(hipHostMalloc((void**)&ctx->async_err, 64, hipHostMallocMapped));
  memset(ctx->async_err, 0, 64);
  return FFD_OK;
}

void ffd_destroy(ffd_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (auto& kv : ctx->raw) (void)hipFree(kv.second.p);
  for (void* p : ctx->owned) (void)hipFree(p);
  for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
  if (ctx->async_err) (void)hipHostFree(ctx->async_err);
  delete ctx;
}

// ---------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------
static void expected_weights(const ffd_model_desc& m, std::vector<std::pair<std::string, size_t>>& out) {
  const size_t d = m.d_model, C = m.n_channels, L = m.max_len, F = m.dim_feedforward;
  if (m.kind == FFD_MODEL_TRANSFORMER) out.push_back({"pos_encoder.embedding.weight", L * d});
  out.push_back({"time_encoder.W", (d + 1) / 2});
  out.push_back({"time_encoder.dense.weight", d * d});
  out.push_back({"time_encoder.dense.bias", d});
  const size_t io = m.kind == FFD_MODEL_MLP ? L * C : C;  // the MLP embeds the flattened series (score_models.py:392-397)
  out.push_back({"embedder.weight", d * io});
  out.push_back({"embedder.bias", d});
  out.push_back({"unembedder.weight", io * d});
  out.push_back({"unembedder.bias", io});
  for (int i = 0; i < m.num_layers; ++i) {
    char p[64];
    if (m.kind == FFD_MODEL_TRANSFORMER) {
      snprintf(p, sizeof p, "backbone.layers.%d.", i);
      std::string s(p);
      out.push_back({s + "self_attn.in_proj_weight", 3 * d * d});
      out.push_back({s + "self_attn.in_proj_bias", 3 * d});
      out.push_back({s + "self_attn.out_proj.weight", d * d});
      out.push_back({s + "self_attn.out_proj.bias", d});
      out.push_back({s + "linear1.weight", F * d});
      out.push_back({s + "linear1.bias", F});
      out.push_back({s + "linear2.weight", d * F});
      out.push_back({s + "linear2.bias", d});
      out.push_back({s + "norm1.weight", d});
      out.push_back({s + "norm1.bias", d});
      out.push_back({s + "norm2.weight", d});
      out.push_back({s + "norm2.bias", d});
    } else if (m.kind == FFD_MODEL_MLP) {
      // torchvision.ops.MLP(d, [d_mlp, d]) = Sequential(Linear, ReLU, Dropout, Linear, Dropout): indices 0 and 3
      snprintf(p, sizeof p, "backbone.%d.", i);
      std::string s(p);
      out.push_back({s + "0.weight", F * d});
      out.push_back({s + "0.bias", F});
      out.push_back({s + "3.weight", d * F});
      out.push_back({s + "3.bias", d});
    } else {
      snprintf(p, sizeof p, "backbone.%d.", i);
      std::string s(p);
      out.push_back({s + "weight_ih_l0", 4 * d * d});
      out.push_back({s + "weight_hh_l0", 4 * d * d});
      out.push_back({s + "bias_ih_l0", 4 * d});
      out.push_back({s + "bias_hh_l0", 4 * d});
    }
  }
}

int ffd_load_weight(ffd_ctx* ctx, const char* name, const float* data, size_t n) {
  if (!ctx) return FFD_ERR_INVALID;
  if (!name || !data) return ctx->fail(FFD_ERR_INVALID, "ffd_load_weight: null argument");
  std::vector<std::pair<std::string, size_t>> exp;
  expected_weights(ctx->desc, exp);
  size_t want = 0;
  for (auto& kv : exp)
    if (kv.first == name) want = kv.second;
  if (!want) return ctx->fail(FFD_ERR_INVALID, "unexpected parameter '%s' for this model", name);
  if (want != n) return ctx->fail(FFD_ERR_INVALID, "parameter '%s': got %zu floats, expected %zu", name, n, want);
  HIPCHECK(hipSetDevice(ctx->device));
  DevBuf& b = ctx->raw[name];
  if (!b.p) {
    hipError_t e = hipMalloc((void**)&b.p, n * sizeof(float) + 256);
    if (e != hipSuccess) return ctx->fail(FFD_ERR_NOMEM, "hipMalloc for '%s' failed: %s", name, hipGetErrorString(e));
    b.n = n;
  }
  HIPCHECK(hipMemcpy(b.p, data, n * sizeof(float), hipMemcpyDefault));
  ctx->finalized = false;
  return FFD_OK;
}

int ffd_finalize_weights(ffd_ctx* ctx) {
  if (!ctx) return FFD_ERR_INVALID;
  const ffd_model_desc& m = ctx->desc;
  std::vector<std::pair<std::string, size_t>> exp;
  expected_weights(m, exp);
  for (auto& kv : exp)
    if (!ctx->raw.count(kv.first)) return ctx->fail(FFD_ERR_STATE, "missing parameter '%s'", kv.first.c_str());
  HIPCHECK(hipSetDevice(ctx->device));
  const int d = m.d_model, F = m.dim_feedforward;
  auto W = [&](const std::string& k) { return ctx->raw[k].p; };
  hipStream_t s = nullptr;
  if (m.kind == FFD_MODEL_TRANSFORMER) {
    HIPCHECK(launch_renorm_rows(W("pos_encoder.embedding.weight"), m.max_len, d, sqrtf((float)d), s));
    const bool first = ctx->packed.empty();
    if (first) ctx->packed.resize(m.num_layers);
    ctx->layers.resize(m.num_layers);
    for (int i = 0; i < m.num_layers; ++i) {
      char p[64];
      snprintf(p, sizeof p, "backbone.layers.%d.", i);
      std::string pre(p);
      LayerPacked& pk = ctx->packed[i];
      if (first) {
        int rc;
        if ((rc = dev_alloc(ctx, &pk.in_wp, dpack_floats(3 * d, d)))) return rc;
        if ((rc = dev_alloc(ctx, &pk.q_wp, dpack_floats(d, d)))) return rc;
        if ((rc = dev_alloc(ctx, &pk.kv_wp, dpack_floats(2 * d, d)))) return rc;
        if (qkv_attention_supported(d, d / m.n_head)) {
          if ((rc = dev_alloc(ctx, &pk.aw_full, attn_pack_floats(d, m.n_head, 1, 0)))) return rc;
          if ((rc = dev_alloc(ctx, &pk.aw_q, attn_pack_floats(d, m.n_head, 1, 1)))) return rc;
          if (attn_kvq_supported(d / m.n_head))
            if ((rc = dev_alloc(ctx, &pk.aw_kvq, attn_pack_floats(d, m.n_head, 1, 2)))) return rc;
          if (m.n_head % 2 == 0) {
            if ((rc = dev_alloc(ctx, &pk.aw_full2, attn_pack_floats(d, m.n_head, 2, 0)))) return rc;
            if ((rc = dev_alloc(ctx, &pk.aw_q2, attn_pack_floats(d, m.n_head, 2, 1)))) return rc;
          }
        }
        if ((rc = dev_alloc(ctx, &pk.out_wp, dpack_floats(d, d)))) return rc;
        if ((rc = dev_alloc(ctx, &pk.w1p, dpack_floats(F, d)))) return rc;
        if ((rc = dev_alloc(ctx, &pk.w2p, w2pack_floats(d, F)))) return rc;
        if ((rc = dev_alloc(ctx, &pk.w2r, w2rem_floats(d, F)))) return rc;
        if (ffn_rows_supported(d, F))
        {
          if ((rc = dev_alloc(ctx, &pk.ring, ffn_ring_floats(d, F)))) return rc;
          if ((rc = dev_alloc(ctx, &pk.ring_op, ffn_ring_oproj_floats(d)))) return rc;
        }
      }
      float* in_w = W(pre + "self_attn.in_proj_weight");
      HIPCHECK(launch_pack_dweight(in_w, pk.in_wp, 3 * d, d, s));
      HIPCHECK(launch_pack_dweight(in_w, pk.q_wp, d, d, s));
      HIPCHECK(launch_pack_dweight(in_w + (size_t)d * d, pk.kv_wp, 2 * d, d, s));
      if (pk.aw_full) {
        const float* in_b = W(pre + "self_attn.in_proj_bias");
        HIPCHECK(launch_pack_attn(in_w, in_b, pk.aw_full, d, m.n_head, 1, 0, s));
        HIPCHECK(launch_pack_attn(in_w, in_b, pk.aw_q, d, m.n_head, 1, 1, s));
        if (pk.aw_kvq) HIPCHECK(launch_pack_attn(in_w, in_b, pk.aw_kvq, d, m.n_head, 1, 2, s));
        if (pk.aw_full2) {
          HIPCHECK(launch_pack_attn(in_w, in_b, pk.aw_full2, d, m.n_head, 2, 0, s));
          HIPCHECK(launch_pack_attn(in_w, in_b, pk.aw_q2, d, m.n_head, 2, 1, s));
        }
      }
      HIPCHECK(launch_pack_dweight(W(pre + "self_attn.out_proj.weight"), pk.out_wp, d, d, s));
      HIPCHECK(launch_pack_dweight(W(pre + "linear1.weight"), pk.w1p, F, d, s));
      HIPCHECK(launch_pack_w2(W(pre + "linear2.weight"), pk.w2p, d, F, s));
      HIPCHECK(launch_pack_w2rem(W(pre + "linear2.weight"), pk.w2r, d, F, s));
      if (pk.ring) {
        HIPCHECK(launch_pack_ffn_ring(W(pre + "linear1.weight"), W(pre + "linear1.bias"), W(pre + "linear2.weight"), pk.ring, d, F, s));
        HIPCHECK(launch_pack_oproj_ring(W(pre + "self_attn.out_proj.weight"), pk.ring_op, d, s));
      }
      if (pk.w1s) HIPCHECK(launch_pack_ffn_split(W(pre + "linear1.weight"), W(pre + "linear2.weight"), pk.w1s, pk.w2s, d, F, s));
      LayerWeights& lw = ctx->layers[i];
      lw.in_w = in_w;
      lw.in_b = W(pre + "self_attn.in_proj_bias");
      lw.out_w = W(pre + "self_attn.out_proj.weight");
      lw.out_b = W(pre + "self_attn.out_proj.bias");
      lw.w1 = W(pre + "linear1.weight");
      lw.b1 = W(pre + "linear1.bias");
      lw.w2 = W(pre + "linear2.weight");
      lw.b2 = W(pre + "linear2.bias");
      lw.n1w = W(pre + "norm1.weight");
      lw.n1b = W(pre + "norm1.bias");
      lw.n2w = W(pre + "norm2.weight");
      lw.n2b = W(pre + "norm2.bias");
      lw.in_wp = pk.in_wp;
      lw.out_wp = pk.out_wp;
      lw.w1p = pk.w1p;
      lw.w2p = pk.w2p;
      lw.w2r = pk.w2r;
      lw.ring = pk.ring;
      lw.ring_op = pk.ring_op;
      lw.w1s = pk.w1s;
      lw.w2s = pk.w2s;
    }
  } else if (m.kind == FFD_MODEL_LSTM) {
    const bool first = ctx->lstm.empty();
    if (first) ctx->lstm.resize(m.num_layers);
    for (int i = 0; i < m.num_layers; ++i) {
      char p[64];
      snprintf(p, sizeof p, "backbone.%d.", i);
      std::string pre(p);
      LstmLayer& l = ctx->lstm[i];
      if (first) {
        int rc;
        if ((rc = dev_alloc(ctx, &l.wih_p, dpack_floats(4 * d, d)))) return rc;
        if ((rc = dev_alloc(ctx, &l.bsum, 4 * d))) return rc;
        if (d % 4 == 0 && d >= 16) {
          if ((rc = dev_alloc(ctx, &l.ih_wpk, lstm_wave_wpack_floats(d)))) return rc;
          if ((rc = dev_alloc(ctx, &l.hh_wpk, lstm_wave_wpack_floats(d)))) return rc;
          if ((rc = dev_alloc(ctx, &l.b_wpk, lstm_wave_bpack_floats(d)))) return rc;
        }
      }
      l.wih = W(pre + "weight_ih_l0");
      l.whh = W(pre + "weight_hh_l0");
      l.bih = W(pre + "bias_ih_l0");
      l.bhh = W(pre + "bias_hh_l0");
      HIPCHECK(launch_pack_dweight(l.wih, l.wih_p, 4 * d, d, s));
      hipLaunchKernelGGL(k_add_vec, dim3(cdiv(4 * d, 256)), dim3(256), 0, s, l.bih, l.bhh, l.bsum, 4 * d);
      HIPCHECK(hipGetLastError());
      if (l.ih_wpk) HIPCHECK(launch_pack_lstm_wave(l.wih, l.whh, l.bsum, l.ih_wpk, l.hh_wpk, l.b_wpk, d, s));
    }
  }
  HIPCHECK(hipStreamSynchronize(s));
  ctx->weight_epoch++;
  ctx->finalized = true;
  return FFD_OK;
}

// ---------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------
static int ensure_workspace(ffd_ctx* ctx, int B) {
  if (B <= ctx->ws_B) return FFD_OK;
  const ffd_model_desc& m = ctx->desc;
  const size_t M = (size_t)B * m.max_len, d = m.d_model;
  HIPCHECK(hipSetDevice(ctx->device));
  int rc;
  ctx->ws_B = 0;  // (nothing vouches for these buffers until all of them exist at the new size)
  if ((rc = dev_regrow(ctx, &ctx->h0, M * d))) return rc;
  if ((rc = dev_regrow(ctx, &ctx->score, M * m.n_channels))) return rc;
  if ((rc = dev_regrow(ctx, &ctx->temb_b, (size_t)B * d))) return rc;
  if (m.kind == FFD_MODEL_MLP) {
    if ((rc = dev_regrow(ctx, &ctx->h1, (size_t)B * d))) return rc;                  // ping-pong of the (B, d) state
    if ((rc = dev_regrow(ctx, &ctx->qkv, (size_t)B * m.dim_feedforward, &ctx->qkv_floats, (size_t)B * m.dim_feedforward))) return rc;  // hidden (B, d_mlp)
  } else if (m.kind == FFD_MODEL_TRANSFORMER) {
    if ((rc = dev_regrow(ctx, &ctx->h1, M * d))) return rc;
    if ((rc = dev_regrow(ctx, &ctx->attn, M * d))) return rc;
  }
  // (the head-major q/k/v regions of the two-kernel attention fallback and the LSTM gate pre-activations are
  //  allocated on first use by ensure_qkv: the default paths never touch them -- 3.6 GB at B = 8192, L = 512)
  ctx->ws_B = B;
  return FFD_OK;
}

static int ensure_qkv(ffd_ctx* ctx, size_t floats) {
  if (floats <= ctx->qkv_floats) return FFD_OK;
  return dev_regrow(ctx, &ctx->qkv, floats, &ctx->qkv_floats, floats);
}

// one score evaluation; temb points at d floats on the device (temb_stride = 0: shared by the batch) or at a
// (B, d) table (temb_stride = d: per-sample diffusion times).
// n_rec < 0: no cache.  Otherwise the E2-CRF mode for |recompute_tokens| = n_rec.
// hidden_out != nullptr (transformer / LSTM): skip the unembedding and return the final hidden state (M x d) instead;
// the sampling loop unembeds inside the SDE-step kernel.
static int forward_impl(ffd_ctx* ctx, const float* x, const float* temb, int temb_stride, float* score_out,
                        float* crf_out, int B, int n_rec, hipStream_t s, const float** hidden_out = nullptr) {
  const ffd_model_desc& m = ctx->desc;
  const int L = m.max_len, C = m.n_channels, d = m.d_model, M = B * L;
  if (m.kind == FFD_MODEL_MLP) {  // MLPScoreModule.forward, score_models.py:406-440
    const int io = L * C, F = m.dim_feedforward;
    // flatten "b t c -> b (t c)" is the memory layout already; time encoding is one (d,) vector per step
    // embedder(X) + time encoding: one (d,) vector per step (second bias), or a (B, d) table (added like a residual)
    HIPCHECK(launch_dense(x, ctx->raw["embedder.weight"].p, ctx->raw["embedder.bias"].p, temb_stride ? nullptr : temb,
                          temb_stride ? temb : nullptr, ctx->h0, B, d, io, 0, s));
    float* cur = ctx->h0;
    float* alt = ctx->h1;
    for (int i = 0; i < m.num_layers; ++i) {
      char p[64];
      snprintf(p, sizeof p, "backbone.%d.", i);
      const std::string pre(p);
      HIPCHECK(launch_dense(cur, ctx->raw[pre + "0.weight"].p, ctx->raw[pre + "0.bias"].p, nullptr, nullptr, ctx->qkv,
                            B, F, d, 1, s));
      HIPCHECK(launch_dense(ctx->qkv, ctx->raw[pre + "3.weight"].p, ctx->raw[pre + "3.bias"].p, nullptr, cur, alt, B,
                            d, F, 0, s));  // X + layer(X)
      std::swap(cur, alt);
    }
    HIPCHECK(launch_dense(cur, ctx->raw["unembedder.weight"].p, ctx->raw["unembedder.bias"].p, nullptr, nullptr,
                          score_out, B, io, d, 0, s));
    return FFD_OK;
  }
  if (m.kind == FFD_MODEL_LSTM) {
    TIMED(FFD_K_EMBED, launch_embed(x, ctx->raw["embedder.weight"].p, ctx->raw["embedder.bias"].p, nullptr, temb,
                                    temb_stride, ctx->h0, B, L, C, d, s));
    if (lstm_wave_selected(B, d) && m.num_layers <= 64 && lstm_wave_max_batch(L, d) >= 16) {  // mid-size batches: the layers as a wavefront (ffd_lstm.hip)
      const int Bw = B < lstm_wave_max_batch(L, d) ? B : lstm_wave_max_batch(L, d);  // samples per launch
      const size_t need = (size_t)16 + 16 * cdiv(Bw, 16);
      if (need > ctx->lstm_prog_ints) {
        float* pbuf = reinterpret_cast<float*>(ctx->lstm_prog);
        ctx->lstm_prog = nullptr;
        if (int rc = dev_regrow(ctx, &pbuf, need, &ctx->lstm_prog_ints, need)) return rc;
        ctx->lstm_prog = reinterpret_cast<int*>(pbuf);
      }
      const float *wih[64], *whh[64], *bs[64];
      for (int i = 0; i < m.num_layers; ++i) wih[i] = ctx->lstm[i].ih_wpk, whh[i] = ctx->lstm[i].hh_wpk, bs[i] = ctx->lstm[i].b_wpk;
      const size_t need_st = lstm_wave_state_floats(Bw, d, m.num_layers);
      if (need_st > ctx->lstm_state_floats) {
        if (int rc = dev_regrow(ctx, &ctx->lstm_state, need_st, &ctx->lstm_state_floats, need_st)) return rc;
      }
      for (int b0 = 0; b0 < B; b0 += Bw) {  // (samples are independent: sub-batches of a tile per CU, one after the other)
        const int nb = B - b0 < Bw ? B - b0 : Bw;
        TIMED(FFD_K_LSTM_REC, launch_lstm_wave(ctx->h0 + (size_t)b0 * L * d, wih, whh, bs, m.num_layers, nb, L, d,
                                               ctx->lstm_prog, ctx->lstm_state, ctx->async_err, s, b0 == 0 ? ctx->lstm_trace : nullptr));
      }
    } else
    for (int i = 0; i < m.num_layers; ++i) {
      const LstmLayer& l = ctx->lstm[i];
      if (int rc = ensure_qkv(ctx, (size_t)M * 4 * d)) return rc;  // gate pre-activations gx
      TIMED(FFD_K_LSTM_GATES, launch_linear(ctx->h0, l.wih_p, l.bsum, ctx->qkv, M, 4 * d, d, 4 * d, s));
      TIMED(FFD_K_LSTM_REC, launch_lstm_layer(ctx->h0, ctx->qkv, l.whh, B, L, d, s));
    }
    if (hidden_out) {
      *hidden_out = ctx->h0;
      return FFD_OK;
    }
    TIMED(FFD_K_UNEMBED, launch_unembed(ctx->h0, ctx->raw["unembedder.weight"].p, ctx->raw["unembedder.bias"].p,
                                        score_out, M, C, d, s));
    return FFD_OK;
  }
  const int H = m.n_head, hd = d / H, F = m.dim_feedforward;
  TIMED(FFD_K_EMBED, launch_embed(x, ctx->raw["embedder.weight"].p, ctx->raw["embedder.bias"].p,
                                  ctx->raw["pos_encoder.embedding.weight"].p, temb, temb_stride, ctx->h0, B, L, C, d, s));
  // mode selection, cached_transformer.py:139-220
  enum { STD, FULL, PURE, MIXED } mode = STD;
  if (n_rec >= 0) {
    if (n_rec == L) mode = FULL;
    else if ((double)n_rec > 0.8 * (double)L) mode = STD;
    else if (n_rec == 0) mode = PURE;
    else mode = MIXED;
  }
  const size_t lt = (size_t)H * L * hd;  // table floats per layer
  const int nreg = (mode == PURE) ? 1 : 3;
  const int n_own = (mode == PURE) ? 0 : (mode == MIXED) ? n_rec : L;
  const bool qkv_attn = g_attn_fused && ctx->packed[0].aw_full != nullptr;
  // q / k / v regions, head-major (B,H,L,hd): only the two-kernel fallback uses them
  if (!qkv_attn)
    if (int rc = ensure_qkv(ctx, (size_t)M * 3 * d)) return rc;
  float* qreg = ctx->qkv;
  float* kreg = ctx->qkv + (size_t)M * d;
  float* vreg = ctx->qkv + 2 * (size_t)M * d;
  float* cur = ctx->h0;  // layer input / residual
  float* alt = ctx->h1;
  auto proj_w = [&](int i) { return mode == PURE ? ctx->packed[i].q_wp : ctx->packed[i].in_wp; };
  // opt-in: the FFN on the bf16 matrix cores as a three-part split (ffd_ffn_split.hip); takes every batch size, so
  // that all parity cases exercise it when it is on
  if (g_ffn_split && !ffn_split_supported(d, F))
    return ctx->fail(FFD_ERR_UNSUPPORTED, "ffn_split needs d_model %% 4 == 0, d_model <= 96, dim_feedforward %% 128 == 0");
  const bool split_ffn = g_ffn_split != 0;
  for (int i = 0; i < m.num_layers; ++i) {
    const LayerWeights& w = ctx->layers[i];
    const LayerPacked& pk = ctx->packed[i];
    float* kt = ctx->kt ? ctx->kt + i * lt : nullptr;
    float* vt = ctx->vt ? ctx->vt + i * lt : nullptr;
    const bool tables = (mode == PURE || mode == MIXED);
    if (qkv_attn) {
      // in-projection + attention in one launch: q/k/v never leave the CU (ffd_qkvattn.hip); in MIXED batch
      // element 0's workgroups also publish their recomputed K/V rows (caching.py:326-328)
      const int hpw = pk.aw_full2 ? qkv_attention_hpw(d, hd, L, B) : 1;
      // (small batches, not a pure cache hit: the split form on the kv | q pack -- q projected for own q-tiles only)
      const bool kvq = g_attn_kvq && pk.aw_kvq != nullptr && mode != PURE && qkv_attention_small_split(B, H, L) != 0;
      const float* pack = kvq ? pk.aw_kvq
                              : hpw == 2 ? (mode == PURE ? pk.aw_q2 : pk.aw_full2) : (mode == PURE ? pk.aw_q : pk.aw_full);
      TIMED(FFD_K_ATTN, launch_qkv_attention(cur, pack, hpw, kvq ? 2 : mode == PURE, tables ? kt : nullptr,
                                             tables ? vt : nullptr, mode == MIXED ? kt : nullptr,
                                             mode == MIXED ? vt : nullptr, ctx->attn, B, L, d, hd, n_own, s));
    } else {
      HIPCHECK(launch_linear_hm(cur, proj_w(i), w.in_b, qreg, kreg, vreg, M, nreg, d, L, H, hd, s));
      HIPCHECK(launch_attention(qreg, kreg, vreg, tables ? kt : nullptr, tables ? vt : nullptr, ctx->attn, B, L, H, hd,
                                n_own, s));
      if (mode == MIXED)  // store batch element 0's recomputed rows (caching.py:326-328, cached_transformer.py:301-305)
        HIPCHECK(launch_kv_store(kreg, vreg, kt, vt, L, H, hd, n_rec, s));
    }
    if (split_ffn && pk.w1s == nullptr) {  // first use: make the packs
      LayerPacked& pkm = ctx->packed[i];
      if (int rc = dev_alloc(ctx, &pkm.w1s, w1split_bytes(d, F) / sizeof(float))) return rc;
      if (int rc = dev_alloc(ctx, &pkm.w2s, w2split_bytes(d, F) / sizeof(float))) return rc;
      HIPCHECK(launch_pack_ffn_split(w.w1, w.w2, pkm.w1s, pkm.w2s, d, F, s));
      ctx->layers[i].w1s = pkm.w1s;
      ctx->layers[i].w2s = pkm.w2s;
    }
    if (!split_ffn && g_ffn_height == 1 && ffn_height_plan(M, d, F)) {
      // 1.4 - 3 16-row tiles per CU: one 32- / 48-row tile per CU, out-proj + LN1 + FFN + LN2 in one launch, in place
      TIMED(FFD_K_FFN, launch_oproj_ffn_ln(ctx->attn, cur, w, cur, M, d, F, s));
    } else if (const int ns = split_ffn ? 0 : small_path_splits(M, d, F)) {
      // small M: out-proj + LN1 recomputed per F split, FFN partials + a deterministic reduce / LN2 launch
      const size_t need = small_path_partial_floats(M, d, ns);
      if (need > ctx->ffn_part_floats) {
        if (int rc = dev_regrow(ctx, &ctx->ffn_part, need, &ctx->ffn_part_floats, need)) return rc;
      }
      TIMED(FFD_K_FFN, launch_oproj_ffn_small(ctx->attn, cur, w, alt, ctx->ffn_part, cur, M, d, F, ns, s));
    } else if (int snw = 0, sns = 0, sunf = 0; !split_ffn && w.ring_op != nullptr && rows_slice_plan(M, d, F, &snw, &sns, &sunf)) {
      // mid-size M: the row-owning kernel over tiles x slices of the hidden dimension and the reduce / LN2 launch
      // (deterministic: the slices are added in order) -- with the out-projection + LN1 inside every unit, or (where
      // the units are short) as one k_linear_res_ln launch in front
      const size_t need = rows_slice_floats(M, d, sns);
      if (need > ctx->ffn_part_floats) {
        if (int rc = dev_regrow(ctx, &ctx->ffn_part, need, &ctx->ffn_part_floats, need)) return rc;
      }
      if (sunf) {
        TIMED(FFD_K_OUTPROJ, launch_linear_res_ln(ctx->attn, pk.out_wp, w.out_b, cur, w.n1w, w.n1b, alt, M, d, s));
        TIMED(FFD_K_FFN, launch_ffn_rows_sliced(alt, w, ctx->ffn_part, cur, M, d, F, snw, sns, s));
      } else {
        TIMED(FFD_K_FFN, launch_oproj_ffn_rows_sliced(ctx->attn, cur, w, ctx->ffn_part, alt, M, d, F, snw, sns, s));
        float* t = cur;
        cur = alt, alt = t;
      }
    } else if (!split_ffn && !mid_path_splits(M, d, F) && w.ring_op != nullptr && ffn_rows_fused_selected(M, d, F)) {
      // large M, d_model 72: out-proj + LN1 + FFN + LN2 in one launch (x1 never leaves the CU); the output goes to the
      // other hidden buffer (rows are read and written by different waves of different tiles: no in-place form)
      TIMED(FFD_K_FFN, launch_oproj_ffn_rows(ctx->attn, cur, w, alt, M, d, F, s));
      float* t = cur;
      cur = alt, alt = t;
    } else {
      TIMED(FFD_K_OUTPROJ, launch_linear_res_ln(ctx->attn, pk.out_wp, w.out_b, cur, w.n1w, w.n1b, alt, M, d, s));
      const int nm = split_ffn ? 0 : mid_path_splits(M, d, F);
      if (nm) {  // mid-size M: 64-row tiles x F slices, partial tiles + the reduce / LN2 launch
        const size_t need = small_path_partial_floats(cdiv(M, 64) * 64, d, nm);
        if (need > ctx->ffn_part_floats) {
          if (int rc = dev_regrow(ctx, &ctx->ffn_part, need, &ctx->ffn_part_floats, need)) return rc;
        }
        TIMED(FFD_K_FFN, launch_ffn_mid(alt, w, ctx->ffn_part, cur, M, d, F, nm, s));
      } else if (split_ffn) TIMED(FFD_K_FFN, launch_ffn_ln_split(alt, w, cur, M, d, F, s));
      else TIMED(FFD_K_FFN, launch_ffn_ln(alt, w, cur, M, d, F, s));
    }
    if (mode == FULL) {
      // K,V of the layer OUTPUT for batch element 0 (cached_transformer.py:144-158), written
      // straight into this layer's tables: head-major (1,H,L,hd) == table layout
      HIPCHECK(launch_linear_hm(cur, pk.kv_wp, w.in_b + d, kt, vt, nullptr, L, 2, d, L, H, hd, s));
    }
    if (n_rec >= 0 && crf_out)  // crf[l] = h_l[0]  (score_models.py:181-194)
      HIPCHECK(hipMemcpyAsync(crf_out + (size_t)i * L * d, cur, sizeof(float) * L * d, hipMemcpyDeviceToDevice, s));
  }
  if (n_rec >= 0) {  // counters, caching.py:283,299,396
    if (mode == FULL) ctx->stats.recompute_count += (int64_t)L * m.num_layers, ctx->table_allocated = true;
    else if (mode == PURE) ctx->stats.cache_hit_count += (int64_t)L * m.num_layers;
    else if (mode == MIXED) {
      ctx->stats.cache_hit_count += (int64_t)(L - n_rec) * m.num_layers;
      ctx->stats.recompute_count += (int64_t)n_rec * m.num_layers;
      ctx->table_allocated = true;
    }
  }
  if (hidden_out) {
    *hidden_out = cur;
    return FFD_OK;
  }
  TIMED(FFD_K_UNEMBED, launch_unembed(cur, ctx->raw["unembedder.weight"].p, ctx->raw["unembedder.bias"].p, score_out,
                                      M, C, d, s));
  return FFD_OK;
}

// A kernel whose bounded wait ran out (k_lstm_wave's progress-word protocol) has left a code in ctx->async_err: the
// results of that launch are void.  Reported once, by whichever entry point comes next (or ffd_async_status).
static int check_async(ffd_ctx* ctx) {
  if (!ctx->async_err) return FFD_OK;
  const int code = *reinterpret_cast<volatile int*>(ctx->async_err);
  if (code == 0) return FFD_OK;
  *reinterpret_cast<volatile int*>(ctx->async_err) = 0;
  return ctx->fail(FFD_ERR_STATE,
                   "k_lstm_wave: unit %d waited longer than %d ms for the unit it depends on (the layer wavefront needs the "
                   "device's compute units to itself: another stream or process was holding some); the results of that "
                   "launch are invalid", code - 1, g_lstm_wave_spin_ms);
}

static int check_ready(ffd_ctx* ctx, int B) {
  if (int rc = check_async(ctx)) return rc;
  if (!ctx->finalized) return ctx->fail(FFD_ERR_STATE, "weights not finalised (call ffd_finalize_weights)");
  if (B < 1) return ctx->fail(FFD_ERR_INVALID, "batch size %d", B);
  if ((double)B * ctx->desc.max_len * 4 * ctx->desc.d_model > 2.0e9)
    return ctx->fail(FFD_ERR_UNSUPPORTED, "batch %d too large for 32-bit row indexing; shard the batch", B);
  return FFD_OK;
}

static int temb_single(ffd_ctx* ctx, float t, hipStream_t s) {
  HIPCHECK(launch_time_embed(nullptr, t, 1, ctx->raw["time_encoder.W"].p, ctx->raw["time_encoder.dense.weight"].p,
                             ctx->raw["time_encoder.dense.bias"].p, ctx->temb1, ctx->desc.d_model, s));
  return FFD_OK;
}

int ffd_score_forward(ffd_ctx* ctx, const float* x, float t, float* score_out, int B, void* stream) {
  if (!ctx) return FFD_ERR_INVALID;
  int rc = check_ready(ctx, B);
  if (rc) return rc;
  if (!x || !score_out) return ctx->fail(FFD_ERR_INVALID, "null buffer");
  HIPCHECK(hipSetDevice(ctx->device));
  if ((rc = ensure_workspace(ctx, B))) return rc;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = temb_single(ctx, t, s))) return rc;
  return forward_impl(ctx, x, ctx->temb1, 0, score_out, nullptr, B, -1, s);
}

int ffd_score_forward_cached(ffd_ctx* ctx, const float* x, float t, float* score_out, float* crf_out, int B,
                             int n_recompute, void* stream) {
  if (!ctx) return FFD_ERR_INVALID;
  int rc = check_ready(ctx, B);
  if (rc) return rc;
  if (ctx->desc.kind != FFD_MODEL_TRANSFORMER)
    return ctx->fail(FFD_ERR_UNSUPPORTED, "caching is only defined for the transformer backbone (SURVEY Q9)");
  if (!ctx->cache_enabled) return ctx->fail(FFD_ERR_STATE, "cache not enabled (call ffd_cache_enable)");
  if (!x || !score_out) return ctx->fail(FFD_ERR_INVALID, "null buffer");
  if (n_recompute < 0 || n_recompute > ctx->desc.max_len)
    return ctx->fail(FFD_ERR_INVALID, "n_recompute=%d outside [0,%d]", n_recompute, ctx->desc.max_len);
  HIPCHECK(hipSetDevice(ctx->device));
  if ((rc = ensure_workspace(ctx, B))) return rc;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = temb_single(ctx, t, s))) return rc;
  return forward_impl(ctx, x, ctx->temb1, 0, score_out, crf_out, B, n_recompute, s);
}

int ffd_score_forward_ts(ffd_ctx* ctx, const float* x, const float* timesteps, float* score_out, float* crf_out,
                         int B, int n_recompute, void* stream) {
  if (!ctx) return FFD_ERR_INVALID;
  int rc = check_ready(ctx, B);
  if (rc) return rc;
  if (!x || !timesteps || !score_out) return ctx->fail(FFD_ERR_INVALID, "null buffer");
  if (n_recompute >= 0) {
    if (ctx->desc.kind != FFD_MODEL_TRANSFORMER)
      return ctx->fail(FFD_ERR_UNSUPPORTED, "caching is only defined for the transformer backbone (SURVEY Q9)");
    if (!ctx->cache_enabled) return ctx->fail(FFD_ERR_STATE, "cache not enabled (call ffd_cache_enable)");
    if (n_recompute > ctx->desc.max_len)
      return ctx->fail(FFD_ERR_INVALID, "n_recompute=%d outside [0,%d]", n_recompute, ctx->desc.max_len);
  }
  HIPCHECK(hipSetDevice(ctx->device));
  if ((rc = ensure_workspace(ctx, B))) return rc;
  hipStream_t s = (hipStream_t)stream;
  // one time embedding per sample: dense(gamma(t_b)) (transformer.py:77-91)
  HIPCHECK(launch_time_embed(timesteps, 0.f, B, ctx->raw["time_encoder.W"].p, ctx->raw["time_encoder.dense.weight"].p,
                             ctx->raw["time_encoder.dense.bias"].p, ctx->temb_b, ctx->desc.d_model, s));
  return forward_impl(ctx, x, ctx->temb_b, ctx->desc.d_model, score_out, n_recompute >= 0 ? crf_out : nullptr, B,
                      n_recompute >= 0 ? n_recompute : -1, s);
}

// ---------------------------------------------------------------------------
// SDE
// ---------------------------------------------------------------------------
static SdeParams sde_params(int sde, double a, double b, double t, float step_size) {
  SdeParams p{};
  p.sde = sde;
  if (sde == FFD_SDE_VP) {
    const double beta = a + t * (b - a);  // sde.py:212-213
    p.a = (float)(-0.5 * beta);
    p.cs = (float)sqrt(beta);
  } else {
    const double r = b / a;
    p.cs = (float)(a * sqrt(2.0 * log(r)) * pow(r, t));  // sde.py:143-147
    p.a = 0.f;
  }
  p.dt = step_size;
  p.sqdt = sqrtf(step_size);
  return p;
}

int ffd_sde_step(const ffd_sde_desc* sde, float* x, const float* score, const float* G, double t, float step_size,
                 const float* z, uint64_t seed, uint64_t sample_offset, int step, int B, int L, int C, void* stream) {
  if (!sde || !x || !score || !G || B < 1 || L < 1 || C < 1) return FFD_ERR_INVALID;
  if (sde->sde != FFD_SDE_VP && sde->sde != FFD_SDE_VE) return FFD_ERR_UNSUPPORTED;
  if (!(step_size > 0.f)) return FFD_ERR_INVALID;  // sde.py:157,238 assert
  hipError_t e = launch_sde_step(x, score, z, G, sde_params(sde->sde, sde->a, sde->b, t, step_size), seed,
                                 sample_offset * (uint64_t)L * C, (uint32_t)step, B, L, C, (hipStream_t)stream);
  return e == hipSuccess ? FFD_OK : FFD_ERR_HIP;
}

int ffd_prior(const ffd_sde_desc* sde, float* x, const float* z, const float* G, uint64_t seed, uint64_t sample_offset,
              int B, int L, int C, void* stream) {
  if (!sde || !x || !G || B < 1 || L < 1 || C < 1) return FFD_ERR_INVALID;
  const float scale = (sde->sde == FFD_SDE_VE) ? (float)sde->b : 1.0f;
  hipError_t e = launch_prior(x, z, G, scale, seed, sample_offset * (uint64_t)L * C, B, L, C, (hipStream_t)stream);
  return e == hipSuccess ? FFD_OK : FFD_ERR_HIP;
}

int ffd_dft(const float* in, float* out, int B, int L, int C, void* stream) {
  if (!in || !out || in == out || B < 0 || L < 1 || C < 1) return FFD_ERR_INVALID;
  hipError_t e = launch_dft(in, out, B, L, C, 0, nullptr, nullptr, (hipStream_t)stream);
  return e == hipSuccess ? FFD_OK : (e == hipErrorInvalidValue ? FFD_ERR_UNSUPPORTED : FFD_ERR_HIP);
}

int ffd_idft(const float* in, float* out, int B, int L, int C, void* stream) {
  if (!in || !out || in == out || B < 0 || L < 1 || C < 1) return FFD_ERR_INVALID;
  hipError_t e = launch_dft(in, out, B, L, C, 1, nullptr, nullptr, (hipStream_t)stream);
  return e == hipSuccess ? FFD_OK : (e == hipErrorInvalidValue ? FFD_ERR_UNSUPPORTED : FFD_ERR_HIP);
}

int ffd_dft_standardize(const float* in, float* out, const float* mean, const float* std, int B, int L, int C,
                        void* stream) {
  if (!in || !out || in == out || !mean || !std || B < 0 || L < 1 || C < 1) return FFD_ERR_INVALID;
  hipError_t e = launch_dft(in, out, B, L, C, 0, mean, std, (hipStream_t)stream);
  return e == hipSuccess ? FFD_OK : (e == hipErrorInvalidValue ? FFD_ERR_UNSUPPORTED : FFD_ERR_HIP);
}

int ffd_unstandardize_idft(const float* in, float* out, const float* mean, const float* std, int B, int L, int C,
                           void* stream) {
  if (!in || !out || in == out || !mean || !std || B < 0 || L < 1 || C < 1) return FFD_ERR_INVALID;
  hipError_t e = launch_dft(in, out, B, L, C, 1, std, mean, (hipStream_t)stream);
  return e == hipSuccess ? FFD_OK : (e == hipErrorInvalidValue ? FFD_ERR_UNSUPPORTED : FFD_ERR_HIP);
}

int ffd_positional_encoding(const float* x, float* weight, float* out, int B, int L, int D, float max_norm,
                            void* stream) {
  if (!x || !weight || !out || B < 1 || L < 1 || D < 1) return FFD_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  // nn.Embedding(max_norm) renormalises the looked-up rows in place at every forward (transformer.py:13-15,26)
  if (max_norm > 0.f && launch_renorm_rows_once(weight, L, D, max_norm, s) != hipSuccess) return FFD_ERR_HIP;
  return launch_add_table(x, weight, nullptr, out, B, L, D, s) == hipSuccess ? FFD_OK : FFD_ERR_HIP;
}

int ffd_time_encoding(const float* x, const float* timesteps, const float* W, const float* dense_w,
                      const float* dense_b, float* temb_work, float* out, int B, int L, int D, void* stream) {
  if (!x || !timesteps || !W || !dense_w || !dense_b || !temb_work || !out || B < 1 || L < 1 || D < 1)
    return FFD_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (launch_time_embed(timesteps, 0.f, B, W, dense_w, dense_b, temb_work, D, s) != hipSuccess) return FFD_ERR_HIP;
  return launch_add_table(x, nullptr, temb_work, out, B, L, D, s) == hipSuccess ? FFD_OK : FFD_ERR_HIP;
}

int ffd_fresca(const float* in, float* out, float* work, int B, int L, int C, float low_scale, float high_scale,
               double cutoff_ratio, int strategy, void* stream) {
  if (!in || !out || in == out || B < 1 || L < 2 || C < 1) return FFD_ERR_INVALID;
  if (strategy != FFD_FRESCA_SPATIAL && strategy != FFD_FRESCA_ENERGY) return FFD_ERR_INVALID;  // fresca.py:60 ValueError
  if (strategy == FFD_FRESCA_ENERGY && !work) return FFD_ERR_INVALID;
  hipError_t e = launch_fresca(in, out, work, B, L, C, low_scale, high_scale, cutoff_ratio, strategy, (hipStream_t)stream);
  return e == hipSuccess ? FFD_OK : (e == hipErrorInvalidValue ? FFD_ERR_UNSUPPORTED : FFD_ERR_HIP);
}

int ffd_fresca2d(const float* in, float* out, float* work, int B, int H, int W, int C, float low_scale,
                 float high_scale, double cutoff_ratio, int strategy, void* stream) {
  if (!in || !out || !work || in == out || B < 1 || H < 1 || W < 1 || C < 1) return FFD_ERR_INVALID;
  if (strategy != FFD_FRESCA_SPATIAL && strategy != FFD_FRESCA_ENERGY) return FFD_ERR_INVALID;  // fresca.py:103 ValueError
  if (!fresca2d_supported(H, W)) return FFD_ERR_UNSUPPORTED;
  hipError_t e = launch_fresca2d(in, out, work, B, H, W, C, low_scale, high_scale, cutoff_ratio, strategy, (hipStream_t)stream);
  return e == hipSuccess ? FFD_OK : (e == hipErrorInvalidValue ? FFD_ERR_UNSUPPORTED : FFD_ERR_HIP);
}

int ffd_fresca_enable(ffd_ctx* ctx, const ffd_fresca_cfg* cfg) {
  if (!ctx) return FFD_ERR_INVALID;
  if (!cfg) return ctx->fail(FFD_ERR_INVALID, "null fresca config");
  if (cfg->strategy != FFD_FRESCA_SPATIAL && cfg->strategy != FFD_FRESCA_ENERGY)
    return ctx->fail(FFD_ERR_INVALID, "unknown cutoff strategy %d", cfg->strategy);
  ctx->fcfg = *cfg;
  ctx->fresca_on = true;
  return FFD_OK;
}

int ffd_fresca_disable(ffd_ctx* ctx) {
  if (!ctx) return FFD_ERR_INVALID;
  ctx->fresca_on = false;
  return FFD_OK;
}

int ffd_freq_decompose(const float* x, float* low, float* high, int B, int L, int D, double low_freq_ratio,
                       void* stream) {
  if (!x || !low || !high || x == low || x == high || low == high || B < 1 || L < 2 || D < 1) return FFD_ERR_INVALID;
  hipError_t e = launch_freq_decompose(x, low, high, B, L, D, low_freq_ratio, (hipStream_t)stream);
  return e == hipSuccess ? FFD_OK : (e == hipErrorInvalidValue ? FFD_ERR_UNSUPPORTED : FFD_ERR_HIP);
}

int ffd_spectral_density(const float* xf, float* out, int B, int L, int C, void* stream) {
  if (!xf || !out || xf == out || B < 1 || L < 1 || C < 1) return FFD_ERR_INVALID;
  return launch_spectral_density(xf, out, B, L, C, (hipStream_t)stream) == hipSuccess ? FFD_OK : FFD_ERR_HIP;
}

// Hermite polynomials H_0..H_order at s (fourier.py:341-394: physicists' recurrence)
static void hermite_row(double s, int order, double* H) {
  H[0] = 1.0;
  if (order >= 1) H[1] = 2.0 * s;
  for (int n = 1; n < order; ++n) H[n + 1] = 2.0 * s * H[n] - 2.0 * n * H[n - 1];
}

int ffd_hermite_predict(const float* history, const double* timesteps, double target, int order, float* out, int K,
                        size_t n, void* stream) {
  if (!history || !timesteps || !out || K < 1 || K > 32 || order < 0 || order > 8) return FFD_ERR_INVALID;
  float w[32] = {0};
  double tmin = timesteps[0], tmax = timesteps[0];
  for (int k = 1; k < K; ++k) tmin = std::min(tmin, timesteps[k]), tmax = std::max(tmax, timesteps[k]);
  if (K < 2 || tmax == tmin) {
    w[K - 1] = 1.f;  // fourier.py:416-428: not enough history -> last value
  } else {
    const int P = order + 1;
    auto norm = [&](double t) {  // fourier.py:431-441, values held in fp32 tensors and clamped to [-1,1]
      double v = (double)(float)(2.0 * (t - tmin) / (tmax - tmin) - 1.0);
      return std::min(1.0, std::max(-1.0, v));
    };
    double Hm[32][9], Ht[9], A[9][18];
    for (int k = 0; k < K; ++k) hermite_row(norm(timesteps[k]), order, Hm[k]);
    hermite_row(norm(target), order, Ht);
    // normal equations with ridge 1e-6 (fourier.py:462-466), inverted by Gauss-Jordan with partial pivoting
    for (int i = 0; i < P; ++i)
      for (int j = 0; j < P; ++j) {
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += Hm[k][i] * Hm[k][j];
        A[i][j] = acc + (i == j ? 1e-6 : 0.0);
        A[i][P + j] = i == j ? 1.0 : 0.0;
      }
    for (int c = 0; c < P; ++c) {
      int piv = c;
      for (int r = c + 1; r < P; ++r)
        if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
      if (A[piv][c] == 0.0) return FFD_ERR_INVALID;
      if (piv != c)
        for (int j = 0; j < 2 * P; ++j) std::swap(A[c][j], A[piv][j]);
      const double inv = 1.0 / A[c][c];
      for (int j = 0; j < 2 * P; ++j) A[c][j] *= inv;
      for (int r = 0; r < P; ++r)
        if (r != c) {
          const double f = A[r][c];
          if (f != 0.0)
            for (int j = 0; j < 2 * P; ++j) A[r][j] -= f * A[c][j];
        }
    }
    // w_k = H_target . (HtH)^-1 . H_k   (prediction = sum_k w_k history_k, fourier.py:476-481)
    for (int k = 0; k < K; ++k) {
      double acc = 0.0;
      for (int i = 0; i < P; ++i) {
        double u = 0.0;
        for (int j = 0; j < P; ++j) u += A[i][P + j] * Hm[k][j];
        acc += Ht[i] * u;
      }
      w[k] = (float)acc;
    }
  }
  return launch_weighted_sum(history, w, out, K, n, (hipStream_t)stream) == hipSuccess ? FFD_OK : FFD_ERR_HIP;
}

int ffd_cache_crf_capture(ffd_ctx* ctx, const ffd_crf_capture_cfg* cfg) {
  if (!ctx) return FFD_ERR_INVALID;
  if (!cfg) {
    ctx->crf_cap_on = false;
    return FFD_OK;
  }
  if ((cfg->ring && (cfg->n_slots < 1 || cfg->every < 1)) || (cfg->last && cfg->last_every < 1))
    return ctx->fail(FFD_ERR_INVALID, "bad CRF capture config (n_slots=%d every=%d last_every=%d)", cfg->n_slots,
                     cfg->every, cfg->last_every);
  ctx->crf_cap = *cfg;
  ctx->crf_cap_on = cfg->ring || cfg->last;
  return FFD_OK;
}

// ---------------------------------------------------------------------------
// cache lifecycle
// ---------------------------------------------------------------------------
int ffd_cache_enable(ffd_ctx* ctx, const ffd_cache_cfg* cfg) {
  if (!ctx) return FFD_ERR_INVALID;
  if (ctx->desc.kind != FFD_MODEL_TRANSFORMER)
    return ctx->fail(FFD_ERR_UNSUPPORTED, "caching is only defined for the transformer backbone (SURVEY Q9)");
  HIPCHECK(hipSetDevice(ctx->device));
  if (cfg) ctx->ccfg = *cfg;
  if (!ctx->kt) {
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->kt, ctx->table_floats()))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->vt, ctx->table_floats()))) return rc;
  }
  ctx->cache_enabled = true;
  return ffd_cache_reset(ctx);
}

int ffd_cache_configure(ffd_ctx* ctx, const ffd_cache_cfg* cfg) {
  if (!ctx) return FFD_ERR_INVALID;
  if (!cfg) return ctx->fail(FFD_ERR_INVALID, "null cache config");
  if (!ctx->cache_enabled) return ctx->fail(FFD_ERR_STATE, "cache not enabled (call ffd_cache_enable)");
  ctx->ccfg = *cfg;
  return FFD_OK;
}

int ffd_cache_disable(ffd_ctx* ctx) {
  if (!ctx) return FFD_ERR_INVALID;
  ctx->cache_enabled = false;
  return FFD_OK;
}

int ffd_cache_reset(ffd_ctx* ctx) {
  if (!ctx) return FFD_ERR_INVALID;
  HIPCHECK(hipSetDevice(ctx->device));
  if (ctx->kt) {
    // an unallocated reference table reads as zeros (cached_transformer.py:252-257)
    HIPCHECK(hipMemset(ctx->kt, 0, ctx->table_floats() * sizeof(float)));
    HIPCHECK(hipMemset(ctx->vt, 0, ctx->table_floats() * sizeof(float)));
  }
  ctx->table_allocated = false;
  ctx->stats = ffd_cache_stats{};
  return FFD_OK;
}

int ffd_cache_stats_get(const ffd_ctx* ctx, ffd_cache_stats* out) {
  if (!ctx || !out) return FFD_ERR_INVALID;
  *out = ctx->stats;
  out->table_allocated = ctx->table_allocated ? 1 : 0;
  return FFD_OK;
}

int ffd_cache_tables_read(ffd_ctx* ctx, float* k_out, float* v_out, void* stream) {
  if (!ctx) return FFD_ERR_INVALID;
  if (!k_out || !v_out) return ctx->fail(FFD_ERR_INVALID, "null buffer");
  if (!ctx->kt) return ctx->fail(FFD_ERR_STATE, "cache not enabled (call ffd_cache_enable)");
  HIPCHECK(hipSetDevice(ctx->device));
  const size_t bytes = ctx->table_floats() * sizeof(float);
  HIPCHECK(hipMemcpyAsync(k_out, ctx->kt, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HIPCHECK(hipMemcpyAsync(v_out, ctx->vt, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return FFD_OK;
}

// ---------------------------------------------------------------------------
// the sampling loop (sampler.py:156-210)
// ---------------------------------------------------------------------------
int ffd_sample_batch(ffd_ctx* ctx, float* x, int B, const float* timesteps, int n_steps, float step_size,
                     int first_step, int n_run, uint64_t seed, uint64_t sample_offset, const float* z_inject,
                     int use_cache, int global_step0, void* stream) {
  if (!ctx) return FFD_ERR_INVALID;
  int rc = check_ready(ctx, B);
  if (rc) return rc;
  if (!x || !timesteps || n_steps < 1 || first_step < 0 || n_run < 0 || first_step + n_run > n_steps)
    return ctx->fail(FFD_ERR_INVALID, "bad argument to ffd_sample_batch (n_steps=%d first=%d run=%d)", n_steps,
                     first_step, n_run);
  if (!(step_size > 0.f)) return ctx->fail(FFD_ERR_INVALID, "step_size must be > 0 (sde.py:157,238)");
  if (use_cache && !ctx->cache_enabled) return ctx->fail(FFD_ERR_STATE, "use_cache without ffd_cache_enable");
  if (use_cache && ctx->desc.kind != FFD_MODEL_TRANSFORMER)
    return ctx->fail(FFD_ERR_UNSUPPORTED, "caching is only defined for the transformer backbone");
  HIPCHECK(hipSetDevice(ctx->device));
  if ((rc = ensure_workspace(ctx, B))) return rc;
  const ffd_model_desc& m = ctx->desc;
  const int d = m.d_model;
  hipStream_t s = (hipStream_t)stream;
  if (ctx->fresca_on && B > ctx->fwork_B) {
    ctx->fwork_B = 0;
    if ((rc = dev_regrow(ctx, &ctx->score2, (size_t)B * m.max_len * m.n_channels))) return rc;
    if ((rc = dev_regrow(ctx, &ctx->fwork, (size_t)B * m.n_channels * (m.max_len / 2 + 1) + 4))) return rc;
    ctx->fwork_B = B;
  }
  // All time embeddings of the trajectory in one launch: t is shared by the batch
  // (sampler.py:59-60).  The table is kept across batches and only rebuilt (with one
  // stream sync) when the timestep grid or the weights changed.
  if ((int)ctx->ts_host.size() != n_steps || memcmp(ctx->ts_host.data(), timesteps, sizeof(float) * n_steps) != 0 ||
      ctx->temb_epoch != ctx->weight_epoch) {
    HIPCHECK(hipStreamSynchronize(s));
    if (n_steps > ctx->temb_cap) {
      ctx->temb_cap = 0;
      if ((rc = dev_regrow(ctx, &ctx->temb_tab, (size_t)n_steps * d))) return rc;
      if ((rc = dev_regrow(ctx, &ctx->ts_dev, (size_t)n_steps))) return rc;
      ctx->temb_cap = n_steps;
    }
    ctx->ts_host.assign(timesteps, timesteps + n_steps);
    HIPCHECK(hipMemcpy(ctx->ts_dev, ctx->ts_host.data(), sizeof(float) * n_steps, hipMemcpyHostToDevice));
    HIPCHECK(launch_time_embed(ctx->ts_dev, 0.f, n_steps, ctx->raw["time_encoder.W"].p,
                               ctx->raw["time_encoder.dense.weight"].p, ctx->raw["time_encoder.dense.bias"].p,
                               ctx->temb_tab, d, s));
    ctx->temb_epoch = ctx->weight_epoch;
  }
  const size_t slab = (size_t)B * m.max_len * m.n_channels;
  const uint64_t elem_off = sample_offset * (uint64_t)m.max_len * m.n_channels;
  // Without FreSca the score is consumed only by the SDE step: unembed inside the step kernel (one launch and a
  // (B, L, C) round trip less per step).  FreSca needs the whole score (FFT along L).
  const bool fuse_tail = g_fuse_tail && !ctx->fresca_on && m.kind != FFD_MODEL_MLP &&
                         unembed_sde_supported(m.n_channels, d) && (m.n_channels % 4 != 0 || elem_off % 4 == 0) &&
                         (m.n_channels % 4 != 0 || reinterpret_cast<uintptr_t>(x) % 16 == 0) &&  // float4 x rows
                         (!z_inject || m.n_channels % 4 != 0 ||  // float4 reads of the injected noise only when C % 4 == 0
                          (reinterpret_cast<uintptr_t>(z_inject) % 16 == 0 && slab % 4 == 0));
  for (int j = 0; j < n_run; ++j) {
    const int i = first_step + j;
    int n_rec = -1;
    if (use_cache) {
      const int gstep = global_step0 + j;
      ctx->stats.current_step = gstep;
      n_rec = ffd_host_gate(gstep, m.max_len, ctx->ccfg.K, ctx->ccfg.R);
    }
    float* crf_dst = nullptr;
    float* crf_copy = nullptr;
    if (use_cache && ctx->crf_cap_on) {  // cache.update_crf(crf) with current_step == global step (sampler.py:70-73)
      const ffd_crf_capture_cfg& cc = ctx->crf_cap;
      const int gstep = global_step0 + j;
      const size_t crf_n = (size_t)m.num_layers * m.max_len * d;
      if (cc.ring && gstep % cc.every == 0) crf_dst = cc.ring + (size_t)((gstep / cc.every) % cc.n_slots) * crf_n;
      if (cc.last && gstep % cc.last_every == 0) {
        const int rest = n_run - 1 - j;  // is there a later qualifying step in this call?
        const int to_next = cc.last_every - (gstep % cc.last_every);
        if (to_next > rest) {
          if (crf_dst) crf_copy = cc.last; else crf_dst = cc.last;
        }
      }
    }
    const float* hidden = nullptr;
    if ((rc = forward_impl(ctx, x, ctx->temb_tab + (size_t)i * d, 0, ctx->score, crf_dst, B, n_rec, s,
                           fuse_tail ? &hidden : nullptr)))
      return rc;
    if (crf_copy)
      HIPCHECK(hipMemcpyAsync(crf_copy, crf_dst, sizeof(float) * (size_t)m.num_layers * m.max_len * d,
                              hipMemcpyDeviceToDevice, s));
    if (use_cache) ctx->stats.current_step = i;  // sampler.py:73-74 (Q4)
    const double t = (double)timesteps[i];
    const float* score = ctx->score;
    if (ctx->fresca_on) {  // sampler.py:79-93 -> fresca.py:220-268
      const ffd_fresca_cfg& f = ctx->fcfg;
      double h = (double)f.high_scale;
      if (f.num_steps > 0 && h > 1.0) h = (1.0 - t / (double)f.num_steps) * (h - 1.0) + 1.0;
      if (!((double)f.low_scale == 1.0 && h == 1.0)) {  // fresca.py:137-138 early exit
        HIPCHECK(launch_fresca(ctx->score, ctx->score2, ctx->fwork, B, m.max_len, m.n_channels, f.low_scale, (float)h,
                               f.cutoff_ratio, f.strategy, s));
        score = ctx->score2;
      }
    }
    if (hidden)
      TIMED(FFD_K_SDE, launch_unembed_sde(hidden, ctx->raw["unembedder.weight"].p, ctx->raw["unembedder.bias"].p, x,
                                          z_inject ? z_inject + (size_t)j * slab : nullptr, ctx->G_dev,
                                          sde_params(m.sde, m.sde_a, m.sde_b, t, step_size), seed, elem_off, (uint32_t)i,
                                          B, m.max_len, m.n_channels, d, s));
    else
      TIMED(FFD_K_SDE, launch_sde_step(x, score, z_inject ? z_inject + (size_t)j * slab : nullptr, ctx->G_dev,
                                       sde_params(m.sde, m.sde_a, m.sde_b, t, step_size), seed, elem_off, (uint32_t)i, B,
                                       m.max_len, m.n_channels, s));
  }
  return FFD_OK;
}

// ---------------------------------------------------------------------------
// benchmark introspection
// ---------------------------------------------------------------------------
double ffd_flops_per_sample_step(const ffd_ctx* ctx, int cache_hit) {
  if (!ctx) return 0.0;
  const ffd_model_desc& m = ctx->desc;
  const double L = m.max_len, d = m.d_model, C = m.n_channels, NL = m.num_layers, F = m.dim_feedforward;
  if (m.kind == FFD_MODEL_LSTM) return NL * 2.0 * L * (2.0 * 4.0 * d * d) + 4.0 * L * C * d + 2.0 * d * d;
  if (m.kind == FFD_MODEL_MLP) return NL * 4.0 * d * F + 4.0 * L * C * d + 2.0 * d * d;
  double per_layer = 2.0 * L * d * 3.0 * d + 2.0 * L * L * d + 2.0 * L * L * d + 2.0 * L * d * d + 4.0 * L * d * F;
  if (cache_hit) per_layer -= 2.0 * L * d * 2.0 * d;
  return NL * per_layer + 4.0 * L * C * d + 2.0 * d * d;
}

double ffd_ffn_flops_per_launch(const ffd_ctx* ctx, int B) {
  if (!ctx) return 0.0;
  const ffd_model_desc& m = ctx->desc;
  return 4.0 * (double)B * m.max_len * m.d_model * m.dim_feedforward;
}

int ffd_lstm_trace(ffd_ctx* ctx, unsigned long long* host_out, int capacity_units, int* n_units_out) {
  if (!ctx || capacity_units < 1) return FFD_ERR_INVALID;
  HIPCHECK(hipSetDevice(ctx->device));
  if (host_out == nullptr) {  // begin: the next k_lstm_wave launches write their units' records
    if (ctx->lstm_trace) (void)hipFree(ctx->lstm_trace);
    ctx->lstm_trace = nullptr;
    HIPCHECK(hipMalloc(&ctx->lstm_trace, sizeof(unsigned long long) * 4 * (size_t)capacity_units));
    HIPCHECK(hipMemset(ctx->lstm_trace, 0, sizeof(unsigned long long) * 4 * (size_t)capacity_units));
    ctx->lstm_trace_units = capacity_units;
    return FFD_OK;
  }
  if (!ctx->lstm_trace) return ctx->fail(FFD_ERR_STATE, "ffd_lstm_trace: no trace begun");
  HIPCHECK(hipDeviceSynchronize());
  const int n = capacity_units < ctx->lstm_trace_units ? capacity_units : ctx->lstm_trace_units;
  HIPCHECK(hipMemcpy(host_out, ctx->lstm_trace, sizeof(unsigned long long) * 4 * (size_t)n, hipMemcpyDeviceToHost));
  (void)hipFree(ctx->lstm_trace);
  ctx->lstm_trace = nullptr;
  if (n_units_out) *n_units_out = n;
  return FFD_OK;
}

int ffd_kernel_timing_begin(ffd_ctx* ctx, uint32_t class_mask, int max_launches) {
  if (!ctx) return FFD_ERR_INVALID;
  if (max_launches < 1 || max_launches > 100000) return ctx->fail(FFD_ERR_INVALID, "max_launches=%d", max_launches);
  HIPCHECK(hipSetDevice(ctx->device));
  while (ctx->ev.size() < (size_t)2 * max_launches) {
    hipEvent_t e;
    HIPCHECK(hipEventCreate(&e));
    ctx->ev.push_back(e);
  }
  ctx->ev_used = 0;
  ctx->ev_cls.clear();
  ctx->time_mask = class_mask;
  return FFD_OK;
}

int ffd_kernel_timing_end(ffd_ctx* ctx) {
  if (!ctx) return FFD_ERR_INVALID;
  ctx->time_mask = 0;
  HIPCHECK(hipSetDevice(ctx->device));
  double tot[FFD_K_COUNT] = {0};
  for (int c = 0; c < FFD_K_COUNT; ++c) ctx->tm_n[c] = 0;
  const size_t n = ctx->ev_used / 2;
  for (size_t i = 0; i < n; ++i) {
    HIPCHECK(hipEventSynchronize(ctx->ev[2 * i + 1]));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[2 * i], ctx->ev[2 * i + 1]));
    const int c = ctx->ev_cls[i];
    tot[c] += ms;
    ctx->tm_n[c]++;
  }
  for (int c = 0; c < FFD_K_COUNT; ++c) ctx->tm_ms[c] = ctx->tm_n[c] ? (float)(tot[c] / ctx->tm_n[c]) : 0.f;
  ctx->ev_used = 0;
  ctx->ev_cls.clear();
  return FFD_OK;
}

int ffd_kernel_timing_get(const ffd_ctx* ctx, int kernel_class, float* avg_ms_out, int* launches_out) {
  if (!ctx || kernel_class < 0 || kernel_class >= FFD_K_COUNT || !avg_ms_out || !launches_out) return FFD_ERR_INVALID;
  *avg_ms_out = ctx->tm_ms[kernel_class];
  *launches_out = ctx->tm_n[kernel_class];
  return FFD_OK;
}

const char* ffd_kernel_work(const ffd_ctx* ctx, int kernel_class, int B, int cache_hit, double* flops_out,
                            double* bytes_out) {
  if (!ctx || B < 1) return nullptr;
  const ffd_model_desc& m = ctx->desc;
  const double L = m.max_len, d = m.d_model, C = m.n_channels, F = m.dim_feedforward, M = (double)B * L;
  const bool tr = m.kind == FFD_MODEL_TRANSFORMER, ls = m.kind == FFD_MODEL_LSTM;
  double fl = 0.0, by = 0.0;
  const char* name = nullptr;
  switch (kernel_class) {
    case FFD_K_FFN:  // 4 d F FLOP per row; x in, y out, both weight matrices once
      if (tr && g_ffn_split) name = "k_ffn_ln_split", fl = 4.0 * M * d * F, by = 4.0 * (2.0 * M * d) + 6.0 * (2.0 * d * F);
      else if (tr && g_ffn_height == 1 && ffn_height_plan((int)M, m.d_model, m.dim_feedforward))  // + the out-projection it absorbs
        name = "k_ffn_ln<oproj>", fl = 4.0 * M * d * F + 2.0 * M * d * d, by = 4.0 * (3.0 * M * d + 2.0 * d * F + d * d);
      else if (tr && small_path_splits((int)M, m.d_model, m.dim_feedforward))  // + the out-projection it absorbs
        name = "k_oproj_ffn_split + k_ffn_reduce_ln", fl = 4.0 * M * d * F + 2.0 * M * d * d,
        by = 4.0 * (3.0 * M * d + 2.0 * d * F + d * d);
      else if (int a_ = 0, b_ = 0, u_ = 0; tr && rows_slice_plan((int)M, m.d_model, m.dim_feedforward, &a_, &b_, &u_)) {
        if (u_) name = "k_ffn_rows<sliced> + k_rows_reduce_ln", fl = 4.0 * M * d * F, by = 4.0 * (2.0 * M * d + 2.0 * d * F);
        else name = "k_ffn_rows<oproj, sliced> + k_rows_reduce_ln", fl = 4.0 * M * d * F + b_ * 2.0 * M * d * d,
             by = 4.0 * (3.0 * M * d + 2.0 * d * F + d * d);
      }
      else if (tr && mid_path_splits((int)M, m.d_model, m.dim_feedforward))
        name = "k_ffn_part", fl = 4.0 * M * d * F, by = 4.0 * (2.0 * M * d + 2.0 * d * F);
      else if (tr && ffn_rows_fused_selected((int)M, m.d_model, m.dim_feedforward))  // + the out-projection it absorbs
        name = "k_ffn_rows<oproj>", fl = 4.0 * M * d * F + 2.0 * M * d * d, by = 4.0 * (3.0 * M * d + 2.0 * d * F + d * d);
      else if (tr && ffn_rows_selected((int)M, m.d_model, m.dim_feedforward))
        name = "k_ffn_rows", fl = 4.0 * M * d * F, by = 4.0 * (2.0 * M * d + 2.0 * d * F);
      else if (tr) name = "k_ffn_ln", fl = 4.0 * M * d * F, by = 4.0 * (2.0 * M * d + 2.0 * d * F);
      break;
    case FFD_K_ATTN:  // in-projection (Q only on a pure-cache step) + QK^T + PV; x in, attention output out
      if (tr) {
        name = "k_qkv_attention";
        fl = M * (2.0 * d * (cache_hit ? d : 3.0 * d) + 4.0 * L * d);
        by = 4.0 * (2.0 * M * d + 3.0 * d * d + (cache_hit ? 2.0 * L * d : 0.0));
      }
      break;
    case FFD_K_OUTPROJ:  // attention output + residual in, LN1 output out
      if (tr && !g_ffn_split && g_ffn_height == 1 && ffn_height_plan((int)M, m.d_model, m.dim_feedforward)) break;  // inside k_ffn_ln<oproj>
      if (int a_ = 0, b_ = 0, u_ = 0; !g_ffn_split && tr && !small_path_splits((int)M, m.d_model, m.dim_feedforward) &&
                                       rows_slice_plan((int)M, m.d_model, m.dim_feedforward, &a_, &b_, &u_)) {
        if (u_) name = "k_linear_res_ln", fl = 2.0 * M * d * d, by = 4.0 * (3.0 * M * d + d * d);
        break;
      }
      if (tr && (g_ffn_split || (!small_path_splits((int)M, m.d_model, m.dim_feedforward) &&
                                 !(!mid_path_splits((int)M, m.d_model, m.dim_feedforward) &&
                                   ffn_rows_fused_selected((int)M, m.d_model, m.dim_feedforward)))))
        name = "k_linear_res_ln", fl = 2.0 * M * d * d, by = 4.0 * (3.0 * M * d + d * d);
      break;
    case FFD_K_LSTM_REC:
      if (ls && lstm_wave_selected(B, m.d_model)) {  // every layer in one launch: x W_ih^T + h W_hh^T
        // per launch: batches past a 16-sample tile per CU go in sub-batches (exact where B is a multiple of that)
        const double Ml = (double)(B < lstm_wave_max_batch(L, m.d_model) ? B : lstm_wave_max_batch(L, m.d_model)) * L;
        name = "k_lstm_wave", fl = m.num_layers * 2.0 * Ml * 8.0 * d * d, by = m.num_layers * 4.0 * (2.0 * Ml * d + 8.0 * d * d);
      }
      else if (ls)  // h W_hh^T for L cell steps; gate pre-activations + residual rows in, rows out
        name = "k_lstm_layer", fl = 2.0 * M * 4.0 * d * d, by = 4.0 * (M * 4.0 * d + 2.0 * M * d + 4.0 * d * d);
      break;
    case FFD_K_LSTM_GATES:
      if (ls && !lstm_wave_selected(B, m.d_model))
        name = "k_linear_rm", fl = 2.0 * M * 4.0 * d * d, by = 4.0 * (M * d + M * 4.0 * d + 4.0 * d * d);
      break;
    case FFD_K_SDE:
      // x, score in; x out (noise generated on chip): 12 B per element; with the unembedding fused
      // into it the score term is replaced by the hidden row: 4 (d + 2 C) B per row
      if (g_fuse_tail && !ctx->fresca_on && m.kind != FFD_MODEL_MLP && unembed_sde_supported(m.n_channels, m.d_model))
        name = "k_unembed_mfma<sde>", fl = 2.0 * M * C * d, by = 4.0 * M * (d + 2.0 * C);
      else
        name = "k_sde_step", by = 12.0 * M * C;
      break;
    case FFD_K_EMBED:
      if (m.kind != FFD_MODEL_MLP) name = "k_embed", fl = 2.0 * M * C * d, by = 4.0 * M * (C + d);
      break;
    case FFD_K_UNEMBED:
      if (m.kind != FFD_MODEL_MLP) name = "k_unembed", fl = 2.0 * M * C * d, by = 4.0 * M * (C + d);
      break;
    default: break;
  }
  if (flops_out) *flops_out = fl;
  if (bytes_out) *bytes_out = by;
  return name;
}

int ffd_bench_ffn(ffd_ctx* ctx, int B, int iters, float* ms_out, void* stream) {
  if (!ctx) return FFD_ERR_INVALID;
  int rc = check_ready(ctx, B);
  if (rc) return rc;
  if (ctx->desc.kind != FFD_MODEL_TRANSFORMER) return ctx->fail(FFD_ERR_UNSUPPORTED, "no FFN in the LSTM backbone");
  if (iters < 1 || !ms_out) return ctx->fail(FFD_ERR_INVALID, "bad argument to ffd_bench_ffn");
  HIPCHECK(hipSetDevice(ctx->device));
  if ((rc = ensure_workspace(ctx, B))) return rc;
  const ffd_model_desc& m = ctx->desc;
  const int M = B * m.max_len, d = m.d_model;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fill_hash, dim3(1024), dim3(256), 0, s, ctx->h1, (size_t)M * d, 0x9E3779B9u);
  HIPCHECK(hipGetLastError());
  // (the form the forward pass takes at this size: with the out-projection + LN1 inside it where that is selected)
  const bool fused = !g_ffn_split && ctx->layers[0].ring_op && !mid_path_splits(M, d, m.dim_feedforward) &&
                     ffn_rows_fused_selected(M, d, m.dim_feedforward);
  if (fused) {
    hipLaunchKernelGGL(k_fill_hash, dim3(1024), dim3(256), 0, s, ctx->attn, (size_t)M * d, 0x85EBCA6Bu);
    HIPCHECK(hipGetLastError());
  }
  auto run = [&]() -> hipError_t {
    return fused ? launch_oproj_ffn_rows(ctx->attn, ctx->h1, ctx->layers[0], ctx->h0, M, d, m.dim_feedforward, s)
                 : launch_ffn_ln(ctx->h1, ctx->layers[0], ctx->h0, M, d, m.dim_feedforward, s);
  };
  for (int i = 0; i < 3; ++i) HIPCHECK(run());
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  HIPCHECK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) HIPCHECK(run());
  HIPCHECK(hipEventRecord(e1, s));
  HIPCHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_out = ms / iters;
  return FFD_OK;
}

int ffd_probe_ffn_clock(ffd_ctx* ctx, int B, double warm_seconds, double* ghz_out, double* loop_us_out,
                        unsigned long long* raw_out, int raw_capacity, int* nwg_out, void* stream) {
  if (!ctx) return FFD_ERR_INVALID;
  int rc = check_ready(ctx, B);
  if (rc) return rc;
  if (ctx->desc.kind != FFD_MODEL_TRANSFORMER) return ctx->fail(FFD_ERR_UNSUPPORTED, "no FFN in this backbone");
  if (!ghz_out || !(warm_seconds >= 0.0) || warm_seconds > 30.0) return ctx->fail(FFD_ERR_INVALID, "bad argument");
  HIPCHECK(hipSetDevice(ctx->device));
  if ((rc = ensure_workspace(ctx, B))) return rc;
  const ffd_model_desc& m = ctx->desc;
  const int M = B * m.max_len, d = m.d_model;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fill_hash, dim3(1024), dim3(256), 0, s, ctx->h1, (size_t)M * d, 0x9E3779B9u);  // random data
  HIPCHECK(hipGetLastError());
  const bool split = g_ffn_split && ctx->layers[0].w1s != nullptr;  // (packed by the first forward with ffn_split on)
  if (g_ffn_split && !split) return ctx->fail(FFD_ERR_STATE, "ffn_split: run one forward first (the packs are made on first use)");
  const int nwg = split ? (cdiv(M, 64) < num_cus() ? cdiv(M, 64) : num_cus()) : cdiv(M, ffn_tile_rows(M));  // upper bound of the grid (the persistent form launches fewer)
  const bool fused = !split && ctx->layers[0].ring_op && !mid_path_splits(M, d, m.dim_feedforward) &&
                     ffn_rows_fused_selected(M, d, m.dim_feedforward);
  if (fused) {
    hipLaunchKernelGGL(k_fill_hash, dim3(1024), dim3(256), 0, s, ctx->attn, (size_t)M * d, 0x85EBCA6Bu);
    HIPCHECK(hipGetLastError());
  }
  auto launch = [&](unsigned long long* st) {
    return split ? launch_ffn_ln_split(ctx->h1, ctx->layers[0], ctx->h0, M, d, m.dim_feedforward, s, st)
           : fused ? launch_oproj_ffn_rows(ctx->attn, ctx->h1, ctx->layers[0], ctx->h0, M, d, m.dim_feedforward, s, st)
                   : launch_ffn_ln(ctx->h1, ctx->layers[0], ctx->h0, M, d, m.dim_feedforward, s, st);
  };
  unsigned long long* stamps = nullptr;
  HIPCHECK(hipMalloc((void**)&stamps, sizeof(unsigned long long) * 8 * nwg));
  HIPCHECK(hipMemsetAsync(stamps, 0, sizeof(unsigned long long) * 8 * nwg, s));
  // back-to-back launches for warm_seconds (the clock the chip settles at under this load), then the stamped one
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  HIPCHECK(hipEventRecord(e0, s));
  double elapsed = 0.0;
  while (elapsed < warm_seconds) {
    for (int i = 0; i < 50; ++i)
      HIPCHECK(launch(nullptr));
    HIPCHECK(hipEventRecord(e1, s));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    elapsed = ms * 1e-3;
  }
  HIPCHECK(launch(stamps));
  std::vector<unsigned long long> h(8 * (size_t)nwg);
  HIPCHECK(hipMemcpyAsync(h.data(), stamps, sizeof(unsigned long long) * 8 * nwg, hipMemcpyDeviceToHost, s));
  HIPCHECK(hipStreamSynchronize(s));
  (void)hipFree(stamps);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  std::vector<double> ghz, us;
  int used = 0;
  for (int i = 0; i < nwg; ++i)
    if (h[8 * i + 1] > 0) {
      ghz.push_back((double)h[8 * i] / (double)h[8 * i + 1] * 0.1);  // shader cycles per 10 ns tick
      us.push_back((double)h[8 * i + 1] * 0.01);
      if (raw_out && used < raw_capacity) memcpy(raw_out + 8 * (size_t)used, &h[8 * i], sizeof(unsigned long long) * 8);
      ++used;
    }
  if (nwg_out) *nwg_out = used;
  if (ghz.empty()) return ctx->fail(FFD_ERR_STATE, "no stamps were written");
  std::sort(ghz.begin(), ghz.end());
  std::sort(us.begin(), us.end());
  *ghz_out = ghz[ghz.size() / 2];
  if (loop_us_out) *loop_us_out = us[us.size() / 2];
  return FFD_OK;
}

int ffd_async_status(ffd_ctx* ctx) {
  if (!ctx) return FFD_ERR_INVALID;
  return check_async(ctx);
}

int ffd_probe_attn(ffd_ctx* ctx, int B, int n_recompute, double warm_seconds, int iters, float* ms_out,
                   unsigned long long* raw_out, int raw_capacity, int* nrec_out, void* stream) {
  if (!ctx) return FFD_ERR_INVALID;
  int rc = check_ready(ctx, B);
  if (rc) return rc;
  const ffd_model_desc& m = ctx->desc;
  if (m.kind != FFD_MODEL_TRANSFORMER) return ctx->fail(FFD_ERR_UNSUPPORTED, "no attention in this backbone");
  if (!ms_out || iters < 1 || !(warm_seconds >= 0.0) || warm_seconds > 30.0) return ctx->fail(FFD_ERR_INVALID, "bad argument");
  if (!ctx->packed[0].aw_full) return ctx->fail(FFD_ERR_UNSUPPORTED, "no fused attention kernel for this shape");
  const int L = m.max_len, d = m.d_model, H = m.n_head, hd = d / H;
  if (n_recompute > L) return ctx->fail(FFD_ERR_INVALID, "n_recompute=%d outside [-1,%d]", n_recompute, L);
  enum { STD, PURE, MIXED } mode = STD;
  if (n_recompute >= 0 && (double)n_recompute <= 0.8 * (double)L) mode = n_recompute == 0 ? PURE : MIXED;
  if (mode != STD && !(ctx->cache_enabled && ctx->table_allocated))
    return ctx->fail(FFD_ERR_STATE, "cached modes need ffd_cache_enable and one full step (the tables)");
  HIPCHECK(hipSetDevice(ctx->device));
  if ((rc = ensure_workspace(ctx, B))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const size_t M = (size_t)B * L;
  hipLaunchKernelGGL(k_fill_hash, dim3(1024), dim3(256), 0, s, ctx->h1, M * d, 0x9E3779B9u);  // random rows
  HIPCHECK(hipGetLastError());
  const LayerPacked& pk = ctx->packed[0];
  const int hpw = pk.aw_full2 ? qkv_attention_hpw(d, hd, L, B) : 1;
  const float* pack = hpw == 2 ? (mode == PURE ? pk.aw_q2 : pk.aw_full2) : (mode == PURE ? pk.aw_q : pk.aw_full);
  const int n_own = mode == PURE ? 0 : mode == MIXED ? n_recompute : L;
  // (MIXED: the recomputed rows of batch element 0 are NOT written back -- the probe leaves the tables as they are)
  auto launch = [&](unsigned long long* st) {
    return launch_qkv_attention(ctx->h1, pack, hpw, mode == PURE, mode != STD ? ctx->kt : nullptr,
                                mode != STD ? ctx->vt : nullptr, nullptr, nullptr, ctx->attn, B, L, d, hd, n_own, s, st);
  };
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  HIPCHECK(hipEventRecord(e0, s));
  double elapsed = 0.0;
  do {
    for (int i = 0; i < 20; ++i) HIPCHECK(launch(nullptr));
    HIPCHECK(hipEventRecord(e1, s));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    elapsed = ms * 1e-3;
  } while (elapsed < warm_seconds);
  HIPCHECK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) HIPCHECK(launch(nullptr));
  HIPCHECK(hipEventRecord(e1, s));
  HIPCHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_out = ms / iters;
  if (nrec_out) *nrec_out = 0;
  if (raw_out && raw_capacity > 0) {
    const size_t nwave = (size_t)B * H * 4;  // upper bound: at most 4 waves per (sample, head) workgroup / head pair
    unsigned long long* stamps = nullptr;
    HIPCHECK(hipMalloc((void**)&stamps, sizeof(unsigned long long) * 16 * nwave));
    HIPCHECK(hipMemsetAsync(stamps, 0, sizeof(unsigned long long) * 16 * nwave, s));
    hipError_t e = launch(stamps);
    if (e != hipSuccess) {
      (void)hipFree(stamps);
      return ctx->fail(FFD_ERR_UNSUPPORTED, "no stamped twin of the attention kernel for this shape / mode");
    }
    std::vector<unsigned long long> h(16 * nwave);
    HIPCHECK(hipMemcpyAsync(h.data(), stamps, sizeof(unsigned long long) * 16 * nwave, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    (void)hipFree(stamps);
    int used = 0;
    for (size_t i = 0; i < nwave && used < raw_capacity; ++i)
      if (h[16 * i + 1] != 0) memcpy(raw_out + 16 * (size_t)used++, &h[16 * i], sizeof(unsigned long long) * 16);
    if (nrec_out) *nrec_out = used;
  }
  return FFD_OK;
}

}  // extern "C"
