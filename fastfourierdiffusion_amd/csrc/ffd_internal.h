// Internal declarations shared by the libffd translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ffd {

constexpr int WAVE = 64;

// d_model / head_dim values with compiled kernels (every d % 4 == 0, d <= 72; hd <= 8)
#define FFD_D_LIST(X) X(8) X(16) X(24) X(32) X(48) X(60) X(64) X(72)
#define FFD_HD_LIST(X) X(2) X(3) X(4) X(5) X(6) X(8)

// ---- MFMA f32 16x16x4 fragment conventions (cdna_hip_programming.md §3) ----
//   A operand: lane l holds A[i = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D      : lane l, reg r holds D[i = 4 * (l >> 4) + r][j = l & 15]
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr __host__ __device__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- LDS-DMA issued by hand + counted waits (k_ffn_rows, k_linear_res_ln) ----
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS bytes [lds_byte, lds_byte + 1024).
// Issued from inline asm on purpose: for the builtin form hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of the
// next ds_read of the same __shared__ array (it cannot tell the ring's slots apart), which drains the ring every slot.
// The kernels order DMA and reads themselves: counted vmcnt (+ s_barrier where waves share the image).
__device__ __forceinline__ void dma_piece(const float* g, unsigned lds_byte) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte), "v"(g)
               : "memory");  // (m0 is a reserved register: hipcc keeps nothing in it across statements)
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(unsigned long)((const __attribute__((address_space(3))) float*)p);
}


// Padded LDS row stride (in floats) for a (rows x D) fp32 tile whose MFMA
// fragments are read with ds_read_b32 as tile[(l&15)*stride + 4s + (l>>4)]:
// stride/2 odd => the 16 rows x 2 k of each 32-lane half hit 32 distinct banks.
constexpr __host__ __device__ int lds_stride(int D) { return D + ((6 - D % 4) % 4); }

// ---- packed weight layouts -------------------------------------------------
// "dpack": a (N x D) weight (PyTorch (out,in) order) packed as the A operand of
// Y^T = W X^T :  [nt = N/16 tiles][g = ceil(ceil(D/4)/4)][lane 64][j 4]
//   = W[16 nt + (lane & 15)][4 (4 g + j) + (lane >> 4)]   (0 outside N x D)
// so that one coalesced float4 load per lane yields 4 consecutive k-steps.
constexpr __host__ __device__ int dpack_groups(int D) { return cdiv(cdiv(D, 4), 4); }
constexpr __host__ __device__ size_t dpack_floats(int N, int D) {
  return (size_t)cdiv(N, 16) * dpack_groups(D) * 64 * 4;
}
// "w2pack": linear2.weight (D x F) packed as the A operand of Y^T += W2 H^T with
// the GEMM1 accumulator as B:  [fc = F/16][ct = ceil(D/16)][lane 64][r 4]
//   = W2[16 ct + (lane & 15)][16 fc + 4 (lane >> 4) + r]      (0 for c >= D)
constexpr __host__ __device__ size_t w2pack_floats(int D, int F) { return (size_t)(F / 16) * cdiv(D, 16) * 64 * 4; }

// ---- launchers (each returns the hipError_t of the launch) -----------------

struct LayerWeights {
  // raw (device) parameters
  const float *in_w, *in_b, *out_w, *out_b, *w1, *b1, *w2, *b2, *n1w, *n1b, *n2w, *n2b;
  // packed
  const float *in_wp;   // dpack (3d x d)
  const float *out_wp;  // dpack (d x d)
  const float *w1p;     // dpack (F x d)
  const float *w2p;     // w2pack
  const float *w2r;     // w2rem (remainder rows d % 16 of linear2.weight, 4x4x1 MFMA A-operand order)
  const float* ring = nullptr;  // CU-shared weight ring pack of the row-owning FFN (ffd_ffn_rows.hip)
  const float* ring_op = nullptr;  // its out-projection slot (fused out-proj + LN1 form)
  const void *w1s = nullptr, *w2s = nullptr;  // three-part bf16 packs of the opt-in split FFN (ffd_ffn_split.hip)
};

hipError_t launch_pack_dweight(const float* W, float* Wp, int N, int D, hipStream_t s);
hipError_t launch_pack_w2(const float* W2, float* W2p, int D, int F, hipStream_t s);
// "w2rem": rows c >= 16*(D/16) of linear2.weight (D x F), in groups of 4, as the A operand of
// v_mfma_f32_4x4x1_16b_f32 with a GEMM1 accumulator register r as B (block b = lane>>2 pairs
// A[lane 4b+i] with B[lane 4b+j]):  [fc = F/16][g][lane 64][r 4]
//   = W2[16*(D/16) + 4 g + (lane & 3)][16 fc + 4 (lane >> 4) + r]
constexpr __host__ __device__ int w2rem_groups(int D) { return (D % 16) / 4; }
constexpr __host__ __device__ size_t w2rem_floats(int D, int F) { return (size_t)(F / 16) * (w2rem_groups(D) ? w2rem_groups(D) : 1) * 64 * 4; }
hipError_t launch_pack_w2rem(const float* W2, float* W2r, int D, int F, hipStream_t s);
hipError_t launch_renorm_rows(float* W, int rows, int D, float max_norm, hipStream_t s);
hipError_t launch_renorm_rows_once(float* W, int rows, int D, float max_norm, hipStream_t s);
hipError_t launch_add_table(const float* x, const float* rowtab, const float* battab, float* out, int B, int L, int D,
                            hipStream_t s);

// temb[n][d] = dense(gamma(t_n)) for n timesteps (transformer.py:77-91)
// (ts == nullptr: a single embedding of the immediate t_imm)
hipError_t launch_time_embed(const float* ts, float t_imm, int n, const float* W, const float* dense_w,
                             const float* dense_b, float* temb, int D, hipStream_t s);
// h[b,l,:] = X[b,l,:] We^T + be (+ pos[l,:]) + temb[b * temb_stride + :]   (temb_stride 0: one embedding for the batch)
hipError_t launch_embed(const float* X, const float* We, const float* be, const float* pos, const float* temb,
                        int temb_stride, float* h, int B, int L, int C, int D, hipStream_t s);
// score[b,l,c] = h[b,l,:] . Wu[c,:] + bu[c]
hipError_t launch_unembed(const float* h, const float* Wu, const float* bu, float* score, int M, int C, int D,
                          hipStream_t s);

struct SdeParams {
  int sde;          // 0 VP 1 VE
  float a;          // VP: (float)(-0.5*beta)        VE: unused
  float cs;         // (float)sqrt(beta) | (float)sqrt_derivative
  float dt, sqdt;   // step_size, sqrtf(step_size)
};
hipError_t launch_sde_step(float* x, const float* score, const float* z, const float* G, SdeParams p, uint64_t seed,
                           uint64_t elem_offset, uint32_t step, int B, int L, int C, hipStream_t s);
// the two fused: x <- step(x, unembed(h)); the score stays in registers.  Needs unembed_sde_supported(C, D).
bool unembed_sde_supported(int C, int D);
hipError_t launch_unembed_sde(const float* h, const float* Wu, const float* bu, float* x, const float* z, const float* G,
                              SdeParams p, uint64_t seed, uint64_t elem_offset, uint32_t step, int B, int L, int C,
                              int D, hipStream_t s);
hipError_t launch_prior(float* x, const float* z, const float* G, float scale, uint64_t seed, uint64_t elem_offset,
                        int B, int L, int C, hipStream_t s);

// Y[M x N] (row stride ldy) = X[M x D] Wp^T + b ; Wp is dpack of the N x D weight.
hipError_t launch_linear(const float* X, const float* Wp, const float* bias, float* Y, int M, int N, int D, int ldy,
                         hipStream_t s);
// Y[M x D] = LayerNorm(R + X Wp^T + b) * g + beta    (out-proj + residual + LN1)
hipError_t launch_linear_res_ln(const float* X, const float* Wp, const float* bias, const float* R, const float* g,
                                const float* beta, float* Y, int M, int D, hipStream_t s);
// Fused FFN: Y = LN2(X + W2 relu(W1 X + b1) + b2)
// stamp != nullptr (diagnostics): per-workgroup (shader-clock, 100 MHz real-time) deltas around the main loop
hipError_t launch_ffn_ln(const float* X, const LayerWeights& w, float* Y, int M, int D, int F, hipStream_t s,
                         unsigned long long* stamp = nullptr);
int ffn_tile_rows(int M);
// 16-row tiles per CU between 1.4 and 3: k_ffn_ln at 32 / 48 rows per workgroup (one tile per CU); 0 = another form
int ffn_height_plan(int M, int D, int F);
// ... as ONE launch, out-projection + LN1 inside (g_ffn_height == 1); Y may be Rres
hipError_t launch_oproj_ffn_ln(const float* attn, const float* Rres, const LayerWeights& w, float* Y, int M, int D, int F,
                               hipStream_t s);
extern thread_local int g_ffn_height;
// Row-owning FFN with a CU-shared LDS weight ring (ffd_ffn_rows.hip): the large-M form
bool ffn_rows_supported(int D, int F);
bool ffn_rows_selected(int M, int D, int F);
size_t ffn_ring_floats(int D, int F);
hipError_t launch_pack_ffn_ring(const float* W1, const float* b1, const float* W2, float* out, int D, int F, hipStream_t s);
hipError_t launch_ffn_rows(const float* X, const LayerWeights& w, float* Y, int M, int D, int F, hipStream_t s,
                           unsigned long long* stamp = nullptr);
extern thread_local int g_ffn_rows, g_ffn_rows_nw, g_ffn_rows_cps, g_ffn_rows_fuse;
// out-proj + LN1 + FFN + LN2 in one launch (the fused form of k_ffn_rows); Y must not alias attn / Rin
bool ffn_rows_fused_selected(int M, int D, int F);
size_t ffn_ring_oproj_floats(int D);
hipError_t launch_pack_oproj_ring(const float* Wo, float* out, int D, hipStream_t s);
hipError_t launch_oproj_ffn_rows(const float* attn, const float* Rin, const LayerWeights& w, float* Y, int M, int D,
                                 int F, hipStream_t s, unsigned long long* stamp = nullptr);
// mid-size M: the fused kernel over tiles x slices of the hidden dimension + a reduce / LN2 launch
extern thread_local int g_rows_slices;
bool rows_slice_plan(int M, int D, int F, int* nw_out, int* nslice_out, int* unfused_out = nullptr);
extern thread_local int g_rows_slices_fuse;
hipError_t launch_ffn_rows_sliced(const float* X1, const LayerWeights& w, float* P, float* Y, int M, int D, int F, int nw,
                                  int nslice, hipStream_t s);
size_t rows_slice_floats(int M, int D, int nslice);
hipError_t launch_oproj_ffn_rows_sliced(const float* attn, const float* Rin, const LayerWeights& w, float* P, float* Y,
                                        int M, int D, int F, int nw, int nslice, hipStream_t s);
// Small M (the reference harness's batch 1): out-proj + LN1 + FFN + LN2 as two launches with F split over NS
// workgroups per 16-row tile (ffd_small.hip).  small_path_splits returns 0 when the large-M kernels should run.
int small_path_splits(int M, int D, int F);
size_t small_path_partial_floats(int M, int D, int NS);
hipError_t launch_oproj_ffn_small(const float* attn, const float* xres, const LayerWeights& w, float* x1, float* P,
                                  float* Y, int M, int D, int F, int NS, hipStream_t s);
// Mid-size M: the 64-row FFN main loop over F slices + the same reduce (ffd_small.hip); 0 = not this form
int mid_path_splits(int M, int D, int F);
hipError_t launch_ffn_mid(const float* x1, const LayerWeights& w, float* P, float* Y, int M, int D, int F, int NS, hipStream_t s);
extern thread_local int g_mid_path;
extern thread_local int g_small_path;
extern thread_local int g_small_wgs;
// Opt-in bf16x3-split FFN (ffd_tune "ffn_split"; ffd_ffn_split.hip)
extern thread_local int g_ffn_split;
extern thread_local int g_embed_threads;
extern thread_local int g_embed_ldsx;
bool ffn_split_supported(int D, int F);
size_t w1split_bytes(int D, int F);
size_t w2split_bytes(int D, int F);
hipError_t launch_pack_ffn_split(const float* W1, const float* W2, void* w1s, void* w2s, int D, int F, hipStream_t s);
hipError_t launch_ffn_ln_split(const float* X, const LayerWeights& w, float* Y, int M, int D, int F, hipStream_t s,
                               unsigned long long* stamp = nullptr);
extern thread_local int g_ffn_mb_override;
extern thread_local int g_ffn_rem;
extern thread_local int g_ffn_persist;
extern thread_local int g_attn_fused;
// fused in-projection + attention (ffd_qkvattn.hip)
size_t attn_pack_floats(int D, int H, int hpw, int q_only);  // q_only = pack mode: 0 q | k | v, 1 q only, 2 tile 0 = k | v, tile 1 = q
bool attn_kvq_supported(int hd);                              // head dims with a mode-2 pack (the split small-batch form's)
hipError_t launch_pack_attn(const float* in_w, const float* in_b, float* pack, int D, int H, int hpw, int q_only,
                            hipStream_t s);
bool qkv_attention_supported(int D, int hd);
int qkv_attention_hpw(int D, int hd, int L, int B);
int qkv_attention_small_split(int B, int H, int L);
extern thread_local int g_attn_small;
int num_cus();  // compute units of the current device (256 on MI355X); ffd_ffn.hip
extern thread_local int g_attn_hpw;
hipError_t launch_qkv_attention(const float* x, const float* awp, int hpw, int q_only, const float* kt,
                                const float* vt, float* kt_out, float* vt_out, float* out, int B, int L, int D, int hd,
                                int n_own, hipStream_t s, unsigned long long* stamp = nullptr);
extern thread_local int g_attn_qg;

// Head-major projection: columns [r*d, (r+1)*d) of Y = X Wp^T + b go to region out[r]
// laid out (B, H, L, hd) -- each (sample, head) slice contiguous, the layout of the K/V tables.
hipError_t launch_linear_hm(const float* X, const float* Wp, const float* bias, float* out0, float* out1, float* out2,
                            int M, int nreg, int D, int L, int H, int hd, hipStream_t s);
// attention over head-major q, k, v (B,H,L,hd); tokens >= n_own take K/V from the
// (H,L,hd) tables kt/vt instead of the sample's own rows. out: (M x d) row-major.
hipError_t launch_attention(const float* q, const float* k, const float* v, const float* kt, const float* vt,
                            float* out, int B, int L, int H, int hd, int n_own, hipStream_t s);
// table[h][l][:] (l < n) <- sample 0's head-major K/V rows
hipError_t launch_kv_store(const float* k, const float* v, float* kt, float* vt, int L, int H, int hd, int n,
                           hipStream_t s);
// crf[l][:] <- h[l][:] for sample 0 is a plain D2D copy (done with hipMemcpyAsync)

hipError_t launch_lstm_layer(float* x, const float* gx, const float* whh, int B, int L, int D, hipStream_t s);
// batches below that kernel's crossover: all layers as a wavefront of (16-sample tile, layer) workgroups, in place on x;
// prog: >= 16 + 16 * ceil(B / 16) ints of device scratch (abort word + progress words, cleared by the launcher);
// err: host-visible word that receives 1 + (unit index) when a wait on a progress word runs out of time
bool lstm_wave_selected(int B, int D);
int lstm_wave_max_batch(int L, int D);  // samples one k_lstm_wave launch takes (a 16-sample tile per CU, rows < 2^31 bytes); larger batches go in sub-batches
extern thread_local int g_lstm_wave, g_lstm_wave_persist, g_lstm_wave_per, g_lstm_wave_chunk, g_lstm_wave_fault, g_lstm_wave_spin_ms;
hipError_t launch_lstm_wave(float* x, const float* const* wih_pk, const float* const* whh_pk, const float* const* bias_pk, int NL,
                            int B, int L, int D, int* prog, float* state, int* err, hipStream_t s,
                            unsigned long long* trace = nullptr);  // trace: 4 u64 per unit (ffd_lstm_trace), diagnostics
size_t lstm_wave_state_floats(int B, int D, int NL);
// launch_lstm_wave takes the weights in k_lstm_wave's fragment order (one pack per role and layer + the summed bias)
size_t lstm_wave_wpack_floats(int D);
size_t lstm_wave_bpack_floats(int D);
hipError_t launch_pack_lstm_wave(const float* wih, const float* whh, const float* bsum, float* ih_out, float* hh_out,
                                 float* b_out, int D, hipStream_t s);

hipError_t launch_dense(const float* X, const float* W, const float* b, const float* b2, const float* R, float* Y,
                        int M, int N, int K, int relu, hipStream_t s);
// a0 / a1 (both or neither; (L, C) on the device): forward out = (dft(in) - a0) / a1, inverse out = idft(in * a0 + a1)
hipError_t launch_dft(const float* in, float* out, int B, int L, int C, int inverse, const float* a0, const float* a1,
                      hipStream_t s);
hipError_t launch_freq_decompose(const float* in, float* low, float* high, int B, int L, int D, double low_freq_ratio,
                                 hipStream_t s);
hipError_t launch_spectral_density(const float* xf, float* out, int B, int L, int C, hipStream_t s);
hipError_t launch_weighted_sum(const float* hist, const float* w_host, float* out, int K, size_t n, hipStream_t s);
// FreSca spectral scaling of a (B,L,C) score; work: B*(L/2+1) + 1 floats; strategy 0 spatial, 1 energy
hipError_t launch_fresca(const float* in, float* out, float* work, int B, int L, int C, float low, float high,
                         double cutoff_ratio, int strategy, hipStream_t s);
// the 4-D (B, H, W, C) branch (rfft2 / irfft2 over H, W); work: fresca2d_work_floats floats
bool fresca2d_supported(int H, int W);
size_t fresca2d_work_floats(int B, int H, int W, int C);
hipError_t launch_fresca2d(const float* in, float* out, float* work, int B, int H, int W, int C, float low, float high,
                           double cutoff_ratio, int strategy, hipStream_t s);

}  // namespace ffd
