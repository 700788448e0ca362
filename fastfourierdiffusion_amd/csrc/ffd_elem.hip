// Element-wise / small kernels of the sampling path (gfx950).
//   - weight packing into MFMA fragment order, nn.Embedding(max_norm) renorm
//   - Gaussian-Fourier time embedding, channel embed / unembed
//   - VP / VE reverse Euler-Maruyama step with on-device Philox4x32-10 noise
//   - KV-table store
// All HBM-bound: one pass over the data, coalesced, no re-reads.
#include <algorithm>

#include "ffd_internal.h"

namespace ffd {

// ---------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------
__global__ void k_pack_dweight(const float* __restrict__ W, float* __restrict__ Wp, int N, int D) {
  const int G = dpack_groups(D);
  const size_t total = dpack_floats(N, D);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int j = i & 3;
    int lane = (i >> 2) & 63;
    size_t rest = i >> 8;
    int g = rest % G;
    int nt = rest / G;
    int n = 16 * nt + (lane & 15);
    int k = 4 * (4 * g + j) + (lane >> 4);
    Wp[i] = (n < N && k < D) ? W[(size_t)n * D + k] : 0.f;
  }
}

__global__ void k_pack_w2(const float* __restrict__ W2, float* __restrict__ W2p, int D, int F) {
  const int CT = cdiv(D, 16);
  const size_t total = w2pack_floats(D, F);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int r = i & 3;
    int lane = (i >> 2) & 63;
    size_t rest = i >> 8;
    int ct = rest % CT;
    int fc = rest / CT;
    int c = 16 * ct + (lane & 15);
    int f = 16 * fc + 4 * (lane >> 4) + r;
    W2p[i] = (c < D) ? W2[(size_t)c * F + f] : 0.f;
  }
}

__global__ void k_pack_w2rem(const float* __restrict__ W2, float* __restrict__ W2r, int D, int F) {
  const int NG = w2rem_groups(D);
  const int c0 = 16 * (D / 16);
  const size_t total = (size_t)(F / 16) * NG * 64 * 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int r = i & 3;
    int lane = (i >> 2) & 63;
    size_t rest = i >> 8;
    int g = rest % NG;
    int fc = rest / NG;
    int c = c0 + 4 * g + (lane & 3);
    int f = 16 * fc + 4 * (lane >> 4) + r;
    W2r[i] = W2[(size_t)c * F + f];
  }
}

hipError_t launch_pack_w2rem(const float* W2, float* W2r, int D, int F, hipStream_t s) {
  if (w2rem_groups(D) == 0) return hipSuccess;
  size_t total = (size_t)(F / 16) * w2rem_groups(D) * 64 * 4;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_pack_w2rem, dim3(blocks), dim3(256), 0, s, W2, W2r, D, F);
  return hipGetLastError();
}

hipError_t launch_pack_dweight(const float* W, float* Wp, int N, int D, hipStream_t s) {
  size_t total = dpack_floats(N, D);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_pack_dweight, dim3(blocks), dim3(256), 0, s, W, Wp, N, D);
  return hipGetLastError();
}

hipError_t launch_pack_w2(const float* W2, float* W2p, int D, int F, hipStream_t s) {
  size_t total = w2pack_floats(D, F);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_pack_w2, dim3(blocks), dim3(256), 0, s, W2, W2p, D, F);
  return hipGetLastError();
}

// nn.Embedding(max_norm): rows with ||w|| > max_norm scaled by max_norm/(||w||+1e-7),
// iterated to the fixed point the reference reaches after a few lookups.
__global__ void k_renorm_rows(float* __restrict__ W, int D, float max_norm) {
  float* w = W + (size_t)blockIdx.x * D;
  __shared__ float red[WAVE];
  for (int it = 0; it < 8; ++it) {
    float ss = 0.f;
    for (int k = threadIdx.x; k < D; k += WAVE) ss += w[k] * w[k];
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    float norm = sqrtf(ss);
    if (!(norm > max_norm)) break;  // wave-uniform
    float scale = max_norm / (norm + 1e-7f);
    for (int k = threadIdx.x; k < D; k += WAVE) w[k] *= scale;
    __syncthreads();
  }
  (void)red;
}

hipError_t launch_renorm_rows(float* W, int rows, int D, float max_norm, hipStream_t s) {
  hipLaunchKernelGGL(k_renorm_rows, dim3(rows), dim3(WAVE), 0, s, W, D, max_norm);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// time embedding: temb[n][:] = dense([sin(2 pi t W), cos(2 pi t W)][:d])
// ---------------------------------------------------------------------------
__global__ void k_time_embed(const float* __restrict__ ts, float t_imm, const float* __restrict__ W,
                             const float* __restrict__ dw, const float* __restrict__ db, float* __restrict__ temb,
                             int D) {
  extern __shared__ float emb[];
  const float t = ts ? ts[blockIdx.x] : t_imm;
  const int half = (D + 1) / 2;
  for (int j = threadIdx.x; j < D; j += blockDim.x) {
    // ((t * W) * 2) * pi, each product rounded to fp32 (transformer.py:80)
    float w = W[j < half ? j : j - half];
    float proj = __fmul_rn(__fmul_rn(__fmul_rn(t, w), 2.0f), 3.14159265358979323846f);
    emb[j] = (j < half) ? sinf(proj) : cosf(proj);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < D; j += blockDim.x) {
    float acc = db[j];
    const float* row = dw + (size_t)j * D;
    for (int k = 0; k < D; ++k) acc = fmaf(emb[k], row[k], acc);
    temb[(size_t)blockIdx.x * D + j] = acc;
  }
}

hipError_t launch_time_embed(const float* ts, float t_imm, int n, const float* W, const float* dense_w,
                             const float* dense_b, float* temb, int D, hipStream_t s) {
  hipLaunchKernelGGL(k_time_embed, dim3(n), dim3(128), D * sizeof(float), s, ts, t_imm, W, dense_w, dense_b, temb, D);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// embed: h[row][j] = be[j] + sum_c X[row][c] We[j][c] (+ pos[l][j]) + temb[j]
// ---------------------------------------------------------------------------
__global__ void k_embed(const float* __restrict__ X, const float* __restrict__ We, const float* __restrict__ be,
                        const float* __restrict__ pos, const float* __restrict__ temb, int temb_stride,
                        float* __restrict__ h, unsigned total4, int L, int C, int D) {
  // The (D x C) embedder weight is staged transposed in LDS ([c][j]), so the C weights of an output float4 are C
  // aligned 16-byte LDS reads instead of 4 C scattered global loads.  One float4 of h per thread and iteration
  // (D % 4 == 0); 32-bit index math only; same operation order as before the staging.
  extern __shared__ __align__(16) float wt[];  // C * D floats
  for (int i = threadIdx.x; i < C * D; i += blockDim.x) {
    const int c = i / D, j = i - c * D;
    wt[i] = We[j * C + c];
  }
  __syncthreads();
  for (unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += gridDim.x * blockDim.x) {
    const unsigned row = (4u * i4) / (unsigned)D;
    const int j = (int)(4u * i4 - row * (unsigned)D);
    const unsigned bidx = row / (unsigned)L;
    const int l = (int)(row - bidx * (unsigned)L);
    const float* x = X + (size_t)row * C;
    float4 v = float4{be[j], be[j + 1], be[j + 2], be[j + 3]};
    for (int c = 0; c < C; ++c) {
      const float xc = x[c];
      const float4 w4 = *reinterpret_cast<const float4*>(&wt[c * D + j]);
      v.x = fmaf(xc, w4.x, v.x), v.y = fmaf(xc, w4.y, v.y), v.z = fmaf(xc, w4.z, v.z), v.w = fmaf(xc, w4.w, v.w);
    }
    if (pos) {
      const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)l * D + j);
      v.x += p.x, v.y += p.y, v.z += p.z, v.w += p.w;
    }
    const float4 t = *reinterpret_cast<const float4*>(temb + (size_t)bidx * temb_stride + j);
    reinterpret_cast<float4*>(h)[i4] = float4{v.x + t.x, v.y + t.y, v.z + t.z, v.w + t.w};
  }
}

hipError_t launch_embed(const float* X, const float* We, const float* be, const float* pos, const float* temb,
                        int temb_stride, float* h, int B, int L, int C, int D, hipStream_t s) {
  const unsigned total4 = (unsigned)((size_t)B * L * D / 4);
  unsigned blocks = (total4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;  // grid-stride: the weight staging is amortised over >= a few rows per thread
  const size_t lds = (size_t)C * D * sizeof(float);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_embed, dim3(blocks), dim3(256), lds, s, X, We, be, pos, temb, temb_stride, h, total4, L, C, D);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// standalone encoders: out[b,l,:] = x[b,l,:] + rowtab[l,:] (positional) or + battab[b,:] (time)
// ---------------------------------------------------------------------------
__global__ void k_add_table(const float* __restrict__ x, const float* __restrict__ rowtab,
                            const float* __restrict__ battab, float* __restrict__ out, unsigned total, int L, int D) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned row = i / (unsigned)D;
    const int j = (int)(i - row * (unsigned)D);
    const unsigned b = row / (unsigned)L;
    const int l = (int)(row - b * (unsigned)L);
    float v = x[i];
    if (rowtab) v += rowtab[(size_t)l * D + j];
    if (battab) v += battab[(size_t)b * D + j];
    out[i] = v;
  }
}

hipError_t launch_add_table(const float* x, const float* rowtab, const float* battab, float* out, int B, int L, int D,
                            hipStream_t s) {
  const unsigned total = (unsigned)((size_t)B * L * D);
  unsigned blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_add_table, dim3(blocks), dim3(256), 0, s, x, rowtab, battab, out, total, L, D);
  return hipGetLastError();
}

// one nn.Embedding(max_norm) lookup-time renormalisation pass (torch embedding_renorm_)
__global__ void k_renorm_rows_once(float* __restrict__ W, int D, float max_norm) {
  float* w = W + (size_t)blockIdx.x * D;
  float ss = 0.f;
  for (int k = threadIdx.x; k < D; k += WAVE) ss += w[k] * w[k];
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float norm = sqrtf(ss);
  if (norm > max_norm) {
    const float scale = max_norm / (norm + 1e-7f);
    for (int k = threadIdx.x; k < D; k += WAVE) w[k] *= scale;
  }
}

hipError_t launch_renorm_rows_once(float* W, int rows, int D, float max_norm, hipStream_t s) {
  hipLaunchKernelGGL(k_renorm_rows_once, dim3(rows), dim3(WAVE), 0, s, W, D, max_norm);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// unembed: score[row][c] = bu[c] + h[row][:] . Wu[c][:]   (16 lanes per row)
// ---------------------------------------------------------------------------
__global__ void k_unembed(const float* __restrict__ h, const float* __restrict__ Wu, const float* __restrict__ bu,
                          float* __restrict__ score, int M, int C, int D) {
  // The row's D values are read once into registers (lane `sub` of the row's 16 holds k = sub, sub+16, ...), the
  // (C x D) weight sits in LDS; per channel: <= 8 FMAs and a 16-lane xor reduction.  D <= 128.
  extern __shared__ float wl[];  // C * D
  for (int i = threadIdx.x; i < C * D; i += blockDim.x) wl[i] = Wu[i];
  __syncthreads();
  constexpr int NI = 8;
  const int sub = threadIdx.x & 15;
  const int rows_per_block = blockDim.x >> 4;
  for (int row = blockIdx.x * rows_per_block + (threadIdx.x >> 4); row < M; row += gridDim.x * rows_per_block) {
    const float* hr = h + (size_t)row * D;
    float hv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) hv[i] = (sub + 16 * i < D) ? hr[sub + 16 * i] : 0.f;
    for (int c = 0; c < C; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i)
        if (16 * i < D) acc = fmaf(hv[i], (sub + 16 * i < D) ? wl[c * D + sub + 16 * i] : 0.f, acc);
      acc += __shfl_xor(acc, 8, 16);
      acc += __shfl_xor(acc, 4, 16);
      acc += __shfl_xor(acc, 2, 16);
      acc += __shfl_xor(acc, 1, 16);
      if (sub == 0) score[(size_t)row * C + c] = acc + bu[c];
    }
  }
}

hipError_t launch_unembed(const float* h, const float* Wu, const float* bu, float* score, int M, int C, int D,
                          hipStream_t s) {
  if (D > 128 || (size_t)C * D * sizeof(float) > 64 * 1024) return hipErrorInvalidValue;
  int blocks = cdiv(M, 16);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_unembed, dim3(blocks), dim3(256), (size_t)C * D * sizeof(float), s, h, Wu, bu, score, M, C, D);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller
// ---------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    U4 n = {hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
  float r = sqrtf(-2.0f * logf(u1));
  float s, c;
  sincosf(6.28318530717958647692f * u2, &s, &c);
  z0 = r * c;
  z1 = r * s;
}

// N(0,1) for global element index g at (seed, stream tag `step`): slot g&3 of Philox(counter g>>2).
__device__ __forceinline__ void normal4(uint64_t g4, uint64_t seed, uint32_t step, float out[4]) {
  U4 c = {(uint32_t)g4, (uint32_t)(g4 >> 32), step, 0x46464446u /* "FFDF" */};
  U4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  box_muller(r.x, r.y, out[0], out[1]);
  box_muller(r.z, r.w, out[2], out[3]);
}

__device__ __forceinline__ void load_normals(const float* z, size_t i0, int n, uint64_t seed, uint64_t elem_offset,
                                             uint32_t step, float zz[4]) {
  if (z) {
    for (int j = 0; j < n; ++j) zz[j] = z[i0 + j];
    return;
  }
  uint64_t g0 = elem_offset + i0;
  float a[4], b[4];
  normal4(g0 >> 2, seed, step, a);
  int sh = (int)(g0 & 3);
  if (sh) normal4((g0 >> 2) + 1, seed, step, b);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int q = sh + j;
    zz[j] = (q < 4) ? a[q & 3] : b[q & 3];
  }
}

// ---------------------------------------------------------------------------
// reverse SDE step (sde.py:129-165, 215-246), elementwise form of the reference's
// diag(L x L) matmuls.  No FMA contraction: each product / sum rounds like the
// reference's separate torch ops.
//   VP: drift = a*x - (g*g)*s ;  x' = (x - drift*dt) + sqdt*(g*z),  g = cs*G[l]
//   VE: drift = -((g*g)*s)
// ---------------------------------------------------------------------------
__global__ void k_sde_step(float* __restrict__ x, const float* __restrict__ score, const float* __restrict__ z,
                           const float* __restrict__ G, SdeParams p, uint64_t seed, uint64_t elem_offset,
                           uint32_t step, size_t total, int L, int C) {
  size_t nvec = (total + 3) / 4;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    size_t i0 = v * 4;
    int n = (int)((total - i0) < 4 ? (total - i0) : 4);
    float zz[4];
    load_normals(z, i0, n, seed, elem_offset, step, zz);
    for (int j = 0; j < n; ++j) {
      size_t i = i0 + j;
      const int l = (int)((i / (size_t)C) % (size_t)L);
      float g = __fmul_rn(p.cs, G[l]);
      float g2 = __fmul_rn(g, g);
      float xi = x[i];
      float gs = __fmul_rn(g2, score[i]);
      float drift = (p.sde == 0) ? __fsub_rn(__fmul_rn(p.a, xi), gs) : -gs;
      float t1 = __fsub_rn(xi, __fmul_rn(drift, p.dt));
      x[i] = __fadd_rn(t1, __fmul_rn(p.sqdt, __fmul_rn(g, zz[j])));
    }
  }
}

hipError_t launch_sde_step(float* x, const float* score, const float* z, const float* G, SdeParams p, uint64_t seed,
                           uint64_t elem_offset, uint32_t step, int B, int L, int C, hipStream_t s) {
  size_t total = (size_t)B * L * C;
  size_t blocks = ((total + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_sde_step, dim3((unsigned)blocks), dim3(256), 0, s, x, score, z, G, p, seed, elem_offset, step,
                     total, L, C);
  return hipGetLastError();
}

__global__ void k_prior(float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ G, float scale,
                        uint64_t seed, uint64_t elem_offset, size_t total, int L, int C) {
  size_t nvec = (total + 3) / 4;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    size_t i0 = v * 4;
    int n = (int)((total - i0) < 4 ? (total - i0) : 4);
    float zz[4];
    load_normals(z, i0, n, seed, elem_offset, 0xFFFFFFFFu, zz);
    for (int j = 0; j < n; ++j) {
      size_t i = i0 + j;
      const int l = (int)((i / (size_t)C) % (size_t)L);
      float v0 = __fmul_rn(G[l], zz[j]);
      x[i] = (scale == 1.0f) ? v0 : __fmul_rn(scale, v0);
    }
  }
}

hipError_t launch_prior(float* x, const float* z, const float* G, float scale, uint64_t seed, uint64_t elem_offset,
                        int B, int L, int C, hipStream_t s) {
  size_t total = (size_t)B * L * C;
  size_t blocks = ((total + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_prior, dim3((unsigned)blocks), dim3(256), 0, s, x, z, G, scale, seed, elem_offset, total, L, C);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// KV table store: table[h][l][:] <- sample 0's head-major K/V rows, l < n (Q1)
// ---------------------------------------------------------------------------
__global__ void k_kv_store(const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ kt,
                           float* __restrict__ vt, int L, int H, int hd, int n) {
  const int per_head = n * hd;
  const int total = H * per_head;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int h = i / per_head, r = i - h * per_head;
    const size_t off = (size_t)h * L * hd + r;  // sample 0: (0*H + h)*L*hd
    kt[off] = k[off];
    vt[off] = v[off];
  }
}

hipError_t launch_kv_store(const float* k, const float* v, float* kt, float* vt, int L, int H, int hd, int n,
                           hipStream_t s) {
  if (n <= 0) return hipSuccess;
  int blocks = cdiv(n * H * hd, 256);
  hipLaunchKernelGGL(k_kv_store, dim3(blocks), dim3(256), 0, s, k, v, kt, vt, L, H, hd, n);
  return hipGetLastError();
}

// predict_hermite's last step (fourier.py:470-495): prediction = sum_k w_k * history[k]; the K weights come
// from the (order+1)^2 normal equations solved on the host.
struct WeightVec { float w[32]; };
__global__ void k_weighted_sum(const float* __restrict__ hist, WeightVec wv, float* __restrict__ out, int K, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(wv.w[k], hist[(size_t)k * n + i], acc);
    out[i] = acc;
  }
}

hipError_t launch_weighted_sum(const float* hist, const float* w_host, float* out, int K, size_t n, hipStream_t s) {
  if (K < 1 || K > 32) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  WeightVec wv{};
  for (int k = 0; k < K; ++k) wv.w[k] = w_host[k];
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(k_weighted_sum, dim3(blocks), dim3(256), 0, s, hist, wv, out, K, n);
  return hipGetLastError();
}

// compute_event_intensity's reduction (caching.py:546-556): mean over rows of || a_r - b_r ||_2.
// One wave per row, per-block partial sums in a fixed order (deterministic); the host adds the <= 256 partials.
__global__ __launch_bounds__(256) void k_row_delta_norm(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ partial, int rows, int D) {
  __shared__ float wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    float ss = 0.f;
    for (int k = lane; k < D; k += 64) {
      const float dlt = fabsf(a[(size_t)r * D + k] - b[(size_t)r * D + k]);
      ss = fmaf(dlt, dlt, ss);
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    acc += sqrtf(ss);
  }
  if (lane == 0) wsum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

hipError_t launch_row_delta_norm(const float* a, const float* b, float* partial, int nblocks, int rows, int D,
                                 hipStream_t s) {
  hipLaunchKernelGGL(k_row_delta_norm, dim3(nblocks), dim3(256), 0, s, a, b, partial, rows, D);
  return hipGetLastError();
}

}  // namespace ffd
