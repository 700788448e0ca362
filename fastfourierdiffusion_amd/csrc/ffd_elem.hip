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
// embed: h[row][j] = be[j] + sum_c X[row][c] We[j][c] (+ pos[l][j]) + temb[b][j]
// ---------------------------------------------------------------------------
// Write-bound (4 (C + D) bytes per row, D >> C).  k_embed_reg (C <= 8): a thread owns ONE (position l, float4 column
// j) of the output and walks the batch: its 4 x C embedder weights, bias, POSITIONAL float4 and (shared) time
// embedding stay in registers, so the loop body is the row's x (C floats, float4 loads when C % 4 == 0; the D/4
// threads of a row share them through L1), 4 C FMAs and one float4 store -- consecutive threads are consecutive
// float4 of consecutive rows, every wave writes one contiguous 1 KiB run.  No LDS, no table reads, no integer
// division in the loop; four samples per iteration keep four rows of loads in flight.
//
// LDSX: the D/4 threads of a row all need the same C floats of x.  As per-lane loads that is two 64-lane float4
// load instructions per stored KiB -- twice the address-path work of the store itself, and the kernel ran at 2.9 TB/s
// where its stores alone reach 6.1 (tools/probes/embed_sweep.py with the loads removed).  Instead the wave fetches the
// (<= 64) contiguous floats of x its 64 lanes' rows need with ONE dword load, parks them in LDS and every lane reads
// its row from there.  Slices are padded to whole waves so that the cooperating lanes share (slice, first row).
template <bool XVEC, bool LDSX, bool TS>
__global__ __launch_bounds__(256) void k_embed_reg(const float* __restrict__ X, const float* __restrict__ We,
                                                   const float* __restrict__ be, const float* __restrict__ pos,
                                                   const float* __restrict__ temb, int temb_stride,
                                                   float* __restrict__ h, int B, int L, int C, int D, int nslices) {
  __shared__ __align__(16) float xsh[LDSX ? 4 * 8 * 64 : 4];  // [wave][2 groups x 4 samples][float of the wave's x rows]
  const unsigned D4 = (unsigned)D >> 2;
  const unsigned LJ = (unsigned)L * D4;
  const unsigned LJP = LDSX ? (LJ + 63u) & ~63u : LJ;  // threads per slice
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned slice = g / LJP;
  if (slice >= (unsigned)nslices) return;
  const unsigned ljr = g - slice * LJP;
  const bool active = ljr < LJ;                    // (LDSX: the pad lanes of a slice's last wave help load, never store)
  const unsigned lj = active ? ljr : LJ - 1;
  const unsigned l = lj / D4;
  const int j = (int)(lj - l * D4) << 2;
  const unsigned lane = threadIdx.x & 63u;
  const unsigned r0 = (ljr - lane) / D4;           // first row of this wave
  const unsigned rl = min(ljr - lane + 63u, LJ - 1) / D4;
  const unsigned nf = (rl - r0 + 1) * (unsigned)C;  // floats of x the wave needs per sample (<= 64, launcher)
  const unsigned rel = (l - r0) * (unsigned)C;
  float* xw = xsh + (LDSX ? (threadIdx.x >> 6) * 512 : 0);
  float4 w[8];
  if (XVEC) {  // C = 4 or 8: the four weight rows j .. j+3 as float4 loads (8 instead of 32 per thread)
    float wr[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(We + (size_t)(j + k) * C);
      const float4 a2 = C > 4 ? *reinterpret_cast<const float4*>(We + (size_t)(j + k) * C + 4) : float4{0.f, 0.f, 0.f, 0.f};
      wr[k][0] = a.x, wr[k][1] = a.y, wr[k][2] = a.z, wr[k][3] = a.w;
      wr[k][4] = a2.x, wr[k][5] = a2.y, wr[k][6] = a2.z, wr[k][7] = a2.w;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) w[c] = float4{wr[0][c], wr[1][c], wr[2][c], wr[3][c]};
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c)
      w[c] = c < C ? float4{We[(j + 0) * C + c], We[(j + 1) * C + c], We[(j + 2) * C + c], We[(j + 3) * C + c]}
                   : float4{0.f, 0.f, 0.f, 0.f};
  }
  const float4 bias = *reinterpret_cast<const float4*>(be + j);
  const float4 p = pos ? *reinterpret_cast<const float4*>(pos + (size_t)l * D + j) : float4{0.f, 0.f, 0.f, 0.f};
  const float4 t0 = *reinterpret_cast<const float4*>(temb + j);  // the shared time embedding (temb_stride == 0)
  auto fetch_x = [&](int b, int u) {  // LDSX: the wave's x floats of sample b -> LDS slot u
    if (lane < nf) xw[u * 64 + lane] = X[((size_t)b * L + r0) * C + lane];
  };
  auto load_x = [&](int b, int u, float (&xv)[8]) {
    const float* x = LDSX ? xw + u * 64 + rel : X + ((size_t)b * L + l) * C;
    if (XVEC) {
      const float4 a = *reinterpret_cast<const float4*>(x);
      xv[0] = a.x, xv[1] = a.y, xv[2] = a.z, xv[3] = a.w;
      if (C > 4) {
        const float4 a2 = *reinterpret_cast<const float4*>(x + 4);
        xv[4] = a2.x, xv[5] = a2.y, xv[6] = a2.z, xv[7] = a2.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) xv[c] = c < C ? x[c] : 0.f;
    }
  };
  auto emit = [&](int b, const float (&xv)[8]) {
    float4 v = bias;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < C) v.x = fmaf(xv[c], w[c].x, v.x), v.y = fmaf(xv[c], w[c].y, v.y), v.z = fmaf(xv[c], w[c].z, v.z), v.w = fmaf(xv[c], w[c].w, v.w);
    if (pos) v.x += p.x, v.y += p.y, v.z += p.z, v.w += p.w;
    // (TS at compile time: as `temb_stride ? *p : t0` hipcc selected between the two ADDRESSES -- t0 parked in scratch,
    //  32 B per lane, and a flat load per stored float4 even where the batch shares one time embedding)
    float4 t = t0;
    if constexpr (TS) t = *reinterpret_cast<const float4*>(temb + (size_t)b * temb_stride + j);
    if (active) *reinterpret_cast<float4*>(h + ((size_t)b * L + l) * D + j) = float4{v.x + t.x, v.y + t.y, v.z + t.z, v.w + t.w};
  };
  int b = (int)slice;
  int grp = 0;  // LDSX: which half of the wave's LDS slots holds the current group of four samples
  if (LDSX && b + 3 * nslices < B) {
#pragma unroll
    for (int u = 0; u < 4; ++u) fetch_x(b + u * nslices, u);
  }
  for (; b + 3 * nslices < B; b += 4 * nslices) {
    float xa[4][8];
    if (LDSX) {  // the next group's x is requested before this group is consumed (the loop is latency-bound otherwise)
      if (b + 7 * nslices < B) {
#pragma unroll
        for (int u = 0; u < 4; ++u) fetch_x(b + (4 + u) * nslices, 4 * (grp ^ 1) + u);
      }
      __builtin_amdgcn_wave_barrier();  // (one wave, in-order LDS: only the compiler must keep the order)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) load_x(b + u * nslices, 4 * grp + u, xa[u]);
    if (LDSX) __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 4; ++u) emit(b + u * nslices, xa[u]);
    grp ^= 1;
  }
  for (; b < B; b += nslices) {
    float xa[8];
    if (LDSX) {
      fetch_x(b, 0);
      __builtin_amdgcn_wave_barrier();
    }
    load_x(b, 0, xa);
    if (LDSX) __builtin_amdgcn_wave_barrier();
    emit(b, xa);
  }
}

__global__ void k_embed(const float* __restrict__ X, const float* __restrict__ We, const float* __restrict__ be,
                        const float* __restrict__ pos, const float* __restrict__ temb, int temb_stride,
                        float* __restrict__ h, unsigned total4, int L, int C, int D) {
  // generic C: the (D x C) embedder weight is staged transposed in LDS ([c][j]); one float4 of h per thread and
  // iteration (D % 4 == 0); same operation order as k_embed_reg.
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

thread_local int g_embed_ldsx = 1;  // ffd_tune "embed_ldsx": 0 = every lane loads its row's x itself
thread_local int g_embed_threads = 262144;  // ffd_tune "embed_threads": threads the embed grid aims at (tools/probes/embed_sweep.py: 114 us at 256 k, 118 at 512 k, 137 at 128 k on the config-5 shape)

hipError_t launch_embed(const float* X, const float* We, const float* be, const float* pos, const float* temb,
                        int temb_stride, float* h, int B, int L, int C, int D, hipStream_t s) {
  const unsigned M = (unsigned)B * (unsigned)L;
  const unsigned total4 = (unsigned)((size_t)M * D / 4);
  if (C <= 8 && D % 4 == 0) {
    // one thread per (l, float4 column) and batch slice: as many slices as give ~512 k threads
    const unsigned D4 = (unsigned)(D / 4);
    const unsigned LJ = (unsigned)L * D4;
    // x through LDS when the rows of 64 consecutive lanes (64 / D4, + 2 for a wave that starts and ends mid-row)
    // need at most 64 floats of x
    // (only where it replaces float4 loads, C = 4 or 8: with one float per row the detour costs more than it saves)
    const bool ldsx = g_embed_ldsx && (C == 4 || C == 8) && (64u / D4 + 2u) * (unsigned)C <= 64u;
    const unsigned LJP = ldsx ? (LJ + 63u) & ~63u : LJ;
    int nslices = (int)((unsigned)g_embed_threads / LJP);
    if (nslices < 1) nslices = 1;
    if (nslices > B) nslices = B;
    const unsigned blocks = (unsigned)(((size_t)nslices * LJP + 255) / 256);
    const bool xvec = (C == 4 || C == 8) && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(We)) & 15) == 0;
#define FFD_EMBED(xv, lx)                                                                                              \
  do {                                                                                                                 \
    if (temb_stride)                                                                                                   \
      hipLaunchKernelGGL((k_embed_reg<xv, lx, true>), dim3(blocks), dim3(256), 0, s, X, We, be, pos, temb, temb_stride, h, B, L, C, D, nslices); \
    else                                                                                                               \
      hipLaunchKernelGGL((k_embed_reg<xv, lx, false>), dim3(blocks), dim3(256), 0, s, X, We, be, pos, temb, temb_stride, h, B, L, C, D, nslices); \
  } while (0)
    if (xvec && ldsx) FFD_EMBED(true, true);
    else if (xvec) FFD_EMBED(true, false);
    else if (ldsx) FFD_EMBED(false, true);
    else FFD_EMBED(false, false);
#undef FFD_EMBED
    return hipGetLastError();
  }
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
// Philox4x32-10 + Box-Muller
// ---------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: the generator is the
    // ALU floor of the noise-drawing kernels (k_prior writes 4 B per element and nothing else)
    const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    U4 n = {hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

// Two N(0,1) draws from two 32-bit words.  The hardware transcendentals (v_log_f32 = log2, v_sin_f32 / v_cos_f32 take
// their argument in revolutions, i.e. u2 itself) keep the generator off the critical path of the HBM-bound step
// kernel: the library logf / sincosf cost ~10x the instructions and made the step ALU-bound (2.8 TB/s).
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1)
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

// N(0,1) for global element index g at (seed, stream tag `step`): slot g&3 of Philox(counter g>>2).
__device__ __forceinline__ void normal4(uint64_t g4, uint64_t seed, uint32_t step, float out[4]) {
  U4 c = {(uint32_t)g4, (uint32_t)(g4 >> 32), step, 0x46464446u /* "FFDF" */};
  U4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  box_muller(r.x, r.y, out[0], out[1]);
  box_muller(r.z, r.w, out[2], out[3]);
}

// the (up to) 4 draws of elements i0 .. i0+3 (global index elem_offset + i0, any alignment), or injected ones
__device__ __forceinline__ void load_normals(const float* z, size_t i0, int n, uint64_t seed, uint64_t elem_offset,
                                             uint32_t step, float zz[4]) {
  if (z) {
    for (int j = 0; j < n; ++j) zz[j] = z[i0 + j];
    return;
  }
  uint64_t g0 = elem_offset + i0;
  float a[4], b[4] = {0.f, 0.f, 0.f, 0.f};
  normal4(g0 >> 2, seed, step, a);
  const int sh = (int)(g0 & 3);
  if (sh) normal4((g0 >> 2) + 1, seed, step, b);
  // zz[j] = (a | b)[sh + j] as selects on static indices (indexed by sh the two arrays lived in scratch: 48 B per lane)
  const float c[7] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2]};
#pragma unroll
  for (int j = 0; j < 4; ++j) zz[j] = sh == 0 ? c[j] : sh == 1 ? c[j + 1] : sh == 2 ? c[j + 2] : c[j + 3];
}

// ---------------------------------------------------------------------------
// reverse SDE step (sde.py:129-165, 215-246), elementwise form of the reference's
// diag(L x L) matmuls.  No FMA contraction: each product / sum rounds like the
// reference's separate torch ops.
//   VP: drift = a*x - (g*g)*s ;  x' = (x - drift*dt) + sqdt*(g*z),  g = cs*G[l]
//   VE: drift = -((g*g)*s)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sde_update(float xi, float sc, float Gl, float zz, const SdeParams& p) {
  const float g = __fmul_rn(p.cs, Gl);
  const float g2 = __fmul_rn(g, g);
  const float gs = __fmul_rn(g2, sc);
  const float drift = (p.sde == 0) ? __fsub_rn(__fmul_rn(p.a, xi), gs) : -gs;
  const float t1 = __fsub_rn(xi, __fmul_rn(drift, p.dt));
  return __fadd_rn(t1, __fmul_rn(p.sqdt, __fmul_rn(g, zz)));
}

// C % 4 == 0, 16-byte aligned buffers: one float4 of x / score (/ z) per thread and iteration; the four elements
// share their row, so G[l] is one load; the draws of an aligned quad are one Philox block.  12 B per element.
__global__ __launch_bounds__(256) void k_sde_step_v4(float* __restrict__ x, const float* __restrict__ score,
                                                     const float* __restrict__ z, const float* __restrict__ G,
                                                     SdeParams p, uint64_t seed, uint64_t elem_offset, uint32_t step,
                                                     size_t nvec, int L, unsigned C4) {
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    const float4 xi = reinterpret_cast<const float4*>(x)[v];
    const float4 sc = reinterpret_cast<const float4*>(score)[v];
    float zz[4];
    if (z) {
      const float4 zv = reinterpret_cast<const float4*>(z)[v];
      zz[0] = zv.x, zz[1] = zv.y, zz[2] = zv.z, zz[3] = zv.w;
    } else {
      normal4((elem_offset >> 2) + v, seed, step, zz);  // elem_offset is a multiple of C
    }
    const float Gl = G[(v / C4) % (size_t)L];
    reinterpret_cast<float4*>(x)[v] = float4{sde_update(xi.x, sc.x, Gl, zz[0], p), sde_update(xi.y, sc.y, Gl, zz[1], p),
                                             sde_update(xi.z, sc.z, Gl, zz[2], p), sde_update(xi.w, sc.w, Gl, zz[3], p)};
  }
}

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
      x[i] = sde_update(x[i], score[i], G[l], zz[j], p);
    }
  }
}

static bool ptr16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

hipError_t launch_sde_step(float* x, const float* score, const float* z, const float* G, SdeParams p, uint64_t seed,
                           uint64_t elem_offset, uint32_t step, int B, int L, int C, hipStream_t s) {
  size_t total = (size_t)B * L * C;
  size_t blocks = ((total + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (C % 4 == 0 && ptr16(x) && ptr16(score) && ptr16(z) && elem_offset % 4 == 0)
    hipLaunchKernelGGL(k_sde_step_v4, dim3((unsigned)blocks), dim3(256), 0, s, x, score, z, G, p, seed, elem_offset, step,
                       total / 4, L, (unsigned)C / 4);
  else
    hipLaunchKernelGGL(k_sde_step, dim3((unsigned)blocks), dim3(256), 0, s, x, score, z, G, p, seed, elem_offset, step,
                       total, L, C);
  return hipGetLastError();
}

// prior: x = G (.) z  (VE: * sigma_max), sde.py:79-87,125-127
__device__ __forceinline__ float prior_value(float Gl, float zz, float scale) {
  const float v0 = __fmul_rn(Gl, zz);
  return (scale == 1.0f) ? v0 : __fmul_rn(scale, v0);
}

__global__ __launch_bounds__(256) void k_prior_v4(float* __restrict__ x, const float* __restrict__ z,
                                                  const float* __restrict__ G, float scale, uint64_t seed,
                                                  uint64_t elem_offset, size_t nvec, int L, unsigned C4) {
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    float zz[4];
    if (z) {
      const float4 zv = reinterpret_cast<const float4*>(z)[v];
      zz[0] = zv.x, zz[1] = zv.y, zz[2] = zv.z, zz[3] = zv.w;
    } else {
      normal4((elem_offset >> 2) + v, seed, 0xFFFFFFFFu, zz);
    }
    const float Gl = G[(v / C4) % (size_t)L];
    reinterpret_cast<float4*>(x)[v] = float4{prior_value(Gl, zz[0], scale), prior_value(Gl, zz[1], scale),
                                             prior_value(Gl, zz[2], scale), prior_value(Gl, zz[3], scale)};
  }
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
      x[i] = prior_value(G[l], zz[j], scale);
    }
  }
}

hipError_t launch_prior(float* x, const float* z, const float* G, float scale, uint64_t seed, uint64_t elem_offset,
                        int B, int L, int C, hipStream_t s) {
  size_t total = (size_t)B * L * C;
  size_t blocks = ((total + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (C % 4 == 0 && ptr16(x) && ptr16(z) && elem_offset % 4 == 0)
    hipLaunchKernelGGL(k_prior_v4, dim3((unsigned)blocks), dim3(256), 0, s, x, z, G, scale, seed, elem_offset, total / 4, L,
                       (unsigned)C / 4);
  else
    hipLaunchKernelGGL(k_prior, dim3((unsigned)blocks), dim3(256), 0, s, x, z, G, scale, seed, elem_offset, total, L, C);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// unembed: score[row][c] = bu[c] + h[row][:] . Wu[c][:]     (score_models.py:113)
// and, fused, the reverse SDE step on that score (the score never reaches HBM).
// ---------------------------------------------------------------------------
// HBM-bound on the read of h (4 D bytes per row against 4 C written).  One wave owns 16 rows per iteration and forms
// score^T (C x 16) on the exact-fp32 matrix core: A = the unembedder weight (channel on the row index, rows >= C
// zero), B = the h rows, each lane loading FOUR consecutive k of its row as one float4 per 16-wide k chunk (the
// MFMA's k index is then a permutation of the chunk's 16 k values, the same for A and B; the qkv kernel's trick).
// The accumulator leaves lane l with channels 4 (l >> 4) .. +3 of row (l & 15): exactly one float4 of score / x
// when C % 4 == 0.  SDE = true continues with the Euler-Maruyama update of x in those registers (same sde_update,
// same Philox indexing as k_sde_step); SDE = false stores the score.  The next tile's h is prefetched under the
// current tile's MFMAs and epilogue.
template <int D, bool SDE>
__global__ __launch_bounds__(256) void k_unembed_mfma(const float* __restrict__ h, const float* __restrict__ Wu,
                                                      const float* __restrict__ bu, float* __restrict__ score,
                                                      float* __restrict__ x, const float* __restrict__ z,
                                                      const float* __restrict__ G, SdeParams p, uint64_t seed,
                                                      uint64_t elem_offset, uint32_t step, int M, int L, int C) {
  constexpr int NG = (D + 15) / 16;  // 16-wide k chunks
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int ntiles = (M + 15) >> 4;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  float4 wf[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int k = 16 * g + 4 * q;
    wf[g] = (r < C && k < D) ? *reinterpret_cast<const float4*>(Wu + (size_t)r * D + k) : float4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 bias;
#pragma unroll
  for (int i = 0; i < 4; ++i) bias[i] = (4 * q + i < C) ? bu[4 * q + i] : 0.f;
  auto load_tile = [&](int t, float4 (&hv)[NG]) {
    const int row = min(16 * t + r, M - 1);
    const float* hr = h + (size_t)row * D + 4 * q;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      hv[g] = (16 * g + 4 * q < D) ? *reinterpret_cast<const float4*>(hr + 16 * g) : float4{0.f, 0.f, 0.f, 0.f};
  };
  // (SDE, C % 4 == 0) the tile's x / injected z quad is fetched together with its h rows
  const bool quad = SDE && (C & 3) == 0 && 4 * q < C;
  auto load_xz = [&](int t, float4& xv, float4& zv) {
    if (quad) {
      const size_t i0 = (size_t)min(16 * t + r, M - 1) * C + 4 * q;
      xv = *reinterpret_cast<const float4*>(x + i0);
      if (z) zv = *reinterpret_cast<const float4*>(z + i0);
    }
  };
  float4 hv[NG], hn[NG];
  float4 xv{}, zv{}, xnx{}, znx{};
  if (wave < ntiles) load_tile(wave, hv), load_xz(wave, xv, zv);
  for (int t = wave; t < ntiles; t += nwaves) {
    const bool more = t + nwaves < ntiles;
    if (more) load_tile(t + nwaves, hn), load_xz(t + nwaves, xnx, znx);
    f32x4 acc = bias;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      acc = mfma16(wf[g].x, hv[g].x, acc);
      acc = mfma16(wf[g].y, hv[g].y, acc);
      acc = mfma16(wf[g].z, hv[g].z, acc);
      acc = mfma16(wf[g].w, hv[g].w, acc);
    }
    const int row = 16 * t + r;
    const int c0 = 4 * q;
    if (row < M && c0 < C) {
      const size_t i0 = (size_t)row * C + c0;
      const int n = min(4, C - c0);
      if (!SDE) {
        if ((C & 3) == 0) *reinterpret_cast<float4*>(score + i0) = float4{acc[0], acc[1], acc[2], acc[3]};
        else
          for (int i = 0; i < n; ++i) score[i0 + i] = acc[i];
      } else {
        const float Gl = G[row % L];
        float zz[4];
        if ((C & 3) == 0) {
          const float4 xi = xv;
          if (z) {
            zz[0] = zv.x, zz[1] = zv.y, zz[2] = zv.z, zz[3] = zv.w;
          } else {
            normal4((elem_offset + i0) >> 2, seed, step, zz);
          }
          *reinterpret_cast<float4*>(x + i0) = float4{sde_update(xi.x, acc[0], Gl, zz[0], p), sde_update(xi.y, acc[1], Gl, zz[1], p),
                                                      sde_update(xi.z, acc[2], Gl, zz[2], p), sde_update(xi.w, acc[3], Gl, zz[3], p)};
        } else {
          load_normals(z, i0, n, seed, elem_offset, step, zz);
          for (int i = 0; i < n; ++i) x[i0 + i] = sde_update(x[i0 + i], acc[i], Gl, zz[i], p);
        }
      }
    }
    if (more) {
#pragma unroll
      for (int g = 0; g < NG; ++g) hv[g] = hn[g];
      xv = xnx, zv = znx;
    }
  }
}

// generic fallback (C > 16): 16 lanes per row, weight in LDS
__global__ void k_unembed(const float* __restrict__ h, const float* __restrict__ Wu, const float* __restrict__ bu,
                          float* __restrict__ score, int M, int C, int D) {
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

bool unembed_sde_supported(int C, int D) {
  if (C > 16 || D % 4 != 0) return false;
#define X(v) if (D == v) return true;
  FFD_D_LIST(X)
#undef X
  return false;
}

template <bool SDE>
static hipError_t launch_unembed_mfma(const float* h, const float* Wu, const float* bu, float* score, float* x,
                                      const float* z, const float* G, SdeParams p, uint64_t seed, uint64_t elem_offset,
                                      uint32_t step, int M, int L, int C, int D, hipStream_t s) {
  int blocks = cdiv(cdiv(M, 16), 4);
  if (blocks > 2048) blocks = 2048;
  switch (D) {
#define X(d)                                                                                                     \
  case d:                                                                                                        \
    hipLaunchKernelGGL((k_unembed_mfma<d, SDE>), dim3(blocks), dim3(256), 0, s, h, Wu, bu, score, x, z, G, p, seed, \
                       elem_offset, step, M, L, C);                                                              \
    break;
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_unembed(const float* h, const float* Wu, const float* bu, float* score, int M, int C, int D,
                          hipStream_t s) {
  if (unembed_sde_supported(C, D) && ptr16(h) && ptr16(Wu) && ptr16(score))
    return launch_unembed_mfma<false>(h, Wu, bu, score, nullptr, nullptr, nullptr, SdeParams{}, 0, 0, 0, M, 1, C, D, s);
  if (D > 128 || (size_t)C * D * sizeof(float) > 64 * 1024) return hipErrorInvalidValue;
  int blocks = cdiv(M, 16);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_unembed, dim3(blocks), dim3(256), (size_t)C * D * sizeof(float), s, h, Wu, bu, score, M, C, D);
  return hipGetLastError();
}

// x <- SDE step(x, unembed(h)): the fused tail of a sampling step (requires unembed_sde_supported(C, D))
hipError_t launch_unembed_sde(const float* h, const float* Wu, const float* bu, float* x, const float* z, const float* G,
                              SdeParams p, uint64_t seed, uint64_t elem_offset, uint32_t step, int B, int L, int C,
                              int D, hipStream_t s) {
  if (!unembed_sde_supported(C, D) || !ptr16(h) || !ptr16(Wu)) return hipErrorInvalidValue;
  if ((C & 3) == 0 && (!ptr16(x) || !ptr16(z))) return hipErrorInvalidValue;  // x / z quads
  if ((C & 3) == 0 && (elem_offset & 3)) return hipErrorInvalidValue;
  return launch_unembed_mfma<true>(h, Wu, bu, nullptr, x, z, G, p, seed, elem_offset, step, B * L, L, C, D, s);
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

}  // namespace ffd
