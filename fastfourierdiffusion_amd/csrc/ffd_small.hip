// Small batches (the reference's own harness regime, cmd/benchmark_cache.py:71-73: sample_batch_size = 1): the part
// of an encoder layer after attention,
//     x1 = LN1(x + attn Wo^T + bo);   y = LN2(x1 + W2 relu(W1 x1 + b1) + b2)        (cached_transformer.py:316-327)
// as two launches that use the whole chip instead of M / 16 workgroups.
//
// At M = 187 rows (ECG, B = 1) the large-M kernels run 3 (out-proj) and 12 (FFN) workgroups, and an FFN workgroup
// walks all 2048 hidden units of its 16 rows as one dependent chain of 32 chunks x 38 MFMAs: 30 us of latency
// per layer, 60 % of the 0.49 ms step.
//   k_oproj_ffn_split : grid (row tiles, NS).  Workgroup (t, s) recomputes the out-projection + LN1 of ITS 16 rows
//                       (90 MFMAs: cheaper than a launch; split 0 also writes x1 for the residual), then runs the FFN
//                       over hidden units [s F / NS, (s + 1) F / NS) only and stores its partial Y tile.
//   k_ffn_reduce_ln   : grid (row tiles).  y = LN2(x1 + b2 + sum_s partial[t][s]) with the partials added in split
//                       order -- deterministic, no cross-workgroup synchronisation, no atomics.
// F split 16 ways: the chain per wave is 2 chunks (interleaved: two independent GEMM1 accumulators).
// Arithmetic as in k_ffn_ln except that the linear1 bias is added after the k-sum instead of before it.
#include "ffd_internal.h"

namespace ffd {

// CPW = chunks (16 hidden units) per wave = F / (64 NS), a template parameter: the chunk loop is straight-line code.
// (As a run-time loop with the next pair's fragments carried in registers, hipcc 7.2 miscompiled it twice -- VGPR ->
// AGPR copies of the loop-carried bias dropped, then loop-carried registers reused while live; see
// tools/check_mfma_operands.py and DESIGN.md.  Without a loop there is nothing loop-carried.)
template <int D, int CPW>
__global__ __launch_bounds__(256) void k_oproj_ffn_split(const float* __restrict__ A, const float* __restrict__ Wop,
                                                         const float* __restrict__ bo, const float* __restrict__ R,
                                                         const float* __restrict__ g1, const float* __restrict__ e1,
                                                         float* __restrict__ X1, const float* __restrict__ W1p,
                                                         const float* __restrict__ b1, const float* __restrict__ W2p,
                                                         float* __restrict__ P, int M, int NS) {
  constexpr int S = lds_stride(D);            // conflict-free fragment reads
  constexpr int KS = D / 4;
  constexpr int G = dpack_groups(D);
  constexpr int CT = cdiv(D, 16);
  constexpr int SP = ((D + 3) / 4) * 4 + 4;   // row stride of partial tiles (16-byte aligned rows)
  constexpr int D4 = D / 4;
  constexpr int NP = (CPW + 1) / 2;           // chunk pairs (two independent GEMM1 chains in flight)
  __shared__ __align__(16) float at[16 * S];      // attention-output rows, then x1
  __shared__ __align__(16) float pre[16 * SP];    // residual rows -> pre-LN1 rows
  __shared__ __align__(16) float part[4 * 16 * SP];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x, split = blockIdx.y;
  const int m0 = tile * 16;
  const int rows_valid = min(16, M - m0);
  const int r = lane & 15, q = lane >> 4;

  // ---- this wave's weight fragments ----
  // out-projection: column tiles ct = wave, wave + 4
  const float4* Woq = reinterpret_cast<const float4*>(Wop);
  float4 wo[2][G];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ct = wave + 4 * i;
#pragma unroll
    for (int g = 0; g < G; ++g) wo[i][g] = ct < CT ? Woq[((size_t)ct * G + g) * 64 + lane] : float4{0.f, 0.f, 0.f, 0.f};
  }
  // FFN: this workgroup's F / NS hidden units are split over the 4 waves, 16 per chunk
  const int fc0 = (split * 4 + wave) * CPW;              // first chunk of this wave
  const float4* W1q = reinterpret_cast<const float4*>(W1p) + lane;
  const float4* W2q = reinterpret_cast<const float4*>(W2p) + lane;
  // the first pair's fragments are requested here, under the staging and the out-projection
  float4 w1[2][G], w2[2][CT], bq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int fc = fc0 + (j < CPW ? j : 0);
#pragma unroll
    for (int g = 0; g < G; ++g) w1[j][g] = W1q[((size_t)fc * G + g) * 64];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) w2[j][ct] = W2q[((size_t)fc * CT + ct) * 64];
    bq[j] = *reinterpret_cast<const float4*>(b1 + 16 * fc + 4 * q);
  }

  // ---- stage the two row tiles ----
  for (int f = threadIdx.x; f < 16 * D4; f += 256) {
    const int rr = f / D4, c4 = f - rr * D4;
    const size_t off = (size_t)(m0 + min(rr, rows_valid - 1)) * D + 4 * c4;
    const float4 a = *reinterpret_cast<const float4*>(A + off);
    const float4 x = *reinterpret_cast<const float4*>(R + off);
    float2* d2 = reinterpret_cast<float2*>(&at[rr * S + 4 * c4]);
    d2[0] = float2{a.x, a.y}, d2[1] = float2{a.z, a.w};
    *reinterpret_cast<float4*>(&pre[rr * SP + 4 * c4]) = x;
  }
  __syncthreads();

  // ---- out-projection + bias + residual: lane holds columns n .. n+3 of row r ----
  {
    float af[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) af[s] = at[r * S + 4 * s + q];
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 w4 = wo[i][s >> 2];
        const float a = (s & 3) == 0 ? w4.x : (s & 3) == 1 ? w4.y : (s & 3) == 2 ? w4.z : w4.w;
        acc[i] = mfma16(a, af[s], acc[i]);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = 16 * (wave + 4 * i) + 4 * q;
      if (wave + 4 * i < CT && n < D) {
        const float4 b4 = *reinterpret_cast<const float4*>(bo + n);
        float4* p4 = reinterpret_cast<float4*>(&pre[r * SP + n]);
        const float4 x4 = *p4;
        *p4 = float4{acc[i][0] + b4.x + x4.x, acc[i][1] + b4.y + x4.y, acc[i][2] + b4.z + x4.z, acc[i][3] + b4.w + x4.w};
      }
    }
  }
  __syncthreads();

  // ---- LayerNorm1: 16 threads per row, float4 chunks c4 = sub, sub + 16 ----
  {
    const int row = threadIdx.x >> 4, sub = threadIdx.x & 15;
    float4 v[2];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c4 = sub + 16 * i;
      v[i] = c4 < D4 ? *reinterpret_cast<const float4*>(&pre[row * SP + 4 * c4]) : float4{0.f, 0.f, 0.f, 0.f};
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
    const float mean = sum * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (sub + 16 * i < D4) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        ss = fmaf(a, a, ss), ss = fmaf(b, b, ss), ss = fmaf(c, c, ss), ss = fmaf(d, d, ss);
      }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 16);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c4 = sub + 16 * i;
      if (c4 < D4) {
        const float4 g4 = *reinterpret_cast<const float4*>(g1 + 4 * c4), e4 = *reinterpret_cast<const float4*>(e1 + 4 * c4);
        const float4 o = {(v[i].x - mean) * rstd * g4.x + e4.x, (v[i].y - mean) * rstd * g4.y + e4.y,
                          (v[i].z - mean) * rstd * g4.z + e4.z, (v[i].w - mean) * rstd * g4.w + e4.w};
        float2* d2 = reinterpret_cast<float2*>(&at[row * S + 4 * c4]);
        d2[0] = float2{o.x, o.y}, d2[1] = float2{o.z, o.w};
        if (split == 0 && row < rows_valid) *reinterpret_cast<float4*>(X1 + (size_t)(m0 + row) * D + 4 * c4) = o;
      }
    }
  }
  __syncthreads();

  // ---- FFN over this wave's chunks, two at a time (independent GEMM1 chains) ----
  float xf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) xf[s] = at[r * S + 4 * s + q];
  f32x4 yacc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) yacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    constexpr bool odd = (CPW & 1) != 0;  // CPW = 1: the second chain repeats the first and is dropped
    f32x4 h[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < (odd ? 1 : 2); ++j) {
        const float4 w4 = w1[j][s >> 2];
        const float a = (s & 3) == 0 ? w4.x : (s & 3) == 1 ? w4.y : (s & 3) == 2 ? w4.z : w4.w;
        h[j] = mfma16(a, xf[s], h[j]);
      }
    // bias + ReLU on the way out of the accumulator
    float hv[2][4];
#pragma unroll
    for (int j = 0; j < (odd ? 1 : 2); ++j) {
      const float bv[4] = {bq[j].x, bq[j].y, bq[j].z, bq[j].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) hv[j][i] = __builtin_amdgcn_fmed3f(h[j][i] + bv[i], 0.f, __builtin_inff());
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < (odd ? 1 : 2); ++j) {
          const float4 w4 = w2[j][ct];
          const float a = i == 0 ? w4.x : i == 1 ? w4.y : i == 2 ? w4.z : w4.w;
          yacc[ct] = mfma16(a, hv[j][i], yacc[ct]);
        }
    if (p + 1 < NP) {  // next pair (compile-time condition after unrolling)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int fc = fc0 + 2 * (p + 1) + j;
#pragma unroll
        for (int g = 0; g < G; ++g) w1[j][g] = W1q[((size_t)fc * G + g) * 64];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w2[j][ct] = W2q[((size_t)fc * CT + ct) * 64];
        bq[j] = *reinterpret_cast<const float4*>(b1 + 16 * fc + 4 * q);
      }
    }
  }

  // ---- sum the four waves' partial tiles in wave order, store the workgroup's partial ----
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int n = 16 * ct + 4 * q;
    if (n < D) *reinterpret_cast<float4*>(&part[(wave * 16 + r) * SP + n]) = float4{yacc[ct][0], yacc[ct][1], yacc[ct][2], yacc[ct][3]};
  }
  __syncthreads();
  float* Pt = P + ((size_t)tile * NS + split) * 16 * D;
  for (int f = threadIdx.x; f < 16 * D4; f += 256) {
    const int rr = f / D4, c4 = f - rr * D4;
    float4 a = *reinterpret_cast<const float4*>(&part[(0 * 16 + rr) * SP + 4 * c4]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 b = *reinterpret_cast<const float4*>(&part[(w * 16 + rr) * SP + 4 * c4]);
      a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
    }
    reinterpret_cast<float4*>(Pt)[f] = a;
  }
}

// Mid-size M (a few thousand to a few ten thousand rows; the reference's default sample_batch_size = 50 and its former
// 200 are here): too many rows for 16-row tiles to stay efficient (every tile streams all weights), too few 64-row
// tiles to fill the chip's 512 workgroup slots evenly (585 tiles = 1.14 rounds cost 2).  k_ffn_part is the 64-row
// main loop of k_ffn_ln over hidden units [s F / NS, (s + 1) F / NS) only, grid (64-row tiles, NS): the work comes in
// NS times finer units, and the partial Y tiles go through the same k_ffn_reduce_ln as the 16-row form (a 64-row tile
// is written as four 16-row partial tiles).  Input is x1 (k_linear_res_ln stays a separate launch here).
template <int D>
__global__ __launch_bounds__(256, 2) void k_ffn_part(const float* __restrict__ X, const float* __restrict__ W1p,
                                                     const float* __restrict__ b1, const float* __restrict__ W2p,
                                                     float* __restrict__ P, int M, int F, int NS) {
  constexpr int S = lds_stride(D);
  constexpr int KS = D / 4;
  constexpr int G = dpack_groups(D);
  constexpr int CT = cdiv(D, 16);
  constexpr int MB = 4, R = 64;
  constexpr int SP = ((D + 3) / 4) * 4 + 4;  // partial-sum row stride (16-byte aligned rows)
  constexpr int D4 = D / 4;
  __shared__ __align__(16) float xs[R * S];
  __shared__ __align__(16) float red[2 * R * SP];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x, split = blockIdx.y;
  const int m0 = tile * R;
  const int rows_valid = min(R, M - m0);
  const int n = lane & 15, q = lane >> 4;

  // this wave's chunks (16 hidden units each) of the workgroup's F / NS slice
  const int nchunk = F / (64 * NS);
  const int fc0 = (split * 4 + wave) * nchunk;
  const float4* W1q = reinterpret_cast<const float4*>(W1p) + (size_t)fc0 * G * 64 + lane;
  const float4* W2q = reinterpret_cast<const float4*>(W2p) + (size_t)fc0 * CT * 64 + lane;
  const float4* b1q = reinterpret_cast<const float4*>(b1 + 16 * fc0) + q;
  float4 w1[G], w2[CT], bv;
  auto load_w1 = [&](int c) {
#pragma unroll
    for (int g = 0; g < G; ++g) w1[g] = W1q[((size_t)c * G + g) * 64];
    bv = b1q[c * 4];
  };
  auto load_w2 = [&](int c) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) w2[ct] = W2q[((size_t)c * CT + ct) * 64];
  };
  load_w1(0);
  load_w2(0);

  // ---- stage the X tile (rows past M repeat the last valid row; their partials are never read) ----
  for (int f = threadIdx.x; f < R * D4; f += 256) {
    const int rr = f / D4, c4 = f - rr * D4;
    const float4 v = *reinterpret_cast<const float4*>(X + (size_t)(m0 + min(rr, rows_valid - 1)) * D + 4 * c4);
    float2* dst = reinterpret_cast<float2*>(&xs[rr * S + 4 * c4]);
    dst[0] = float2{v.x, v.y}, dst[1] = float2{v.z, v.w};
  }
  __syncthreads();
  float xf[MB][KS];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[mb][s] = xs[(16 * mb + n) * S + 4 * s + q];
  f32x4 yacc[CT][MB];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) yacc[ct][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the single-register-buffer software pipeline of k_ffn_ln: W1(c + 1) streams in under GEMM2(c), W2(c + 1) under
  // GEMM1(c + 1); the sched_barriers pin the issue points
  for (int ci = 0; ci < nchunk; ++ci) {
    const int nx = (ci + 1 < nchunk) ? ci + 1 : ci;
    f32x4 h[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) h[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 w4 = w1[s >> 2];
      const float a = (s & 3) == 0 ? w4.x : (s & 3) == 1 ? w4.y : (s & 3) == 2 ? w4.z : w4.w;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) h[mb] = mfma16(a, xf[mb][s], h[mb]);
    }
    const float bvv[4] = {bv.x, bv.y, bv.z, bv.w};
    __builtin_amdgcn_sched_barrier(0);
    load_w1(nx);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[mb][r] = __builtin_amdgcn_fmed3f(h[mb][r] + bvv[r], 0.f, __builtin_inff());
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float4 w4 = w2[ct];
        const float a = r == 0 ? w4.x : r == 1 ? w4.y : r == 2 ? w4.z : w4.w;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) yacc[ct][mb] = mfma16(a, h[mb][r], yacc[ct][mb]);
      }
    __builtin_amdgcn_sched_barrier(0);
    load_w2(nx);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- (w0 + w2) + (w1 + w3) through two LDS images, then the partial tile leaves as four 16-row tiles ----
  auto img_at = [&](int img, int mb, int ct) { return red + ((size_t)img * R + 16 * mb + n) * SP + 16 * ct + 4 * q; };
  if (wave < 2) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        if (16 * ct + 4 * q < D)
          *reinterpret_cast<float4*>(img_at(wave, mb, ct)) = float4{yacc[ct][mb][0], yacc[ct][mb][1], yacc[ct][mb][2], yacc[ct][mb][3]};
  }
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        if (16 * ct + 4 * q < D) {
          float4* p = reinterpret_cast<float4*>(img_at(wave - 2, mb, ct));
          const float4 a = *p;
          *p = float4{a.x + yacc[ct][mb][0], a.y + yacc[ct][mb][1], a.z + yacc[ct][mb][2], a.w + yacc[ct][mb][3]};
        }
  }
  __syncthreads();
  for (int f = threadIdx.x; f < R * D4; f += 256) {
    const int rr = f / D4, c4 = f - rr * D4;
    const float4 a = *reinterpret_cast<const float4*>(&red[rr * SP + 4 * c4]);
    const float4 b = *reinterpret_cast<const float4*>(&red[(R + rr) * SP + 4 * c4]);
    float* Pt = P + ((size_t)(4 * tile + (rr >> 4)) * NS + split) * 16 * D + (size_t)(rr & 15) * D;
    *reinterpret_cast<float4*>(Pt + 4 * c4) = float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
  }
}

template <int GN>
__device__ __forceinline__ float4 sum_partials(const float* p0, int NS, int stride) {
  float4 acc = float4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < NS; s0 += GN) {
    float4 p[GN];
#pragma unroll
    for (int j = 0; j < GN; ++j) p[j] = *reinterpret_cast<const float4*>(p0 + (size_t)min(s0 + j, NS - 1) * stride);
#pragma unroll
    for (int j = 0; j < GN; ++j) {
      const float k = s0 + j < NS ? 1.f : 0.f;
      acc.x = fmaf(p[j].x, k, acc.x), acc.y = fmaf(p[j].y, k, acc.y);
      acc.z = fmaf(p[j].z, k, acc.z), acc.w = fmaf(p[j].w, k, acc.w);
    }
  }
  return acc;
}

// y = LN2(x1 + b2 + sum_s P[tile][s]), partials added in split order; 16 threads per row
template <int D>
__global__ __launch_bounds__(256) void k_ffn_reduce_ln(const float* __restrict__ X1, const float* __restrict__ P,
                                                       const float* __restrict__ b2, const float* __restrict__ g2,
                                                       const float* __restrict__ e2, float* __restrict__ Y, int M,
                                                       int NS) {
  constexpr int D4 = D / 4;
  const int tile = blockIdx.x;
  const int row = threadIdx.x >> 4, sub = threadIdx.x & 15;
  const int m = tile * 16 + row;
  const int mr = min(m, M - 1);
  const float* Pt = P + (size_t)tile * NS * 16 * D + (size_t)row * D;
  float4 v[2];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c4 = sub + 16 * i;
    v[i] = float4{0.f, 0.f, 0.f, 0.f};
    if (c4 < D4) {
      const float4 x = *reinterpret_cast<const float4*>(X1 + (size_t)mr * D + 4 * c4);
      // all partials of a group requested before the first is used (index clamped, surplus ones add 0), summed in
      // split order; 16 splits (the batch-1 form) go out as one group: one memory round trip instead of two
      const float4 acc = NS > 8 ? sum_partials<16>(Pt + 4 * c4, NS, 16 * D) : sum_partials<8>(Pt + 4 * c4, NS, 16 * D);
      const float4 b = *reinterpret_cast<const float4*>(b2 + 4 * c4);
      v[i] = float4{(x.x + acc.x) + b.x, (x.y + acc.y) + b.y, (x.z + acc.z) + b.z, (x.w + acc.w) + b.w};
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
  const float mean = sum * (1.0f / D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (sub + 16 * i < D4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss = fmaf(a, a, ss), ss = fmaf(b, b, ss), ss = fmaf(c, c, ss), ss = fmaf(d, d, ss);
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 16);
  const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
  if (m < M) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c4 = sub + 16 * i;
      if (c4 < D4) {
        const float4 g4 = *reinterpret_cast<const float4*>(g2 + 4 * c4), e4 = *reinterpret_cast<const float4*>(e2 + 4 * c4);
        *reinterpret_cast<float4*>(Y + (size_t)m * D + 4 * c4) =
            float4{(v[i].x - mean) * rstd * g4.x + e4.x, (v[i].y - mean) * rstd * g4.y + e4.y,
                   (v[i].z - mean) * rstd * g4.z + e4.z, (v[i].w - mean) * rstd * g4.w + e4.w};
      }
    }
  }
}

thread_local int g_small_wgs = 0;   // ffd_tune "small_wgs": most workgroups (row tiles x F splits) of the split pair; 0 = heuristic
thread_local int g_small_path = 1;  // ffd_tune "small_path": 0 disables the split out-proj + FFN pair

// F splits for M rows (0: use the large-M kernels).  The most splits (<= 16, F / (64 NS) chunks per wave in
// {2, 4, 8, 16}: the kernel's instances) that keep the grid within 2.5 workgroups per CU; failing that, within 6
// (tools/sweep_mid.py on the ECG shape: the pair beats every tile height of k_ffn_ln up to M ~ 12 000 rows, where
// 16-row tiles no longer fit the chip in one round and 64-row tiles leave half of it idle).
int small_path_splits(int M, int D, int F) {
  if (!g_small_path || D % 4 != 0 || D > 128 || F % 64 != 0) return 0;
  if (g_small_wgs == 0 && ffn_height_plan(M, D, F)) return 0;  // (one 32- / 48-row tile per CU is the faster form there)
  const int tiles = cdiv(M, 16);
  const int caps[2] = {g_small_wgs > 0 ? g_small_wgs : 5 * num_cus() / 2, g_small_wgs > 0 ? g_small_wgs : 6 * num_cus()};
  for (int cap : caps)
    for (int ns = 16; ns >= 2; ns >>= 1) {
      if ((F / 64) % ns != 0 || tiles * ns > cap) continue;
      const int cpw = F / (64 * ns);
      if (cpw == 2 || cpw == 4 || cpw == 8 || cpw == 16) return ns;
    }
  return 0;
}

thread_local int g_mid_path = 1;  // ffd_tune "mid_path": 0 off, 1 heuristic, 2 / 4 / 8 force that many F slices where the form applies

// F slices of the 64-row form for M rows, 0 = not this form (checked after small_path_splits).  Model fitted to
// tools/sweep_mid.py: a CU retires a 64-row tile of k_ffn_ln every ~73 us whether it hosts one workgroup or two, so
// the persistent kernel costs ceil(tiles / CUs) tile times; four F slices cost ceil(4 tiles / CUs) / 4 of them, times
// 1.15 for what a slice pays per unit (X staging, first weight fetch, partial tile, no 4x4x1 remainder path, the
// reduce launch).  Wins at 281 tiles (B = 96 at L = 187: 165 -> 127 us), 374 (167 -> 151), 585 (B = 200: 244 -> 228);
// two slices were never better than four, eight only equal.
int mid_path_splits(int M, int D, int F) {
  if (!g_mid_path || D % 4 != 0 || D > 128) return 0;
  if (g_mid_path == 1 && ffn_height_plan(M, D, F)) return 0;
  if (g_mid_path > 1) return (F / 64) % g_mid_path == 0 ? g_mid_path : 0;
  if ((F / 64) % 4 != 0) return 0;
  const int tiles = cdiv(M, 64), cus = num_cus();
  return 1.15 * cdiv(4 * tiles, cus) / 4.0 < (double)cdiv(tiles, cus) ? 4 : 0;
}

template <int D>
static hipError_t launch_mid_t(const float* x1, const LayerWeights& w, float* P, float* Y, int M, int F, int NS, hipStream_t s) {
  hipLaunchKernelGGL(k_ffn_part<D>, dim3(cdiv(M, 64), NS), dim3(256), 0, s, x1, w.w1p, w.b1, w.w2p, P, M, F, NS);
  hipLaunchKernelGGL(k_ffn_reduce_ln<D>, dim3(cdiv(M, 16)), dim3(256), 0, s, x1, P, w.b2, w.n2w, w.n2b, Y, M, NS);
  return hipGetLastError();
}

// y = LN2(x1 + FFN(x1)) with F cut into NS slices per 64-row tile; P: small_path_partial_floats(M rounded up to 64, D, NS)
hipError_t launch_ffn_mid(const float* x1, const LayerWeights& w, float* P, float* Y, int M, int D, int F, int NS, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  switch (D) {
#define X(d) \
  case d: return launch_mid_t<d>(x1, w, P, Y, M, F, NS, s);
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

size_t small_path_partial_floats(int M, int D, int NS) { return (size_t)cdiv(M, 16) * NS * 16 * D; }

template <int D>
static hipError_t launch_small_t(const float* attn, const float* xres, const LayerWeights& w, float* x1, float* P,
                                 float* Y, int M, int F, int NS, hipStream_t s) {
  const dim3 grid(cdiv(M, 16), NS), block(256);
  switch (F / (64 * NS)) {
#define FFD_CPW(c)                                                                                                    \
  case c:                                                                                                             \
    hipLaunchKernelGGL((k_oproj_ffn_split<D, c>), grid, block, 0, s, attn, w.out_wp, w.out_b, xres, w.n1w, w.n1b, x1, \
                       w.w1p, w.b1, w.w2p, P, M, NS);                                                                 \
    break;
    FFD_CPW(2) FFD_CPW(4) FFD_CPW(8) FFD_CPW(16)
#undef FFD_CPW
    default: return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(k_ffn_reduce_ln<D>, dim3(cdiv(M, 16)), block, 0, s, x1, P, w.b2, w.n2w, w.n2b, Y, M, NS);
  return hipGetLastError();
}

hipError_t launch_oproj_ffn_small(const float* attn, const float* xres, const LayerWeights& w, float* x1, float* P,
                                  float* Y, int M, int D, int F, int NS, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  switch (D) {
#define X(d) \
  case d: return launch_small_t<d>(attn, xres, w, x1, P, Y, M, F, NS, s);
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ffd
