// One launch for everything of an encoder layer that is row-local (cached_transformer.py:314-327
// plus the next layer's in_proj, :228-234 / nn.MultiheadAttention in_proj):
//
//   x1 = LayerNorm1( x + attn Wo^T + bo )                      (prologue, 360 MFMA / tile)
//   y  = LayerNorm2( x1 + W2 relu(W1 x1 + b1) + b2 )           (main loop, 19 456 MFMA / tile)
//   q,k,v(next layer) = y Win'^T + bin'   -> head-major         (epilogue, <= 1 008 MFMA / tile)
//
// per 64-row tile.  The row tile lives in LDS / registers from the attention output to the
// next layer's Q/K/V: compared with three launches this removes two kernel boundaries, the
// x1 round trip (2 x M*d*4 bytes), one re-read of y, and two extra X-tile stagings, and the small
// GEMMs run inside the FFN kernel's resident workgroups.  Only attention itself (which mixes
// rows of a sample) stays a separate kernel.
//
// The FFN main loop is the one of k_ffn_ln (ffd_ffn.hip): exact-fp32 v_mfma_f32_16x16x4_f32,
// GEMM1 accumulator registers used directly as GEMM2's B operand, 4 waves splitting F, pinned
// software pipeline for the packed weights, deterministic cross-wave reduction.
#include "ffd_internal.h"

namespace ffd {

template <int D>
__global__ __launch_bounds__(256, 2) void k_layer(const float* __restrict__ attn, const float* __restrict__ xres,
                                                  LayerWeights w, float* __restrict__ Y, NextProj nx, int M, int F, int stagger) {
  constexpr int MB = 4;
  constexpr int R = 16 * MB;
  constexpr int S = lds_stride(D);
  constexpr int KS = D / 4;
  constexpr int G = dpack_groups(D);
  constexpr int CT = cdiv(D, 16);
  constexpr int S2 = ((D + 3) / 4) * 4 + 4;
  __shared__ __align__(16) float xs[R * S];
  __shared__ __align__(16) float red[3 * R * S2];  // prologue: attention tile; epilogue: partial sums
  float* as = red;

  const int m0 = blockIdx.x * R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows_valid = min(R, M - m0);

  if (stagger > 0) {  // de-phase the two workgroups of a CU (see k_ffn_ln)
    const unsigned hwid = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    if (hwid & 1)
      for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }
  // ------------------------------------------------------------------ prologue
  // out-proj weight fragments first: their L2 latency hides under the tile staging
  float4 wo[CT][G];
  {
    const float4* Wq = reinterpret_cast<const float4*>(w.out_wp);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int g = 0; g < G; ++g) wo[ct][g] = Wq[((size_t)ct * G + g) * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    const float4* A4 = reinterpret_cast<const float4*>(attn + (size_t)m0 * D);
    const float4* X4 = reinterpret_cast<const float4*>(xres + (size_t)m0 * D);
    for (int i4 = threadIdx.x; i4 < R * D / 4; i4 += 256) {
      const int r = (4 * i4) / D, k = 4 * i4 - r * D;
      const bool ok = r < rows_valid;
      const float4 a = ok ? A4[i4] : float4{0.f, 0.f, 0.f, 0.f};
      const float4 x = ok ? X4[i4] : float4{0.f, 0.f, 0.f, 0.f};
      float2* da = reinterpret_cast<float2*>(&as[r * S + k]);
      float2* dx = reinterpret_cast<float2*>(&xs[r * S + k]);
      da[0] = float2{a.x, a.y};
      da[1] = float2{a.z, a.w};
      dx[0] = float2{x.x, x.y};
      dx[1] = float2{x.z, x.w};
    }
  }
  __syncthreads();
  {
    // out-proj for m-block `wave`: rows 16*wave + (lane&15), all CT column tiles
    float af[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) af[s] = as[(16 * wave + (lane & 15)) * S + 4 * s + (lane >> 4)];
    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float4 q = wo[ct][s >> 2];
        const float a = (s & 3) == 0 ? q.x : (s & 3) == 1 ? q.y : (s & 3) == 2 ? q.z : q.w;
        acc[ct] = mfma16(a, af[s], acc[ct]);
      }
    }
    float* xrow = &xs[(16 * wave + (lane & 15)) * S];
    float v[CT][4];
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * ct + 4 * (lane >> 4) + r;
        if (n < D) {
          v[ct][r] = acc[ct][r] + w.out_b[n] + xrow[n];
          sum += v[ct][r];
        } else {
          v[ct][r] = 0.f;
        }
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * ct + 4 * (lane >> 4) + r;
        if (n < D) {
          const float dlt = v[ct][r] - mean;
          ss = fmaf(dlt, dlt, ss);
        }
      }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * ct + 4 * (lane >> 4) + r;
        if (n < D) xrow[n] = (v[ct][r] - mean) * rstd * w.n1w[n] + w.n1b[n];  // x1: FFN input and residual
      }
  }
  // ------------------------------------------------------------------ FFN main loop
  // first weight fragments are requested before the barrier that publishes x1
  const int nchunk = F / 64;
  const int fc0 = wave * nchunk;
  const float4* W1q = reinterpret_cast<const float4*>(w.w1p) + (size_t)fc0 * G * 64 + lane;
  const float4* W2q = reinterpret_cast<const float4*>(w.w2p) + (size_t)fc0 * CT * 64 + lane;
  const float4* b1q = reinterpret_cast<const float4*>(w.b1 + 16 * fc0) + (lane >> 4);
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
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  float xf[MB][KS];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[mb][s] = xs[(16 * mb + (lane & 15)) * S + 4 * s + (lane >> 4)];

  f32x4 yacc[CT][MB];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) yacc[ct][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int ci = 0; ci < nchunk; ++ci) {
    const int nxc = (ci + 1 < nchunk) ? ci + 1 : ci;
    f32x4 h[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) h[mb] = f32x4{bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 q = w1[s >> 2];
      const float a = (s & 3) == 0 ? q.x : (s & 3) == 1 ? q.y : (s & 3) == 2 ? q.z : q.w;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) h[mb] = mfma16(a, xf[mb][s], h[mb]);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_w1(nxc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[mb][r] = __builtin_amdgcn_fmed3f(h[mb][r], 0.f, __builtin_inff());
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float4 q = w2[ct];
        const float a = r == 0 ? q.x : r == 1 ? q.y : r == 2 ? q.z : q.w;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) yacc[ct][mb] = mfma16(a, h[mb][r], yacc[ct][mb]);
      }
    __builtin_amdgcn_sched_barrier(0);
    load_w2(nxc);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ------------------------------------------------------------------ reduction + LN2
  __syncthreads();  // every wave is done reading `as` (aliases red) long ago; xs fragments are in registers
  if (wave == 0) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * ct + 4 * (lane >> 4) + r;
          if (c < D) xs[(16 * mb + (lane & 15)) * S + c] += yacc[ct][mb][r];
        }
  } else {
    float* rw = red + (size_t)(wave - 1) * R * S2;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int c = 16 * ct + 4 * (lane >> 4);
        if (c < D)
          *reinterpret_cast<float4*>(&rw[(16 * mb + (lane & 15)) * S2 + c]) =
              float4{yacc[ct][mb][0], yacc[ct][mb][1], yacc[ct][mb][2], yacc[ct][mb][3]};
      }
  }
  __syncthreads();

  constexpr int TPR = 256 / R;  // 4 threads per row
  const int row = threadIdx.x / TPR, sub = threadIdx.x % TPR;
  const int m = m0 + row;
  float vals[cdiv(D, TPR)];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < cdiv(D, TPR); ++i) {
    const int c = sub + i * TPR;
    float v = 0.f;
    if (c < D) {
      v = xs[row * S + c];
      v += red[(0 * R + row) * S2 + c];
      v += red[(1 * R + row) * S2 + c];
      v += red[(2 * R + row) * S2 + c];
      v += w.b2[c];
      sum += v;
    }
    vals[i] = v;
  }
#pragma unroll
  for (int o = TPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum * (1.0f / D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < cdiv(D, TPR); ++i) {
    const int c = sub + i * TPR;
    if (c < D) {
      const float dlt = vals[i] - mean;
      ss = fmaf(dlt, dlt, ss);
    }
  }
#pragma unroll
  for (int o = TPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
#pragma unroll
  for (int i = 0; i < cdiv(D, TPR); ++i) {
    const int c = sub + i * TPR;
    if (c < D) {
      const float y = (vals[i] - mean) * rstd * w.n2w[c] + w.n2b[c];
      if (m < M) Y[(size_t)m * D + c] = y;
      xs[row * S + c] = y;  // stays resident for the next layer's projection
    }
  }
  if (nx.nreg == 0) return;
  __syncthreads();

  // ------------------------------------------------------------------ next layer's Q / K / V
  // Results are staged in LDS (the free `red` region) in exactly the head-major order of the
  // destination -- [(region, head)][row][e] -- and then copied out as contiguous runs.
  {
    const int N = nx.nreg * D;
    const int NT = cdiv(N, 16);
    const float4* Wq = reinterpret_cast<const float4*>(nx.wp);
    float4 wq[G], wn[G];
    if (wave < NT) {
#pragma unroll
      for (int g = 0; g < G; ++g) wq[g] = Wq[((size_t)wave * G + g) * 64 + lane];
    }
    float yf[MB][KS];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int s = 0; s < KS; ++s) yf[mb][s] = xs[(16 * mb + (lane & 15)) * S + 4 * s + (lane >> 4)];
    float* stage = red;
    const int hd = nx.hd;
    for (int nt = wave; nt < NT; nt += 4) {
      const int nn4 = nt + 4;
      if (nn4 < NT) {
#pragma unroll
        for (int g = 0; g < G; ++g) wn[g] = Wq[((size_t)nn4 * G + g) * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float4 q = wq[s >> 2];
        const float a = (s & 3) == 0 ? q.x : (s & 3) == 1 ? q.y : (s & 3) == 2 ? q.z : q.w;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = mfma16(a, yf[mb][s], acc[mb]);
      }
      const int n = 16 * nt + 4 * (lane >> 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n + r;
        if (nn < N) {
          const int rh = nn / hd, e = nn - rh * hd;  // rh = region * H + head  (D = H * hd)
          const float bb = nx.bias[nn];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) stage[(rh * R + 16 * mb + (lane & 15)) * hd + e] = acc[mb][r] + bb;
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) wq[g] = wn[g];
    }
    __syncthreads();
    // copy-out: run (region, head) = R*hd contiguous floats in LDS; rows of one sample are contiguous in HBM
    const int run = R * hd;
    const int total = nx.nreg * nx.H * run;
    for (int i = threadIdx.x; i < total; i += 256) {
      const int rh = i / run, rem = i - rh * run;
      const int rr = rem / hd, e = rem - rr * hd;
      const int mm = m0 + rr;
      if (mm < M) {
        const int reg = rh / nx.H, hh = rh - reg * nx.H;
        const int b = mm / nx.L, l = mm - b * nx.L;
        float* dst = reg == 0 ? nx.q : reg == 1 ? nx.k : nx.v;
        dst[(((size_t)b * nx.H + hh) * nx.L + l) * hd + e] = stage[i];
      }
    }
  }
}

int g_fuse_layer = 0;  // 0 off (default: measured slower than the 3-kernel form), 1 on, -1 auto when the grid fills the chip

hipError_t launch_layer(const float* attn, const float* xres, const LayerWeights& w, float* Y, const NextProj& nx,
                        int M, int D, int F, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  if (F % 64 != 0) return hipErrorInvalidValue;
  dim3 grid(cdiv(M, 64)), block(256);
  switch (D) {
#define X(d) \
    case d: hipLaunchKernelGGL(k_layer<d>, grid, block, 0, s, attn, xres, w, Y, nx, M, F, g_ffn_stagger > 0 ? g_ffn_stagger : 0); break;
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace ffd
