// Fused feed-forward block of the post-norm encoder layer (cached_transformer.py:325-327,
// nn.TransformerEncoderLayer._ff_block):
//     Y = LayerNorm2( X + W2 relu(W1 X + b1) + b2 )
// >= 86 % of the FLOPs of a score evaluation.  MFMA-bound (exact fp32
// v_mfma_f32_16x16x4_f32, 157 TFLOP/s peak); the (rows x F) hidden never leaves
// registers:
//
//   GEMM1  H^T[f][m] = sum_k W1[f][k] X[m][k]      A = W1 fragment, B = X fragment
//   relu(H + b1) stays in the accumulator registers, whose layout
//       lane l, reg r  <->  H^T[f = 4 (l>>4) + r][m = l & 15]
//   is exactly the B-operand layout (k = l>>4, j = l&15) of the next MFMA when the
//   A operand is pre-packed as W2[c = l&15][f = 4 (l>>4) + r]  ("w2pack"), so
//   GEMM2  Y^T[c][m] += sum_f W2[c][f] H^T[f][m]   needs no LDS and no lane movement.
//
// Work split: one workgroup = 16*MB rows; its 4 waves (one per SIMD) split F four
// ways, each keeping the X fragments (MB*D/4 VGPRs) and a private Y^T accumulator
// (MB*ceil(D/16)*4 VGPRs) resident and streaming only packed weights from L2
// (2 coalesced float4 per lane per 16-wide F chunk).  Partial Y^T tiles are summed
// in wave order through the LDS copy of the X tile (deterministic), which also
// provides the residual; LayerNorm2 runs on that tile and rows are written back
// fully coalesced.
#include "ffd_internal.h"

namespace ffd {

// REM = true: the d % 16 remainder rows of GEMM2 (8 of 72) run on v_mfma_f32_4x4x1_16b_f32
// (8 cycles per instruction) instead of a zero-padded 16-row tile: block b = lane>>2 multiplies
// A[lane 4b+i] by B[lane 4b+j], and with a GEMM1 accumulator register as B that is
//   W2[c0 + 4g + i][16 fc + 4q + r] * H^T[16 fc + 4q + r][16 mb + 4 mq + j],  q = lane>>4, mq = (lane>>2)&3,
// so each lane-quarter q accumulates the partial sum over its hidden units and the four
// partials are added with two cross-lane xor-adds once per tile.
template <int D, int MB, bool REM>
__global__ __launch_bounds__(256, (MB <= 4 ? 2 : 1)) void k_ffn_ln(const float* __restrict__ X, const float* __restrict__ W1p,
                                                  const float* __restrict__ b1, const float* __restrict__ W2p,
                                                  const float* __restrict__ W2r, const float* __restrict__ b2,
                                                  const float* __restrict__ gam, const float* __restrict__ bet,
                                                  float* __restrict__ Y, int M, int F, int stagger) {
  constexpr int S = lds_stride(D);
  constexpr int KS = D / 4;
  constexpr int G = dpack_groups(D);
  constexpr int CTP = cdiv(D, 16);                       // tiles in the packed w2 image
  constexpr int NG = (REM && D >= 16) ? w2rem_groups(D) : 0;  // 4-row remainder groups on the 4x4x1 path
  constexpr int CT = (NG > 0) ? D / 16 : cdiv(D, 16);     // 16-row tiles on the 16x16x4 path
  constexpr int NGA = NG > 0 ? NG : 1;
  constexpr int R = 16 * MB;
  constexpr int S2 = ((D + 3) / 4) * 4 + 4;  // partial-sum row stride (16-byte aligned rows)
  __shared__ __align__(16) float xs[R * S];
  __shared__ __align__(16) float red[3 * R * S2];

  const int m0 = blockIdx.x * R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // ---- stage the X tile (coalesced float4) and pull this wave's B fragments ----
  {
    const float4* X4 = reinterpret_cast<const float4*>(X + (size_t)m0 * D);
    const int rows_valid = min(R, M - m0);
    for (int i4 = threadIdx.x; i4 < R * D / 4; i4 += 256) {
      const int r = (4 * i4) / D, k = 4 * i4 - r * D;
      float4 v = (r < rows_valid) ? X4[i4] : float4{0.f, 0.f, 0.f, 0.f};
      float2* dst = reinterpret_cast<float2*>(&xs[r * S + k]);  // S even: 8-byte aligned
      dst[0] = float2{v.x, v.y};
      dst[1] = float2{v.z, v.w};
    }
  }
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
  f32x4 yrem[NGA][MB];
#pragma unroll
  for (int g = 0; g < NGA; ++g)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) yrem[g][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- main loop over this wave's quarter of F, 16 hidden units per chunk ----
  const int nchunk = F / 64;  // chunks per wave
  const int fc0 = wave * nchunk;
  const float4* W1q = reinterpret_cast<const float4*>(W1p) + (size_t)fc0 * G * 64 + lane;
  const float4* W2q = reinterpret_cast<const float4*>(W2p) + (size_t)fc0 * CTP * 64 + lane;
  const float4* W2rq = reinterpret_cast<const float4*>(W2r) + (size_t)fc0 * NGA * 64 + lane;
  const float4* b1q = reinterpret_cast<const float4*>(b1 + 16 * fc0) + (lane >> 4);

  // Software pipeline with single register buffers: W1(ci+1) streams in while GEMM2(ci)
  // runs (the W1 registers are dead after GEMM1), W2(ci+1) streams in while GEMM1(ci+1)
  // runs.  Each load set therefore has >= 18*MB MFMAs (>= 4.6k cycles at MB = 8) to land
  // from L2.  The sched_barriers pin the issue points: left alone, hipcc sinks each load
  // next to its first use and exposes a vmcnt(0) stall every ~16 MFMAs.
  float4 w1[G], w2[CT], w2r[NGA], bv;
  auto load_w1 = [&](int c) {
#pragma unroll
    for (int g = 0; g < G; ++g) w1[g] = W1q[((size_t)c * G + g) * 64];
    bv = b1q[c * 4];
  };
  auto load_w2 = [&](int c) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) w2[ct] = W2q[((size_t)c * CTP + ct) * 64];
    if (NG > 0) {
#pragma unroll
      for (int g = 0; g < NGA; ++g) w2r[g] = W2rq[((size_t)c * NGA + g) * 64];
    }
  };
  load_w1(0);
  load_w2(0);
  // De-phase the two waves that share a SIMD (they come from two workgroups that start together and
  // run the same instruction stream, so without this they also stall together): the wave in the odd
  // hardware wave slot starts its main loop `stagger` x 64 cycles late.
  if (stagger > 0) {
    const unsigned hwid = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID[3:0] = wave slot in the SIMD
    if (hwid & 1)
      for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }
  for (int ci = 0; ci < nchunk; ++ci) {
    const int nx = (ci + 1 < nchunk) ? ci + 1 : ci;  // clamped: the last prefetch is a harmless re-read
    // GEMM1: H^T chunk (16 hidden x 16*MB rows), K = D; bias is the initial accumulator
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
    load_w1(nx);
    __builtin_amdgcn_sched_barrier(0);
    // relu as one v_med3_f32 (x, 0, +inf) per element
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[mb][r] = __builtin_amdgcn_fmed3f(h[mb][r], 0.f, __builtin_inff());
    // GEMM2: Y^T += W2[:, chunk] H^T chunk ; accumulator register r is the k-step
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float4 q = w2[ct];
        const float a = r == 0 ? q.x : r == 1 ? q.y : r == 2 ? q.z : q.w;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) yacc[ct][mb] = mfma16(a, h[mb][r], yacc[ct][mb]);
      }
    if (NG > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < NGA; ++g) {
          const float4 q = w2r[g];
          const float a = r == 0 ? q.x : r == 1 ? q.y : r == 2 ? q.z : q.w;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            yrem[g][mb] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, h[mb][r], yrem[g][mb], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    load_w2(nx);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- deterministic cross-wave reduction: wave 0 adds its partial into the LDS X tile
  // (= the residual), waves 1..3 park theirs in red[]; one barrier; every thread then sums
  // x + p0 (already in xs) + p1 + p2 + p3 + b2 in that fixed order.
  if (NG > 0) {
    // add the four lane-quarter partials: afterwards every quarter holds the full sums
    // yrem[g][mb][i] = Y^T[c0 + 4g + i][16 mb + (lane & 15)]
#pragma unroll
    for (int g = 0; g < NGA; ++g)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = yrem[g][mb][i];
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          yrem[g][mb][i] = v;
        }
  }
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
    if (NG > 0 && (lane >> 4) < NGA) {  // quarter q handles remainder group g = q
      const int g = lane >> 4;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = 0.f;
#pragma unroll
          for (int gg = 0; gg < NGA; ++gg) v = (gg == g) ? yrem[gg][mb][i] : v;
          xs[(16 * mb + (lane & 15)) * S + 16 * CT + 4 * g + i] += v;
        }
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
    if (NG > 0 && (lane >> 4) < NGA) {
      const int g = lane >> 4;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        float4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gg = 0; gg < NGA; ++gg)
          if (gg == g) v = float4{yrem[gg][mb][0], yrem[gg][mb][1], yrem[gg][mb][2], yrem[gg][mb][3]};
        *reinterpret_cast<float4*>(&rw[(16 * mb + (lane & 15)) * S2 + 16 * CT + 4 * g]) = v;
      }
    }
  }
  __syncthreads();

  // ---- + b2, LayerNorm2, coalesced store ----
  constexpr int TPR = 256 / R;  // threads per row (2..16), power of two
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
      v += b2[c];
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
      float dlt = vals[i] - mean;
      ss = fmaf(dlt, dlt, ss);
    }
  }
#pragma unroll
  for (int o = TPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
  if (m < M) {
#pragma unroll
    for (int i = 0; i < cdiv(D, TPR); ++i) {
      const int c = sub + i * TPR;
      if (c < D) Y[(size_t)m * D + c] = (vals[i] - mean) * rstd * gam[c] + bet[c];
    }
  }
}

int g_ffn_stagger = -1;     // x64 cycles of start delay for the odd wave slot of each SIMD; -1 = heuristic (ffd_tune "ffn_stagger")
int g_ffn_rem = 1;          // 1: remainder rows of GEMM2 on the 4x4x1 MFMA (ffd_tune "ffn_rem")
int g_ffn_mb_override = 0;  // 0 = heuristic; 1/2/4/8 forces the tile height (ffd_tune "ffn_mb")

template <int D>
static hipError_t launch_ffn_d(const float* X, const LayerWeights& w, float* Y, int M, int F, hipStream_t s) {
  // Tile height 16*MB rows.  MB = 4 keeps two workgroups (two waves per SIMD) resident per CU
  // and is the default once the grid fills the chip; smaller tiles for small batches.
  int mb = g_ffn_mb_override;
  if (mb != 1 && mb != 2 && mb != 4 && mb != 8) {
    const int target = 2 * 256;
    mb = cdiv(M, 64) >= target ? 4 : cdiv(M, 32) >= target ? 2 : 1;
  }
  // With two resident workgroups per CU and at least two rounds of tiles, de-phase the pair by about one
  // prologue + epilogue so that one workgroup's non-MFMA phases run under the other's main loop
  // (measured 469 -> 457 us on the 95744 x 72 x 2048 shape, tools/sweep_stagger.py).
  const int stagger = g_ffn_stagger >= 0 ? g_ffn_stagger : (mb == 4 && cdiv(M, 64) >= 4 * 256) ? 11 * D : 0;
  dim3 block(256);
#define FFD_LAUNCH_FFN(MBV)                                                                                       \
  do {                                                                                                            \
    if (g_ffn_rem && MBV == 4 && D >= 16 && w2rem_groups(D) > 0)                                                                       \
      hipLaunchKernelGGL((k_ffn_ln<D, MBV, true>), dim3(cdiv(M, 16 * MBV)), block, 0, s, X, w.w1p, w.b1, w.w2p,   \
                         w.w2r, w.b2, w.n2w, w.n2b, Y, M, F, stagger);                                             \
    else                                                                                                          \
      hipLaunchKernelGGL((k_ffn_ln<D, MBV, false>), dim3(cdiv(M, 16 * MBV)), block, 0, s, X, w.w1p, w.b1, w.w2p,  \
                         w.w2r, w.b2, w.n2w, w.n2b, Y, M, F, stagger);                                             \
  } while (0)
  switch (mb) {
    case 8: FFD_LAUNCH_FFN(8); break;
    case 4: FFD_LAUNCH_FFN(4); break;
    case 2: FFD_LAUNCH_FFN(2); break;
    default: FFD_LAUNCH_FFN(1); break;
  }
#undef FFD_LAUNCH_FFN
  return hipGetLastError();
}

hipError_t launch_ffn_ln(const float* X, const LayerWeights& w, float* Y, int M, int D, int F, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  if (F % 64 != 0) return hipErrorInvalidValue;
  switch (D) {
#define X(d) \
    case d: return launch_ffn_d<d>(X, w, Y, M, F, s);
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ffd
