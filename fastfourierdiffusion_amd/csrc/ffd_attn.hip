// Multi-head self-attention for tiny heads (head_dim 2..8; 6 on the default model) over head-major q / k / v in HBM:
// the two-kernel fallback of the fused in-projection + attention kernel (ffd_qkvattn.hip), used for shapes that kernel
// has no instance for and by ffd_tune("attn_fused", 0).  cached_transformer.py:309-311: softmax(q k^T / sqrt(hd)) v
// per (sample, head).
//
// E2-CRF modes (cached_transformer.py:237-305): keys l < n_own come from the sample's own K/V projections, keys
// l >= n_own from the shared (H, L, hd) tables -- n_own = L is the standard layer, 0 the pure-cache step.
#include "ffd_internal.h"

namespace ffd {

// ---------------------------------------------------------------------------
// Q K^T on the matrix cores, softmax + P V on the vector ALU.
//   S^T tile (32 keys x 32 queries) = K_tile (32 x hd) . Qs_tile^T (hd x 32) is ceil(hd/2)
//   v_mfma_f32_32x32x2_f32 (no padding waste at hd = 6).  Its accumulator layout puts the
//   query on the lane (l & 31) and 16 keys on the registers
//       key = 32 kt + (r & 3) + 8 (r >> 2) + 4 (l >> 5),
//   so softmax over keys is an in-lane reduction and P V is hd FMAs per score with V rows
//   broadcast-read from LDS (address depends only on the lane half).  The two lane halves
//   keep separate online-softmax states for the same query and are merged once per
//   q-tile.  QG q-tiles share every V read.  K fragments of all key tiles stay in VGPRs.
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// (launch bound: waves per SIMD; capping the VGPRs at 128 / 168 gave +5..7 % over the unconstrained
// build; software-pipelining the MFMAs one key tile ahead cost more occupancy than it hid: -10 %)
template <int HD, int QG>
__global__ __launch_bounds__(256, (QG <= 2 ? 4 : 3)) void k_attention_mfma(const float* __restrict__ qg, const float* __restrict__ kg,
                                                        const float* __restrict__ vg, const float* __restrict__ kt,
                                                        const float* __restrict__ vt, float* __restrict__ out,
                                                        int B, int L, int H, int n_own) {
  constexpr int KST = (HD + 1) / 2;  // MFMA k-steps
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  const int pair = blockIdx.x;  // one workgroup = one (sample, head); its waves split the q-tile groups
  const int d = H * HD;
  const int b = pair / H, h = pair - b * H;
  const int KT = (L + 31) >> 5;
  const int Lp = KT * 32;
  // LDS image of the head, staged once by the whole workgroup: V rows [Lp][8] (zero padded), K^T [2*KST][Lp]
  float* vs = lds;
  float* kts = vs + (size_t)Lp * 8;
  const int half = lane >> 5, l31 = lane & 31;

  for (int idx = threadIdx.x; idx < Lp * (8 + 2 * KST); idx += blockDim.x) vs[idx] = 0.f;
  __syncthreads();
  const size_t slice = (size_t)pair * L * HD;
  {
    const float* kown = kg + slice;
    const float* vown = vg + slice;
    const float* ktab = kt ? kt + (size_t)h * L * HD : kown;
    const float* vtab = vt ? vt + (size_t)h * L * HD : vown;
    const int own_elems = n_own * HD;
#pragma unroll 4
    for (int idx = threadIdx.x; idx < L * HD; idx += blockDim.x) {
      const int j = idx / HD, e = idx - j * HD;
      const bool own = idx < own_elems;
      const float kx = (own ? kown : ktab)[idx];
      const float vx = (own ? vown : vtab)[idx];
      vs[j * 8 + e] = vx;
      kts[e * Lp + j] = kx;
    }
  }
  __syncthreads();

  const float c = 1.4426950408889634f / sqrtf((float)HD);
  const int QT = KT;
  for (int qt0 = wave * QG; qt0 < QT; qt0 += nwaves * QG) {
    float qf[QG][KST], m[QG], lsum[QG], acc[QG][HD];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      int q = 32 * (qt0 + g) + l31;
      if (q >= L) q = L - 1;
#pragma unroll
      for (int s = 0; s < KST; ++s) {
        const int e = 2 * s + half;
        qf[g][s] = (e < HD) ? qg[slice + (size_t)q * HD + e] * c : 0.f;
      }
      m[g] = -INFINITY, lsum[g] = 0.f;
#pragma unroll
      for (int e = 0; e < HD; ++e) acc[g][e] = 0.f;
    }
#pragma unroll 1
    for (int t = 0; t < KT; ++t) {
      float kf[KST];
#pragma unroll
      for (int s = 0; s < KST; ++s) kf[s] = kts[(2 * s + half) * Lp + 32 * t + l31];
      f32x16 sc[QG];
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KST; ++s) z = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[g][s], z, 0, 0, 0);
        sc[g] = z;
      }
      const int kbase = 32 * t + 4 * half;
      if (32 * t + 32 > L) {  // last, ragged key tile: mask keys >= L
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool dead = kbase + (r & 3) + 8 * (r >> 2) >= L;
#pragma unroll
          for (int g = 0; g < QG; ++g) sc[g][r] = dead ? -INFINITY : sc[g][r];
        }
      }
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        float bm = sc[g][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) bm = fmaxf(bm, sc[g][r]);
        const float mnew = fmaxf(m[g], bm);
        const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
        const float corr = __builtin_amdgcn_exp2f(m[g] - msafe);
        m[g] = mnew;
        float l = lsum[g] * corr;
#pragma unroll
        for (int e = 0; e < HD; ++e) acc[g][e] *= corr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(sc[g][r] - msafe);
          sc[g][r] = p;
          l += p;
        }
        lsum[g] = l;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vr = vs + (size_t)(kbase + (r & 3) + 8 * (r >> 2)) * 8;
        float vv[8];
        const float4 v0 = *reinterpret_cast<const float4*>(vr);
        vv[0] = v0.x, vv[1] = v0.y, vv[2] = v0.z, vv[3] = v0.w;
        if (HD > 4) {
          const float4 v1 = *reinterpret_cast<const float4*>(vr + 4);
          vv[4] = v1.x, vv[5] = v1.y, vv[6] = v1.z, vv[7] = v1.w;
        }
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
          for (int e = 0; e < HD; ++e) acc[g][e] = fmaf(sc[g][r], vv[e], acc[g][e]);
      }
    }
    // ---- merge the two lane halves (same query, disjoint keys) and store ----
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      const float mo = __shfl_xor(m[g], 32);
      const float mm = fmaxf(m[g], mo);
      const float f = (m[g] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m[g] - mm);
      float l = lsum[g] * f;
      l += __shfl_xor(l, 32);
      const float inv = 1.0f / l;
      const int q = 32 * (qt0 + g) + l31;
      float o[HD];
#pragma unroll
      for (int e = 0; e < HD; ++e) {
        float a = acc[g][e] * f;
        a += __shfl_xor(a, 32);
        o[e] = a * inv;
      }
      if (half == 0 && q < L && qt0 + g < QT) {
        float* orow = out + ((size_t)b * L + q) * d + h * HD;
#pragma unroll
        for (int e = 0; e < HD; ++e) orow[e] = o[e];
      }
    }
  }
}

template <int HD, int QG>
static hipError_t launch_attn_mfma_t(const float* q, const float* k, const float* v, const float* kt,
                                     const float* vt, float* out, int B, int L, int H, int n_own, hipStream_t s) {
  constexpr int KST = (HD + 1) / 2;
  const int KT = (L + 31) / 32;
  const size_t lds = (size_t)KT * 32 * (8 + 2 * KST) * sizeof(float);
  int nwaves = cdiv(KT, QG);  // one wave per q-tile group, at most 4 (further groups are looped)
  if (nwaves > 4) nwaves = 4;
  hipLaunchKernelGGL((k_attention_mfma<HD, QG>), dim3(B * H), dim3(64 * nwaves), lds, s, q, k, v, kt, vt, out, B, L, H,
                     n_own);
  return hipGetLastError();
}

thread_local int g_attn_qg = 0;  // 0 heuristic; 1/2/3 force the q-tile group size (ffd_tune "attn_qg")

template <int HD>
static hipError_t launch_attn_mfma_hd(const float* q, const float* k, const float* v, const float* kt,
                                      const float* vt, float* out, int B, int L, int H, int n_own, hipStream_t s) {
  const int QT = (L + 31) / 32;
  if (g_attn_qg == 1 || QT == 1) return launch_attn_mfma_t<HD, 1>(q, k, v, kt, vt, out, B, L, H, n_own, s);
  if (g_attn_qg == 2) return launch_attn_mfma_t<HD, 2>(q, k, v, kt, vt, out, B, L, H, n_own, s);
  if (g_attn_qg == 3) return launch_attn_mfma_t<HD, 3>(q, k, v, kt, vt, out, B, L, H, n_own, s);
  if (QT % 3 == 0) return launch_attn_mfma_t<HD, 3>(q, k, v, kt, vt, out, B, L, H, n_own, s);
  return launch_attn_mfma_t<HD, 2>(q, k, v, kt, vt, out, B, L, H, n_own, s);
}

hipError_t launch_attention(const float* qkv, const float* k, const float* v, const float* kt, const float* vt,
                            float* out, int B, int L, int H, int hd, int n_own, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  switch (hd) {
#define X(h) \
    case h: return launch_attn_mfma_hd<h>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
    FFD_HD_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ffd
