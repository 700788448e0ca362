// Multi-head self-attention for tiny heads (head_dim 4..8; 6 on the default model).
// cached_transformer.py:309-311 : softmax(q k^T / sqrt(hd)) v, per (sample, head).
//
// With hd = 6 the MFMA shapes waste 25 % (QK^T, K padded to 8) to 60-80 % (PV, N
// padded to 16/32) of their issue slots, so this kernel keeps the whole thing on the
// vector ALU with queries on lanes: one wave owns one (sample, head); each lane
// carries QPL queries (q, running max, running sum, hd accumulators in VGPRs); the
// head's K/V rows are staged once in LDS and broadcast-read (one ds_read_b128 serves
// 64 lanes x QPL queries); softmax is online over blocks of 4 keys with the
// 1/sqrt(hd)*log2(e) scale folded into q so p = exp2(s - m) is a single v_exp_f32.
//
// E2-CRF modes (cached_transformer.py:237-305): keys l < n_own come from the
// sample's own K/V projections (qkv buffer), keys l >= n_own from the shared
// (H, L, hd) tables -- n_own = L is the standard layer, 0 the pure-cache step.
#include <type_traits>

#include "ffd_internal.h"

namespace ffd {

template <int HD>
struct KvStride {
  static constexpr int value = ((2 * HD + 3) / 4) * 4;
};

template <int HD, int QPL, int NK>
__device__ __forceinline__ void attend_keys(const float* __restrict__ kv, float (&q)[QPL][HD], float (&acc)[QPL][HD],
                                            float (&mrun)[QPL], float (&lrun)[QPL]) {
  constexpr int KVS = KvStride<HD>::value;
  float kk[NK][HD], vv[NK][HD];
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    float row[KVS];
#pragma unroll
    for (int i = 0; i < KVS / 4; ++i) {
      float4 t = *reinterpret_cast<const float4*>(kv + j * KVS + 4 * i);  // wave-uniform address: LDS broadcast
      row[4 * i] = t.x, row[4 * i + 1] = t.y, row[4 * i + 2] = t.z, row[4 * i + 3] = t.w;
    }
#pragma unroll
    for (int e = 0; e < HD; ++e) kk[j][e] = row[e], vv[j][e] = row[HD + e];
  }
#pragma unroll
  for (int qi = 0; qi < QPL; ++qi) {
    float s[NK];
    float bm = -INFINITY;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      float t = q[qi][0] * kk[j][0];
#pragma unroll
      for (int e = 1; e < HD; ++e) t = fmaf(q[qi][e], kk[j][e], t);
      s[j] = t;
      bm = fmaxf(bm, t);
    }
    const float mnew = fmaxf(mrun[qi], bm);
    const float corr = __builtin_amdgcn_exp2f(mrun[qi] - mnew);
    mrun[qi] = mnew;
    float l = lrun[qi] * corr;
#pragma unroll
    for (int e = 0; e < HD; ++e) acc[qi][e] *= corr;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const float p = __builtin_amdgcn_exp2f(s[j] - mnew);
      l += p;
#pragma unroll
      for (int e = 0; e < HD; ++e) acc[qi][e] = fmaf(p, vv[j][e], acc[qi][e]);
    }
    lrun[qi] = l;
  }
}

template <int HD, int QPL>
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qg, const float* __restrict__ kg,
                                                   const float* __restrict__ vg, const float* __restrict__ kt,
                                                   const float* __restrict__ vt, float* __restrict__ out, int B,
                                                   int L, int H, int n_own) {
  constexpr int KVS = KvStride<HD>::value;
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int pair = blockIdx.x * wpb + wave;  // (b, h)
  const int d = H * HD;
  const bool active = pair < B * H;
  const int b = active ? pair / H : 0, h = active ? pair % H : 0;
  float* kv = lds + (size_t)wave * L * KVS;

  // ---- stage K/V rows of this head: [key][k0..k(hd-1), v0..v(hd-1), pad] ----
  // q/k/v are head-major (B,H,L,hd): the (b,h) slice is one contiguous run, as are the tables.
  const size_t slice = (size_t)pair * L * HD;
  if (active) {
    const float* kown = kg + slice;
    const float* vown = vg + slice;
    const float* ktab = kt ? kt + (size_t)h * L * HD : kown;
    const float* vtab = vt ? vt + (size_t)h * L * HD : vown;
    const int own_elems = n_own * HD;
#pragma unroll 4
    for (int idx = lane; idx < L * HD; idx += 64) {
      const int j = idx / HD, e = idx - j * HD;
      const bool own = idx < own_elems;
      const float kx = (own ? kown : ktab)[idx];
      const float vx = (own ? vown : vtab)[idx];
      kv[j * KVS + e] = kx;
      kv[j * KVS + HD + e] = vx;
    }
  }
  __syncthreads();
  if (!active) return;

  // ---- per-lane queries, pre-scaled by log2(e)/sqrt(hd) ----
  const float c = 1.4426950408889634f / sqrtf((float)HD);
  float q[QPL][HD], acc[QPL][HD], mrun[QPL], lrun[QPL];
#pragma unroll
  for (int qi = 0; qi < QPL; ++qi) {
    int l = qi * 64 + lane;
    if (l >= L) l = L - 1;
    const float* row = qg + slice + (size_t)l * HD;
#pragma unroll
    for (int e = 0; e < HD; ++e) q[qi][e] = row[e] * c, acc[qi][e] = 0.f;
    mrun[qi] = -INFINITY, lrun[qi] = 0.f;
  }

  int j = 0;
  for (; j + 4 <= L; j += 4) attend_keys<HD, QPL, 4>(kv + j * KVS, q, acc, mrun, lrun);
  for (; j < L; ++j) attend_keys<HD, QPL, 1>(kv + j * KVS, q, acc, mrun, lrun);

#pragma unroll
  for (int qi = 0; qi < QPL; ++qi) {
    const int l = qi * 64 + lane;
    if (l < L) {
      const float inv = 1.0f / lrun[qi];
      float* o = out + ((size_t)b * L + l) * d + h * HD;
#pragma unroll
      for (int e = 0; e < HD; ++e) o[e] = acc[qi][e] * inv;
    }
  }
}

// ---------------------------------------------------------------------------
// Hybrid form (default): Q K^T on the matrix cores, softmax + P V on the vector ALU.
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

// ---------------------------------------------------------------------------
// k_attention_pk (attn_impl = 3): same tiling as k_attention_mfma, with the VALU work per (query, key)
// pair cut from ~11 issue slots to ~6:
//  * the running-max subtraction rides on the score MFMA: dimension HD of the contraction is
//    (K side) 1, (Q side) -m_ref[q], so the matrix core returns s - m_ref directly.  m_ref is a *stale*
//    reference, refreshed only when a tile's maximum exceeds it by more than 2^8 (and on the first tile),
//    which is exact -- any common factor 2^-m_ref cancels between numerator and denominator;
//  * P.V and the row sum use packed fp32 (v_pk_fma_f32 / v_pk_add_f32): two output dims per instruction;
//  * the block maximum uses v_max3_f32.
// Both lane halves of a query share m_ref, so their partial sums add without a rescale at the end.
// ---------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int HD, int QG>
__global__ __launch_bounds__(256, (QG <= 2 ? 4 : 3)) void k_attention_pk(const float* __restrict__ qg, const float* __restrict__ kg,
                                                        const float* __restrict__ vg, const float* __restrict__ kt,
                                                        const float* __restrict__ vt, float* __restrict__ out,
                                                        int B, int L, int H, int n_own) {
  constexpr int KST = (HD + 1) / 2;   // k-steps of K^T staged in LDS (real dims)
  constexpr int KSX = (HD + 2) / 2;   // k-steps issued: dims 0..HD, dim HD carrying the max reference
  constexpr int SX = HD / 2;          // the k-step that holds dim HD ...
  constexpr int HX = HD & 1;          // ... in this lane half
  constexpr int HP = (HD + 1) / 2;    // packed output pairs
  constexpr float T = 8.0f;
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  const int pair = blockIdx.x;
  const int d = H * HD;
  const int b = pair / H, h = pair - b * H;
  const int KT = (L + 31) >> 5;
  const int Lp = KT * 32;
  float* vs = lds;
  float* kts = vs + (size_t)Lp * 8;
  const int half = lane >> 5, l31 = lane & 31;

  // This wave's first query group is requested before the K/V staging so both latencies overlap.
  const size_t slice = (size_t)pair * L * HD;
  auto load_q = [&](int qt0, float(&dst)[QG][KSX]) {
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      int q = 32 * (qt0 + g) + l31;
      if (q >= L) q = L - 1;
#pragma unroll
      for (int s = 0; s < KSX; ++s) {
        const int e = 2 * s + half;
        dst[g][s] = (e < HD) ? qg[slice + (size_t)q * HD + e] : 0.f;
      }
    }
  };
  float qnext[QG][KSX];
  load_q(wave * QG < KT ? wave * QG : 0, qnext);
  __builtin_amdgcn_sched_barrier(0);

  // Stage the head once: one thread per key row, all of a row's loads in flight together, zero padding
  // written in the same pass (rows >= L, V columns >= HD, the odd K^T pad row) -- a single barrier.
  {
    const float* kown = kg + slice;
    const float* vown = vg + slice;
    const float* ktab = kt ? kt + (size_t)h * L * HD : kown;
    const float* vtab = vt ? vt + (size_t)h * L * HD : vown;
#pragma unroll 2
    for (int j = threadIdx.x; j < Lp; j += blockDim.x) {
      float kx[2 * KST], vx[8];
#pragma unroll
      for (int e = 0; e < 2 * KST; ++e) kx[e] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) vx[e] = 0.f;
      if (j < L) {
        const float* kp = (j < n_own ? kown : ktab) + (size_t)j * HD;
        const float* vp = (j < n_own ? vown : vtab) + (size_t)j * HD;
        if constexpr (HD % 2 == 0) {  // rows are 8-byte aligned (slice and row sizes are multiples of 2 floats)
#pragma unroll
          for (int e = 0; e < HD; e += 2) {
            const float2 a = *reinterpret_cast<const float2*>(kp + e);
            const float2 c2 = *reinterpret_cast<const float2*>(vp + e);
            kx[e] = a.x, kx[e + 1] = a.y, vx[e] = c2.x, vx[e + 1] = c2.y;
          }
        } else {
#pragma unroll
          for (int e = 0; e < HD; ++e) kx[e] = kp[e], vx[e] = vp[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 2 * KST; ++e) kts[e * Lp + j] = kx[e];
      *reinterpret_cast<float4*>(vs + (size_t)j * 8) = float4{vx[0], vx[1], vx[2], vx[3]};
      *reinterpret_cast<float4*>(vs + (size_t)j * 8 + 4) = float4{vx[4], vx[5], vx[6], vx[7]};
    }
  }
  __syncthreads();

  const float c = 1.4426950408889634f / sqrtf((float)HD);
  const bool xlane = half == HX;  // lanes whose operand slot in k-step SX is dim HD
  constexpr int PF = 4;           // V rows in flight
  auto load_v = [&](int r, int kbase, f32x2(&dst)[4]) {
    const float* vr = vs + (size_t)(kbase + (r & 3) + 8 * (r >> 2)) * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(vr);
    dst[0] = f32x2{v0.x, v0.y}, dst[1] = f32x2{v0.z, v0.w};
    if (HD > 4) {
      const float4 v1 = *reinterpret_cast<const float4*>(vr + 4);
      dst[2] = f32x2{v1.x, v1.y}, dst[3] = f32x2{v1.z, v1.w};
    } else {
      dst[2] = f32x2{0.f, 0.f}, dst[3] = f32x2{0.f, 0.f};
    }
  };
  const int QT = KT;
  for (int qt0 = wave * QG; qt0 < QT; qt0 += nwaves * QG) {
    float qf[QG][KSX], mref[QG];
    f32x2 lsum[QG], acc[QG][HP];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
#pragma unroll
      for (int s = 0; s < KSX; ++s) qf[g][s] = qnext[g][s] * c;  // dim HD (the max reference) starts at -m_ref = 0
      mref[g] = 0.f;
      lsum[g] = f32x2{0.f, 0.f};
#pragma unroll
      for (int e = 0; e < HP; ++e) acc[g][e] = f32x2{0.f, 0.f};
    }
    if (qt0 + nwaves * QG < QT) load_q(qt0 + nwaves * QG, qnext);  // next group's queries under this one's work
#pragma unroll 1
    for (int t = 0; t < KT; ++t) {
      float kf[KSX];
#pragma unroll
      for (int s = 0; s < KSX; ++s) {
        const int e = 2 * s + half;
        kf[s] = (s < KST && (2 * s + 1 < HD || half == 0)) ? kts[(size_t)e * Lp + 32 * t + l31] : 0.f;
      }
      if (xlane) kf[SX] = 1.0f;  // the "ones" row that multiplies -m_ref
      f32x16 sc[QG];
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KSX; ++s) z = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[g][s], z, 0, 0, 0);
        sc[g] = z;
      }
      const int kbase = 32 * t + 4 * half;
      f32x2 vb[PF][4];
#pragma unroll
      for (int r = 0; r < PF; ++r) load_v(r, kbase, vb[r]);
      __builtin_amdgcn_sched_barrier(0);
      if (32 * t + 32 > L) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool dead = kbase + (r & 3) + 8 * (r >> 2) >= L;
#pragma unroll
          for (int g = 0; g < QG; ++g) sc[g][r] = dead ? -INFINITY : sc[g][r];
        }
      }
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        float bm = __builtin_fmaxf(__builtin_fmaxf(sc[g][0], sc[g][1]), sc[g][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) bm = __builtin_fmaxf(__builtin_fmaxf(bm, sc[g][r]), sc[g][r + 1]);
        bm = __builtin_fmaxf(bm, sc[g][15]);
        const float bmx = fmaxf(bm, __shfl_xor(bm, 32));  // both halves of a query move together
        if (t == 0 || bmx > T) {  // refresh the reference: new m_ref = old + bmx (tile 0 always has a live key)
          const float delta = bmx;
          mref[g] += delta;
          if (t != 0) {
            const float corr = __builtin_amdgcn_exp2f(-delta);
            lsum[g] *= corr;
#pragma unroll
            for (int e = 0; e < HP; ++e) acc[g][e] *= corr;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[g][r] -= delta;
          if (xlane) qf[g][SX] = -mref[g];
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(sc[g][r]);
          const float p1 = __builtin_amdgcn_exp2f(sc[g][r + 1]);
          sc[g][r] = p0;
          sc[g][r + 1] = p1;
          lsum[g] += f32x2{p0, p1};
        }
      }
      // P.V with the V rows (LDS broadcast reads) kept PF rows ahead of their use; the first PF rows were
      // requested before the softmax phase.  sched_barrier pins the requests where they are written.
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f32x2 vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = vb[r % PF][e];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
          const f32x2 p2 = f32x2{sc[g][r], sc[g][r]};
#pragma unroll
          for (int e = 0; e < HP; ++e) acc[g][e] = __builtin_elementwise_fma(p2, vv[e], acc[g][e]);
        }
        if (r + PF < 16) {
          load_v(r + PF, kbase, vb[r % PF]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- add the two lane halves (same query and reference, disjoint keys) and store ----
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      float l = lsum[g].x + lsum[g].y;
      l += __shfl_xor(l, 32);
      const float inv = 1.0f / l;
      const int q = 32 * (qt0 + g) + l31;
      float o[2 * HP];
#pragma unroll
      for (int e = 0; e < HP; ++e) {
        float a0 = acc[g][e].x, a1 = acc[g][e].y;
        a0 += __shfl_xor(a0, 32);
        a1 += __shfl_xor(a1, 32);
        o[2 * e] = a0 * inv, o[2 * e + 1] = a1 * inv;
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
static hipError_t launch_attn_pk_t(const float* q, const float* k, const float* v, const float* kt, const float* vt,
                                   float* out, int B, int L, int H, int n_own, hipStream_t s) {
  constexpr int KST = (HD + 1) / 2;
  const int KT = (L + 31) / 32;
  const size_t lds = (size_t)KT * 32 * (8 + 2 * KST) * sizeof(float);
  int nwaves = cdiv(KT, QG);
  if (nwaves > 4) nwaves = 4;
  hipLaunchKernelGGL((k_attention_pk<HD, QG>), dim3(B * H), dim3(64 * nwaves), lds, s, q, k, v, kt, vt, out, B, L, H,
                     n_own);
  return hipGetLastError();
}

int g_attn_qg = 0;  // 0 heuristic; 1/2/3 force the q-tile group size (ffd_tune "attn_qg")

template <int HD>
static hipError_t launch_attn_pk_hd(const float* q, const float* k, const float* v, const float* kt, const float* vt,
                                    float* out, int B, int L, int H, int n_own, hipStream_t s) {
  const int QT = (L + 31) / 32;
  if (g_attn_qg == 1 || QT == 1) return launch_attn_pk_t<HD, 1>(q, k, v, kt, vt, out, B, L, H, n_own, s);
  if (g_attn_qg == 2) return launch_attn_pk_t<HD, 2>(q, k, v, kt, vt, out, B, L, H, n_own, s);
  if (g_attn_qg == 3) return launch_attn_pk_t<HD, 3>(q, k, v, kt, vt, out, B, L, H, n_own, s);
  if (QT % 3 == 0) return launch_attn_pk_t<HD, 3>(q, k, v, kt, vt, out, B, L, H, n_own, s);
  return launch_attn_pk_t<HD, 2>(q, k, v, kt, vt, out, B, L, H, n_own, s);
}

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

// ---------------------------------------------------------------------------
// v3 (experiment, ffd_tune attn_impl=2): both products on the 16-block 4x4x1 matrix instruction.
// Measured: the instruction issues every 11.1 cycles (8 independent chains) with 40 cycles of
// dependent latency (tools/probes/bench_mfma4x4.hip), i.e. 0.6 cycles per (query,key) pair at
// best -- slower than the hybrid kernel's VALU-bound 0.42; kept for reference (113 us vs 98 us).
//
// v_mfma_f32_4x4x1_16b_f32 computes, for each of 16 lane blocks b, the 4x4 outer product
// D_b[i][j] += A[lane 4*bsel+i] * B[lane 4b+j]; with cbsz = 4 every block takes its A values
// from block `abid`.  With one query per lane (B operand = that lane's own scalar):
//   scores : A = K^T register (lane l holds K[k0+l][e]), abid = g  ->  D[i] = q_e * K[k0+4g+i][e]
//            summed over e by chaining: 4 scores (keys k0+4g..+3) per lane, hd instructions
//   P V    : A = V register (lane l holds V[k0+(l>>2)][l&3]), abid = kk, B = p[kk]
//            -> D[i] += p * V[k0+kk][i] : the lane's 4 output dims, one instruction per key
//            (a second register / instruction covers dims 4..7)
// so neither product pads to a 16- or 32-wide tile (hd = 6: 6 + 2 = 8 cycles-8 instructions
// per 4 / 1 keys), the P values never leave the lane, and the online-softmax state is one
// (m, l, O[hd]) per lane.  K^T / V rows of the (sample, head) are staged once per workgroup
// (ceil(L/64) waves = all query blocks of that head) and read with coalesced ds_read_b32.
// ---------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int HD>
__global__ __launch_bounds__(512) void k_attention_v3(const float* __restrict__ qg, const float* __restrict__ kg,
                                                      const float* __restrict__ vg, const float* __restrict__ kt,
                                                      const float* __restrict__ vt, float* __restrict__ out, int B,
                                                      int L, int H, int n_own) {
  constexpr bool HI = HD > 4;
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.x;  // (b, h)
  const int b = pair / H, h = pair - b * H;
  const int d = H * HD;
  const int SB = (L + 63) >> 6;  // 64-key super-blocks
  const int Lp = SB * 64;
  float* kts = lds;                  // K^T [HD][Lp]
  float* vlo = kts + HD * Lp;        // V dims 0..3   [Lp][4]
  float* vhi = vlo + 4 * Lp;         // V dims 4..7   [Lp][4]  (zero beyond HD)

  for (int idx = threadIdx.x; idx < Lp * (HD + 8); idx += blockDim.x) lds[idx] = 0.f;
  __syncthreads();
  const size_t slice = (size_t)pair * L * HD;
  {
    const float* kown = kg + slice;
    const float* vown = vg + slice;
    const float* ktab = kt ? kt + (size_t)h * L * HD : kown;
    const float* vtab = vt ? vt + (size_t)h * L * HD : vown;
    const int own_elems = n_own * HD;
    for (int idx = threadIdx.x; idx < L * HD; idx += blockDim.x) {
      const int j = idx / HD, e = idx - j * HD;
      const bool own = idx < own_elems;
      const float kx = (own ? kown : ktab)[idx];
      const float vx = (own ? vown : vtab)[idx];
      kts[e * Lp + j] = kx;
      if (e < 4) vlo[4 * j + e] = vx;
      else vhi[4 * j + (e - 4)] = vx;
    }
  }
  __syncthreads();

  const int q0 = 64 * wave;
  if (q0 >= L) return;
  const int qi = min(q0 + lane, L - 1);
  const float c = 1.4426950408889634f / sqrtf((float)HD);
  float q[HD];
#pragma unroll
  for (int e = 0; e < HD; ++e) q[e] = qg[slice + (size_t)qi * HD + e] * c;
  float m = -INFINITY, lsum = 0.f;
  f32x4 olo = {0.f, 0.f, 0.f, 0.f}, ohi = {0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int sb = 0; sb < SB; ++sb) {
    const int k0 = 64 * sb;
    float kr[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) kr[e] = kts[e * Lp + k0 + lane];
    float vl[4], vh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      vl[g] = vlo[4 * (k0 + 16 * g) + lane];
      vh[g] = HI ? vhi[4 * (k0 + 16 * g) + lane] : 0.f;
    }
    const bool ragged = k0 + 64 > L;
    static_for<0, 4>([&](auto gq_c) {
      constexpr int gq = decltype(gq_c)::value;
      // ---- 16 scores per lane: keys k0 + 16 gq + 4 gi + i
      f32x4 sc[4];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) sc[gi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < HD; ++e) {
        sc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(kr[e], q[e], sc[0], 4, 4 * gq + 0, 0);
        sc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(kr[e], q[e], sc[1], 4, 4 * gq + 1, 0);
        sc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(kr[e], q[e], sc[2], 4, 4 * gq + 2, 0);
        sc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(kr[e], q[e], sc[3], 4, 4 * gq + 3, 0);
      }
      if (ragged) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (k0 + 16 * gq + 4 * gi + i >= L) sc[gi][i] = -INFINITY;
      }
      // ---- online softmax on the 16 scores
      float bm = sc[0][0];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int i = 0; i < 4; ++i) bm = fmaxf(bm, sc[gi][i]);
      const float mnew = fmaxf(m, bm);
      const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
      const float corr = __builtin_amdgcn_exp2f(m - msafe);
      m = mnew;
      float l = lsum * corr;
#pragma unroll
      for (int i = 0; i < 4; ++i) olo[i] *= corr, ohi[i] *= corr;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = __builtin_amdgcn_exp2f(sc[gi][i] - msafe);
          sc[gi][i] = p;
          l += p;
        }
      lsum = l;
      // ---- O += p_k * V[k] : one instruction per key (two when hd > 4)
      static_for<0, 16>([&](auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        const float p = sc[kk >> 2][kk & 3];
        olo = __builtin_amdgcn_mfma_f32_4x4x1f32(vl[gq], p, olo, 4, kk, 0);
        if (HI) ohi = __builtin_amdgcn_mfma_f32_4x4x1f32(vh[gq], p, ohi, 4, kk, 0);
      });
    });
  }

  if (q0 + lane < L) {
    const float inv = 1.0f / lsum;
    float* orow = out + ((size_t)b * L + q0 + lane) * d + h * HD;
#pragma unroll
    for (int e = 0; e < HD; ++e) orow[e] = (e < 4 ? olo[e] : ohi[e - 4]) * inv;
  }
}

template <int HD>
static hipError_t launch_attn_v3(const float* q, const float* k, const float* v, const float* kt, const float* vt,
                                 float* out, int B, int L, int H, int n_own, hipStream_t s) {
  const int SB = (L + 63) / 64;
  const size_t lds = (size_t)SB * 64 * (HD + 8) * sizeof(float);
  hipLaunchKernelGGL((k_attention_v3<HD>), dim3(B * H), dim3(64 * SB), lds, s, q, k, v, kt, vt, out, B, L, H, n_own);
  return hipGetLastError();
}

int g_attn_impl = 0;  // 0 = 32x32x2 QK^T + VALU softmax/PV (default, fastest measured), 1 = all VALU, 2 = 4x4x1-MFMA products

template <int HD, int QPL>
static hipError_t launch_attn_t(const float* q, const float* k, const float* v, const float* kt, const float* vt,
                                float* out, int B, int L, int H, int n_own, hipStream_t s) {
  constexpr int KVS = KvStride<HD>::value;
  const size_t per_wave = (size_t)L * KVS * sizeof(float);
  int wpb = 4;
  while (wpb > 1 && per_wave * wpb > 48 * 1024) wpb >>= 1;
  const int pairs = B * H;
  hipLaunchKernelGGL((k_attention<HD, QPL>), dim3(cdiv(pairs, wpb)), dim3(64 * wpb), per_wave * wpb, s, q, k, v, kt,
                     vt, out, B, L, H, n_own);
  return hipGetLastError();
}

template <int HD>
static hipError_t launch_attn_hd(const float* qkv, const float* k, const float* v, const float* kt, const float* vt,
                                 float* out, int B, int L, int H, int n_own, hipStream_t s) {
  const int qpl = cdiv(L, 64);
  switch (qpl) {
    case 1: return launch_attn_t<HD, 1>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
    case 2: return launch_attn_t<HD, 2>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
    case 3: return launch_attn_t<HD, 3>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
    case 4: return launch_attn_t<HD, 4>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
    case 5:
    case 6: return launch_attn_t<HD, 6>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
    case 7:
    case 8: return launch_attn_t<HD, 8>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
    default: return hipErrorInvalidValue;  // L > 512
  }
}

hipError_t launch_attention(const float* qkv, const float* k, const float* v, const float* kt, const float* vt,
                            float* out, int B, int L, int H, int hd, int n_own, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  if (g_attn_impl == 2) {
    switch (hd) {
#define X(h) \
      case h: return launch_attn_v3<h>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
      FFD_HD_LIST(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  if (g_attn_impl == 3) {
    switch (hd) {
#define X(h) \
      case h: return launch_attn_pk_hd<h>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
      FFD_HD_LIST(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  if (g_attn_impl == 0) {
    switch (hd) {
#define X(h) \
      case h: return launch_attn_mfma_hd<h>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
      FFD_HD_LIST(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  switch (hd) {
#define X(h) \
    case h: return launch_attn_hd<h>(qkv, k, v, kt, vt, out, B, L, H, n_own, s);
    FFD_HD_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ffd
