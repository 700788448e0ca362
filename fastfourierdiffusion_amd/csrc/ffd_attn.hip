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
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv, const float* __restrict__ kt,
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
  if (active) {
    for (int idx = lane; idx < L * HD; idx += 64) {
      const int j = idx / HD, e = idx - j * HD;
      float kx, vx;
      if (j < n_own) {
        const float* row = qkv + ((size_t)b * L + j) * (3 * d) + h * HD + e;
        kx = row[d];
        vx = row[2 * d];
      } else {
        const size_t t = ((size_t)h * L + j) * HD + e;
        kx = kt[t];
        vx = vt[t];
      }
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
    const float* row = qkv + ((size_t)b * L + l) * (3 * d) + h * HD;
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

template <int HD, int QPL>
static hipError_t launch_attn_t(const float* qkv, const float* kt, const float* vt, float* out, int B, int L, int H,
                                int n_own, hipStream_t s) {
  constexpr int KVS = KvStride<HD>::value;
  const size_t per_wave = (size_t)L * KVS * sizeof(float);
  int wpb = 4;
  while (wpb > 1 && per_wave * wpb > 48 * 1024) wpb >>= 1;
  const int pairs = B * H;
  hipLaunchKernelGGL((k_attention<HD, QPL>), dim3(cdiv(pairs, wpb)), dim3(64 * wpb), per_wave * wpb, s, qkv, kt, vt,
                     out, B, L, H, n_own);
  return hipGetLastError();
}

template <int HD>
static hipError_t launch_attn_hd(const float* qkv, const float* kt, const float* vt, float* out, int B, int L, int H,
                                 int n_own, hipStream_t s) {
  const int qpl = cdiv(L, 64);
  switch (qpl) {
    case 1: return launch_attn_t<HD, 1>(qkv, kt, vt, out, B, L, H, n_own, s);
    case 2: return launch_attn_t<HD, 2>(qkv, kt, vt, out, B, L, H, n_own, s);
    case 3: return launch_attn_t<HD, 3>(qkv, kt, vt, out, B, L, H, n_own, s);
    case 4: return launch_attn_t<HD, 4>(qkv, kt, vt, out, B, L, H, n_own, s);
    case 5:
    case 6: return launch_attn_t<HD, 6>(qkv, kt, vt, out, B, L, H, n_own, s);
    case 7:
    case 8: return launch_attn_t<HD, 8>(qkv, kt, vt, out, B, L, H, n_own, s);
    default: return hipErrorInvalidValue;  // L > 512
  }
}

hipError_t launch_attention(const float* qkv, const float* kt, const float* vt, float* out, int B, int L, int H,
                            int hd, int n_own, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  switch (hd) {
    case 4: return launch_attn_hd<4>(qkv, kt, vt, out, B, L, H, n_own, s);
    case 5: return launch_attn_hd<5>(qkv, kt, vt, out, B, L, H, n_own, s);
    case 6: return launch_attn_hd<6>(qkv, kt, vt, out, B, L, H, n_own, s);
    case 8: return launch_attn_hd<8>(qkv, kt, vt, out, B, L, H, n_own, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ffd
