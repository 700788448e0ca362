// Generic dense layer  Y[M,N] = act(X[M,K] . W[N,K]^T + b[N] (+ b2[N])) (+ R[M,N])
// for the MLP score backbone (MLPScoreModule, score_models.py:363-440): embedder (L*C -> d), the per-layer
// d -> d_mlp -> d blocks with their residual, unembedder (d -> L*C).  Arbitrary M, N, K: one wave owns one
// 16 x 16 output tile on v_mfma_f32_16x16x4_f32 (exact fp32); K is walked 16 at a time with each lane holding
// 4 consecutive k of its row (float4 when K % 4 == 0) -- the MFMA's k index is then a permutation of the 16 k
// values, identical for A and B, which the sum does not see.  Rows / columns / k beyond the edges read as zero.
#include "ffd_internal.h"

namespace ffd {

template <bool VEC4>
__device__ __forceinline__ float4 load_k4(const float* __restrict__ base, int row, int nrows, int k, int K) {
  float4 v{0.f, 0.f, 0.f, 0.f};
  if (row < nrows) {
    const float* p = base + (size_t)row * K + k;
    if (VEC4) {
      if (k < K) v = *reinterpret_cast<const float4*>(p);  // K % 4 == 0 and k % 4 == 0: whole or nothing
    } else {
      if (k + 0 < K) v.x = p[0];
      if (k + 1 < K) v.y = p[1];
      if (k + 2 < K) v.z = p[2];
      if (k + 3 < K) v.w = p[3];
    }
  }
  return v;
}

template <bool VEC4, bool RELU>
__global__ __launch_bounds__(256) void k_dense(const float* __restrict__ X, const float* __restrict__ W,
                                               const float* __restrict__ b, const float* __restrict__ b2,
                                               const float* __restrict__ R, float* __restrict__ Y, int M, int N,
                                               int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.y * 16;
  const int n0 = (blockIdx.x * 4 + wave) * 16;
  if (n0 >= N) return;
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc{0.f, 0.f, 0.f, 0.f};
  float4 a = load_k4<VEC4>(X, m0 + r, M, 4 * q, K);
  float4 w = load_k4<VEC4>(W, n0 + r, N, 4 * q, K);
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float4 an = load_k4<VEC4>(X, m0 + r, M, k0 + 16 + 4 * q, K);  // next step's operands under the MFMAs
    const float4 wn = load_k4<VEC4>(W, n0 + r, N, k0 + 16 + 4 * q, K);
    acc = mfma16(a.x, w.x, acc);
    acc = mfma16(a.y, w.y, acc);
    acc = mfma16(a.z, w.z, acc);
    acc = mfma16(a.w, w.w, acc);
    a = an;
    w = wn;
  }
  // D: lane holds rows m0 + 4q + i (i = 0..3) of column n0 + r
  const int n = n0 + r;
  if (n < N) {
    float bias = b ? b[n] : 0.f;
    if (b2) bias += b2[n];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 4 * q + i;
      if (m < M) {
        float v = acc[i] + bias;
        if (RELU) v = fmaxf(v, 0.f);
        if (R) v += R[(size_t)m * N + n];
        Y[(size_t)m * N + n] = v;
      }
    }
  }
}

hipError_t launch_dense(const float* X, const float* W, const float* b, const float* b2, const float* R, float* Y,
                        int M, int N, int K, int relu, hipStream_t s) {
  if (M <= 0 || N <= 0) return hipSuccess;
  if (K < 1) return hipErrorInvalidValue;
  dim3 grid(cdiv(N, 64), cdiv(M, 16)), block(256);
  const bool vec = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) % 16 == 0);
  if (vec) {
    if (relu) hipLaunchKernelGGL((k_dense<true, true>), grid, block, 0, s, X, W, b, b2, R, Y, M, N, K);
    else hipLaunchKernelGGL((k_dense<true, false>), grid, block, 0, s, X, W, b, b2, R, Y, M, N, K);
  } else {
    if (relu) hipLaunchKernelGGL((k_dense<false, true>), grid, block, 0, s, X, W, b, b2, R, Y, M, N, K);
    else hipLaunchKernelGGL((k_dense<false, false>), grid, block, 0, s, X, W, b, b2, R, Y, M, N, K);
  }
  return hipGetLastError();
}

}  // namespace ffd
