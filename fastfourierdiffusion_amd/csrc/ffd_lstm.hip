// Residual LSTM layer of LSTMScoreModule (score_models.py:472-477, 502-504):
//     x <- x + LSTM(x)[0]      nn.LSTM(d, d, batch_first), zero initial state,
//     gates (i, f, g, o);  c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c').
// The input projection gx = x W_ih^T + (b_ih + b_hh) for all L positions is one
// MFMA GEMM (k_linear); this kernel runs the L-step recurrence.  The recurrence is
// latency-bound (L sequential cell steps), so a workgroup owns BT samples for the
// whole sequence: thread j keeps row j of W_hh (d floats) in VGPRs for all L steps,
// h lives in LDS and is broadcast-read, c lives in the registers of the cell threads.
#include "ffd_internal.h"

namespace ffd {

constexpr int LSTM_BT = 2;

// sigmoid / tanh from one v_exp_f32 + one v_rcp_f32 each (absolute error ~1e-7, far inside the
// parity tolerance; the library expf / tanhf cost ~10x the instructions on the critical path)
__device__ __forceinline__ float sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
  // 1 - 2/(exp(2x)+1); exp2 overflow -> rcp(inf) = 0 -> 1, underflow -> 1 - 2 = -1
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

template <int D>
__global__ __launch_bounds__(320) void k_lstm_layer(float* __restrict__ x, const float* __restrict__ gx,
                                                    const float* __restrict__ whh, int B, int L) {
  constexpr int G4 = 4 * D;
  __shared__ __align__(16) float hbuf[LSTM_BT][D];
  __shared__ float gates[LSTM_BT][G4];
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * LSTM_BT;
  const bool gate_thread = tid < G4;

  float w[D];
  if (gate_thread) {
#pragma unroll
    for (int k = 0; k < D; ++k) w[k] = whh[(size_t)tid * D + k];
  }
  // cell threads: (bt, e)
  const bool cell_thread = tid < LSTM_BT * D;
  const int cbt = tid / D, ce = tid - cbt * D;
  const bool cell_valid = cell_thread && (b0 + cbt) < B;
  float c = 0.f;
  for (int i = tid; i < LSTM_BT * D; i += blockDim.x) (&hbuf[0][0])[i] = 0.f;
  __syncthreads();

  float gnext[LSTM_BT];
#pragma unroll
  for (int bt = 0; bt < LSTM_BT; ++bt) {
    const int b = min(b0 + bt, B - 1);
    gnext[bt] = gate_thread ? gx[((size_t)b * L) * G4 + tid] : 0.f;
  }
  for (int s = 0; s < L; ++s) {
    float g[LSTM_BT];
#pragma unroll
    for (int bt = 0; bt < LSTM_BT; ++bt) g[bt] = gnext[bt];
    if (s + 1 < L) {
#pragma unroll
      for (int bt = 0; bt < LSTM_BT; ++bt) {
        const int b = min(b0 + bt, B - 1);
        gnext[bt] = gate_thread ? gx[((size_t)b * L + s + 1) * G4 + tid] : 0.f;
      }
    }
    if (gate_thread) {
      // four partial sums per sample: the recurrence is latency-bound, so the 72-long dot
      // product must not be one dependent FMA chain
      float p1[LSTM_BT], p2[LSTM_BT], p3[LSTM_BT];
#pragma unroll
      for (int bt = 0; bt < LSTM_BT; ++bt) p1[bt] = p2[bt] = p3[bt] = 0.f;
#pragma unroll
      for (int k = 0; k < D; k += 4) {
#pragma unroll
        for (int bt = 0; bt < LSTM_BT; ++bt) {
          const float4 hv = *reinterpret_cast<const float4*>(&hbuf[bt][k]);  // broadcast
          g[bt] = fmaf(w[k], hv.x, g[bt]);
          p1[bt] = fmaf(w[k + 1], hv.y, p1[bt]);
          p2[bt] = fmaf(w[k + 2], hv.z, p2[bt]);
          p3[bt] = fmaf(w[k + 3], hv.w, p3[bt]);
        }
      }
#pragma unroll
      for (int bt = 0; bt < LSTM_BT; ++bt) gates[bt][tid] = (g[bt] + p1[bt]) + (p2[bt] + p3[bt]);
    }
    __syncthreads();
    if (cell_thread) {
      const float gi = gates[cbt][ce], gf = gates[cbt][D + ce], gg = gates[cbt][2 * D + ce],
                  go = gates[cbt][3 * D + ce];
      const float si = sigmoid_fast(gi);
      const float sf = sigmoid_fast(gf);
      const float so = sigmoid_fast(go);
      c = sf * c + si * tanh_fast(gg);
      const float h = so * tanh_fast(c);
      hbuf[cbt][ce] = h;
      if (cell_valid) {
        float* xr = x + ((size_t)(b0 + cbt) * L + s) * D + ce;
        *xr = *xr + h;
      }
    }
    __syncthreads();
  }
}

hipError_t launch_lstm_layer(float* x, const float* gx, const float* whh, int B, int L, int D, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  dim3 grid(cdiv(B, LSTM_BT));
  switch (D) {
#define X(d) \
    case d: hipLaunchKernelGGL(k_lstm_layer<d>, grid, dim3(((4 * d + 63) / 64) * 64), 0, s, x, gx, whh, B, L); break;
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace ffd
