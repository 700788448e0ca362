// Fused feed-forward block, large M (cached_transformer.py:325-327, nn.TransformerEncoderLayer._ff_block):
//     Y = LayerNorm2( X + W2 relu(W1 X + b1) + b2 )
// Row-owning waves + a CU-shared weight ring (round 3; replaces the F-split workgroup of k_ffn_ln at large M).
//
//   * Every wave owns 32 rows and walks the WHOLE hidden dimension on v_mfma_f32_32x32x2_f32 (64 cycles per
//     instruction: half the instructions per FLOP of the 16x16x4 form, and a single accumulation chain already runs
//     at the issue rate, so one wave per SIMD can keep the matrix pipe busy on its own).  Its Y^T accumulators hold
//     complete rows, so b2 + residual + LayerNorm2 happen in registers (a lane has 4 consecutive columns of one row
//     per 8-column group; a row is spread over the two lane halves: one xor-shuffle per statistic) and rows leave as
//     float4 stores.  No partial tiles through LDS, no reduction barriers.
//   * The weights are streamed ONCE PER CU: a ring of NSLOT slots in LDS, one slot = the packed fragments of 32 CPS
//     hidden units (W1 rows, W2 columns, b1) in exactly the order the waves consume them, filled by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPRs) NSLOT-1 slots ahead and read by all NW waves
//     with ds_read_b128 (one read = the A operands of 4 MFMAs), PD reads ahead of their MFMAs.
//   * One s_barrier per slot, placed BETWEEN the two products of the slot's last chunk: the fragments the first MFMAs
//     after the barrier need are already in registers, and the slot boundary itself has no barrier.  The barrier
//     certifies slot i+1 (every wave waits for its own DMA pieces with a counted vmcnt first) and frees slot i-1.
//   * relu'd GEMM1 accumulators are GEMM2's B operand (register r of the 32x32 accumulator = hidden units
//     f_r and f_r + 4 of the two lane halves, f_r = (r & 3) + 8 (r >> 2): a valid k pair); the d % 32 remainder
//     columns run on v_mfma_f32_4x4x1_16b_f32, all of a chunk's together (switching between the two MFMA forms
//     instruction by instruction was measured at 47 cycles per 4x4x1 instead of 8).
//   * Persistent: one workgroup per CU walks tiles of 32 NW rows; the weight stream simply wraps around.
// Summation order per output element is fixed (hidden units in packed order), independent of grid and tile assignment.
#include "ffd_internal.h"

namespace ffd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- ring pack: [F/32 chunks][SGC groups][64 lanes][4], groups in the order the waves consume them ---------------
// MFMA 32x32x2 operands: A lane l = A[i = l & 31][k = l >> 5], B lane l = B[k = l >> 5][j = l & 31],
//                        C/D lane l reg r = D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31]
//   groups 0 .. NQ1-1   W1 stream of chunk p, item idx = 4 g + j = k-step s:   W1[32 p + (lane & 31)][kperm(s, lane >> 5)]
//                       kperm: the k pair of step s is (c, c + 4) with c = 32 (s / 16) + 8 ((s % 16) / 4) + s % 4 -- the
//                       two lane halves of register s of a 32x32 accumulator tile that holds the row (ring_kperm), so
//                       that X rows loaded (or computed) in accumulator layout ARE the B operands, no lane movement
//   4 CT groups         item idx: r = idx / CT, ct = idx % CT:   W2[32 ct + (lane & 31)][32 p + f_r + 4 (lane >> 5)]
//   4 NG LOGICAL groups (4x4x1 A operands) item idx: r = idx / NG, g = idx % NG:
//                                                        W2[32 CT + 4 g + (lane & 3)][32 p + f_r + 4 (lane >> 5)]
//                       -- a lane's operand depends on (lane & 3, lane >> 5) only: 8 distinct float4 per group.  Round 4:
//                       stored COMPACTED, eight logical groups per 1 KiB piece ([logical group % 8][lane class = (lane & 3)
//                       + 4 (lane >> 5)][4]), read with a broadcast ds_read_b128: the remainder columns cost an eighth
//                       of the ring bytes and DMA pieces (d_model 60, with 7 remainder groups: 17 pieces per chunk
//                       instead of 41, which is what lets it fit the LDS; d_model 72 keeps whole groups: ring_compact)
//   last group          lanes 0..7: b1[32 p + 4 lane + j]
constexpr __host__ __device__ int ring_ct(int D) { return D / 32; }
constexpr __host__ __device__ int ring_ng(int D) { return (D % 32) / 4; }
constexpr __host__ __device__ int ring_ks2(int D) { return 16 * ring_ct(D) + 4 * ((ring_ng(D) + 1) / 2); }  // k pairs of GEMM1
constexpr __host__ __device__ int ring_nq1(int D) { return cdiv(ring_ks2(D), 4); }
// column of X that lane half `half` supplies to k-step s (>= D: none, the operand is zero)
constexpr __host__ __device__ int ring_kperm(int D, int s, int half) {
  return s < 16 * ring_ct(D) ? 32 * (s / 16) + 8 * ((s % 16) / 4) + 4 * half + s % 4
                             : 32 * ring_ct(D) + 4 * (2 * ((s - 16 * ring_ct(D)) / 4) + half) + (s - 16 * ring_ct(D)) % 4;
}
constexpr __host__ __device__ int ring_nmain(int D) { return ring_nq1(D) + 4 * ring_ct(D); }   // full 1 KiB groups of a chunk
constexpr __host__ __device__ int ring_nrem(int D) { return 4 * ring_ng(D); }                    // logical remainder groups
constexpr __host__ __device__ int ring_nfc(int D) { return ring_nmain(D) + ring_nrem(D); }       // logical fragment groups of a chunk
// d_model 72 keeps its remainder groups as full 1 KiB groups (the ring fits, and the compacted form measured 0.8 % slower
// there in an A/B on one box: 425.3 -> 428-429.5 us per launch at ECG B = 512); every other d_model stores them compacted
constexpr __host__ __device__ bool ring_compact(int D) { return D != 72; }
constexpr __host__ __device__ int ring_nfull(int D) { return ring_compact(D) ? ring_nmain(D) : ring_nfc(D); }  // groups stored whole
constexpr __host__ __device__ int ring_chunk_groups(int D) { return ring_nfull(D) + cdiv(ring_nfc(D) - ring_nfull(D), 8) + 1; }  // 1 KiB pieces
// float offset (inside its chunk / slot) of what `lane` reads of logical group k, k < nmain full groups, then compacted ones
constexpr __host__ __device__ int ring_frag_off(int nmain, int k, int lane) {
  return k < nmain ? k * 256 + lane * 4
                   : (nmain + (k - nmain) / 8) * 256 + (((k - nmain) % 8) * 8 + (lane & 3) + 4 * (lane >> 5)) * 4;
}
#ifndef FFD_ROWS_EXTRA_D  // (the pack kernels, knobs, planner and entry points live in the d_model 72 translation unit only)
size_t ffn_ring_floats(int D, int F) { return (size_t)(F / 32) * ring_chunk_groups(D) * 256; }

__global__ void k_pack_ffn_ring(const float* __restrict__ W1, const float* __restrict__ b1,
                                const float* __restrict__ W2, float* __restrict__ out, int D, int F) {
  const int KS2 = ring_ks2(D), CT = ring_ct(D), NG = ring_ng(D), NQ1 = ring_nq1(D), SG = ring_chunk_groups(D);
  const size_t total = (size_t)(F / 32) * SG * 256;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(o & 3), lane = (int)((o >> 2) & 63);
    const int g = (int)((o >> 8) % SG), p = (int)((o >> 8) / SG);
    const int half = lane >> 5;
    float v = 0.f;
    if (g < NQ1) {
      const int s = 4 * g + j;
      const int kcol = s < KS2 ? ring_kperm(D, s, half) : D;
      if (kcol < 32 * CT + 4 * NG) v = W1[(size_t)(32 * p + (lane & 31)) * D + kcol];
    } else if (g < NQ1 + 4 * CT) {
      const int idx = 4 * (g - NQ1) + j, r = idx / CT, ct = idx % CT;
      v = W2[(size_t)(32 * ct + (lane & 31)) * F + 32 * p + (r & 3) + 8 * (r >> 2) + 4 * half];
    } else if (g < SG - 1 && !ring_compact(D)) {  // a whole group per logical remainder group
      const int idx = 4 * (g - NQ1 - 4 * CT) + j, r = idx / NG, gq = idx % NG;
      v = W2[(size_t)(32 * CT + 4 * gq + (lane & 3)) * F + 32 * p + (r & 3) + 8 * (r >> 2) + 4 * half];
    } else if (g < SG - 1) {  // compacted remainder piece: [logical group % 8][lane class][4]
      const int t = (int)(o & 255), lc = (t >> 2) & 7, kr = (g - NQ1 - 4 * CT) * 8 + (t >> 5);
      if (kr < 4 * NG) {
        const int idx = 4 * kr + j, r = idx / NG, gq = idx % NG;
        v = W2[(size_t)(32 * CT + 4 * gq + (lc & 3)) * F + 32 * p + (r & 3) + 8 * (r >> 2) + 4 * (lc >> 2)];
      }
    } else if (lane < 8) {
      v = b1[32 * p + 4 * lane + j];
    }
    out[o] = v;
  }
}

#endif
// ---- out-projection slot (fused out-proj + LN1 form): [NPM main-tile groups][NPR remainder groups][64 lanes][4] --------
//   main groups      item idx = 4 g + j: k-step s = idx / CT, column tile ct = idx % CT:  Wo[32 ct + (lane & 31)][kperm(s, lane >> 5)]
//   remainder groups item idx: s = idx / NG, gq = idx % NG (4x4x1 A operands):   Wo[32 CT + 4 gq + (lane & 3)][kperm(s, lane >> 5)]
//                    (logical groups, stored compacted eight to a piece like the ring's remainder groups)
// (the attention rows, loaded in accumulator layout like the FFN's X rows, are the B operands under the same k permutation)
constexpr __host__ __device__ int oproj_npm(int D) { return cdiv(ring_ks2(D) * ring_ct(D), 4); }
constexpr __host__ __device__ int oproj_npr(int D) { return cdiv(ring_ks2(D) * ring_ng(D), 4); }
#ifndef FFD_ROWS_EXTRA_D
size_t ffn_ring_oproj_floats(int D) { return (size_t)2 * ring_chunk_groups(D) * 256; }  // one two-chunk slot

__global__ void k_pack_oproj_ring(const float* __restrict__ Wo, float* __restrict__ out, int D) {
  const int KS2 = ring_ks2(D), CT = ring_ct(D), NG = ring_ng(D), NPM = oproj_npm(D), NPR = oproj_npr(D);
  const int total = 2 * ring_chunk_groups(D) * 256;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
    const int j = o & 3, lane = (o >> 2) & 63, g = o >> 8, half = lane >> 5;
    float v = 0.f;
    if (g < NPM) {
      const int idx = 4 * g + j, s = idx / CT, ct = idx % CT;
      if (s < KS2) {
        const int kcol = ring_kperm(D, s, half);
        if (kcol < D) v = Wo[(size_t)(32 * ct + (lane & 31)) * D + kcol];
      }
    } else if (!ring_compact(D) && g < NPM + NPR) {  // a whole group per logical remainder group
      const int idx = 4 * (g - NPM) + j, s = idx / NG, gq = idx % NG;
      if (s < KS2) {
        const int kcol = ring_kperm(D, s, half);
        if (kcol < D) v = Wo[(size_t)(32 * CT + 4 * gq + (lane & 3)) * D + kcol];
      }
    } else if (ring_compact(D) && g < NPM + (NPR + 7) / 8) {  // compacted remainder piece: [logical group % 8][lane class][4]
      const int t = o & 255, lc = (t >> 2) & 7, kr = (g - NPM) * 8 + (t >> 5);
      if (kr < NPR) {
        const int idx = 4 * kr + j, s = idx / NG, gq = idx % NG;
        if (s < KS2) {
          const int kcol = ring_kperm(D, s, lc >> 2);
          if (kcol < D) v = Wo[(size_t)(32 * CT + 4 * gq + (lc & 3)) * D + kcol];
        }
      }
    }
    out[o] = v;
  }
}

hipError_t launch_pack_oproj_ring(const float* Wo, float* out, int D, hipStream_t s) {
  if (D % 4 != 0 || D < 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pack_oproj_ring, dim3(64), dim3(256), 0, s, Wo, out, D);
  return hipGetLastError();
}

hipError_t launch_pack_ffn_ring(const float* W1, const float* b1, const float* W2, float* out, int D, int F,
                                hipStream_t s) {
  if (F % 32 != 0 || D % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pack_ffn_ring, dim3(512), dim3(256), 0, s, W1, b1, W2, out, D, F);
  return hipGetLastError();
}

#endif

__device__ __forceinline__ float f4e(const float4& q, int j) { return j == 0 ? q.x : j == 1 ? q.y : j == 2 ? q.z : q.w; }

template <int D, int NW, int CPS, int NSLOT>
struct FfnRowsCfg {
  static constexpr int KS2 = ring_ks2(D);
  static constexpr int CT = ring_ct(D), NG = ring_ng(D), NQ1 = ring_nq1(D);
  static constexpr int SGC = ring_chunk_groups(D);           // 1 KiB pieces per chunk (the last one: bias)
  static constexpr int NMAIN = ring_nfull(D);                // groups of a chunk stored whole; those after them are compacted
  static constexpr int NFC = ring_nfc(D);                    // (logical) fragment groups of a chunk = ds_read_b128 per lane
  static constexpr int SLOT_G = CPS * SGC;
  static constexpr int SLOT_FLOATS = SLOT_G * 256;
  static constexpr int LNP = 6 * D;                          // b2, gamma2, beta2; out-proj bias, gamma1, beta1 (fused form)
  static constexpr int LNP_PAD = cdiv(LNP, 4) * 4;
  static constexpr int LDS_FLOATS = NSLOT * SLOT_FLOATS + LNP_PAD;
  static constexpr int NST = 4 * CT + (NG + 1) / 2;          // float4 stores per lane in a tile epilogue (at most)
  static constexpr int AHEAD = NSLOT - 1;                    // a slot's DMA is issued AHEAD slots before its use
  static constexpr int PD = 3;                               // fragment groups requested ahead of their MFMAs
  static constexpr int NPW = cdiv(SLOT_G, NW);               // DMA pieces of a slot per wave (the same for every wave: the
                                                             // last pieces of a slot are fetched twice when NW does not divide)
  static_assert((AHEAD - 2) * NPW + NST <= 63 && AHEAD >= 2, "vmcnt range");
  static_assert(NPW <= 2 * (NFC - NQ1), "at most two DMA pieces per fragment group after the barrier");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS");
  // fused out-projection slot
  static constexpr int NPM = oproj_npm(D), NPR = oproj_npr(D), NFP = NPM + NPR;  // logical groups of the slot
  static constexpr int NPF = ring_compact(D) ? NPM : NFP;                        // its groups stored whole
  static constexpr int NFPP = NPF + cdiv(NFP - NPF, 8);                          // its 1 KiB pieces
};

// b + residual + LayerNorm of the wave's 32 rows, in registers.  acc[ct][4 t + i] = Y^T[c = 32 ct + 8 t + 4 half + i][row m]
// (four consecutive columns per (ct, t)), rem[g][e] = this lane half's partial sum of column 32 CT + 4 g + e; res / resrem
// the residual rows in the same layout; p = {bias[D], gamma[D], beta[D]} in LDS.  Returns the normalised rows in
// v / vr (lane half h owns remainder groups g = 2 i + h).
template <int D>
__device__ __forceinline__ void ln_rows(const f32x16* acc, const f32x4* rem, const float4 (*res)[4], const float4* resrem,
                                        const float* p, int half, float4 (*v)[4], float4* vr) {
  constexpr int CT = ring_ct(D), NG = ring_ng(D);
  float sum = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float4 bq = *reinterpret_cast<const float4*>(&p[32 * ct + 8 * t + 4 * half]);
      const float4 x4 = res[ct][t];
      v[ct][t] = float4{x4.x + (acc[ct][4 * t] + bq.x), x4.y + (acc[ct][4 * t + 1] + bq.y),
                        x4.z + (acc[ct][4 * t + 2] + bq.z), x4.w + (acc[ct][4 * t + 3] + bq.w)};
      sum += (v[ct][t].x + v[ct][t].y) + (v[ct][t].z + v[ct][t].w);
    }
  if (NG > 0) {
    // the two lane halves hold partial sums over their own k values: add them (both halves get the total);
    // half h then finishes groups g = 2 i + h
#pragma unroll
    for (int i = 0; i < (NG + 1) / 2; ++i) {
      float a[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t0 = rem[2 * i][e];
        t0 += __shfl_xor(t0, 32);
        float t1 = 0.f;
        if (2 * i + 1 < NG) {
          t1 = rem[(2 * i + 1 < NG) ? 2 * i + 1 : 0][e];
          t1 += __shfl_xor(t1, 32);
        }
        a[e] = half ? t1 : t0;
      }
      vr[i] = float4{0.f, 0.f, 0.f, 0.f};
      if (2 * i + half < NG) {
        const float4 bq = *reinterpret_cast<const float4*>(&p[32 * CT + 4 * (2 * i + half)]);
        const float4 x4 = resrem[i];
        vr[i] = float4{x4.x + (a[0] + bq.x), x4.y + (a[1] + bq.y), x4.z + (a[2] + bq.z), x4.w + (a[3] + bq.w)};
        sum += (vr[i].x + vr[i].y) + (vr[i].z + vr[i].w);
      }
    }
  }
  sum += __shfl_xor(sum, 32);
  const float mean = sum * (1.0f / D);
  float ss = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float a = v[ct][t].x - mean, b = v[ct][t].y - mean, c2 = v[ct][t].z - mean, d = v[ct][t].w - mean;
      ss = fmaf(a, a, ss), ss = fmaf(b, b, ss), ss = fmaf(c2, c2, ss), ss = fmaf(d, d, ss);
    }
#pragma unroll
  for (int i = 0; i < (NG + 1) / 2; ++i)
    if (2 * i + half < NG) {
      const float a = vr[i].x - mean, b = vr[i].y - mean, c2 = vr[i].z - mean, d = vr[i].w - mean;
      ss = fmaf(a, a, ss), ss = fmaf(b, b, ss), ss = fmaf(c2, c2, ss), ss = fmaf(d, d, ss);
    }
  ss += __shfl_xor(ss, 32);
  const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c0 = 32 * ct + 8 * t + 4 * half;
      const float4 g4 = *reinterpret_cast<const float4*>(&p[D + c0]);
      const float4 e4 = *reinterpret_cast<const float4*>(&p[2 * D + c0]);
      v[ct][t] = float4{(v[ct][t].x - mean) * rstd * g4.x + e4.x, (v[ct][t].y - mean) * rstd * g4.y + e4.y,
                        (v[ct][t].z - mean) * rstd * g4.z + e4.z, (v[ct][t].w - mean) * rstd * g4.w + e4.w};
    }
#pragma unroll
  for (int i = 0; i < (NG + 1) / 2; ++i)
    if (2 * i + half < NG) {
      const int c0 = 32 * CT + 4 * (2 * i + half);
      const float4 g4 = *reinterpret_cast<const float4*>(&p[D + c0]);
      const float4 e4 = *reinterpret_cast<const float4*>(&p[2 * D + c0]);
      vr[i] = float4{(vr[i].x - mean) * rstd * g4.x + e4.x, (vr[i].y - mean) * rstd * g4.y + e4.y,
                     (vr[i].z - mean) * rstd * g4.z + e4.z, (vr[i].w - mean) * rstd * g4.w + e4.w};
    }
}

// The sliced form's tile end: the wave's partial rows as they are (slice 0: + bias + residual), same layout as ln_rows.
template <int D>
__device__ __forceinline__ void raw_rows(const f32x16* acc, const f32x4* rem, const float4 (*res)[4], const float4* resrem,
                                         const float* p, int half, bool first, float4 (*v)[4], float4* vr) {
  constexpr int CT = ring_ct(D), NG = ring_ng(D);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v[ct][t] = float4{acc[ct][4 * t], acc[ct][4 * t + 1], acc[ct][4 * t + 2], acc[ct][4 * t + 3]};
      if (first) {
        const float4 bq = *reinterpret_cast<const float4*>(&p[32 * ct + 8 * t + 4 * half]);
        const float4 x4 = res[ct][t];
        v[ct][t] = float4{x4.x + (v[ct][t].x + bq.x), x4.y + (v[ct][t].y + bq.y), x4.z + (v[ct][t].z + bq.z),
                          x4.w + (v[ct][t].w + bq.w)};
      }
    }
#pragma unroll
  for (int i = 0; i < (NG + 1) / 2; ++i) {
    float a[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t0 = rem[2 * i][e];
      t0 += __shfl_xor(t0, 32);
      float t1 = 0.f;
      if (2 * i + 1 < NG) {
        t1 = rem[(2 * i + 1 < NG) ? 2 * i + 1 : 0][e];
        t1 += __shfl_xor(t1, 32);
      }
      a[e] = half ? t1 : t0;
    }
    vr[i] = float4{a[0], a[1], a[2], a[3]};
    if (first && 2 * i + half < NG) {
      const float4 bq = *reinterpret_cast<const float4*>(&p[32 * CT + 4 * (2 * i + half)]);
      const float4 x4 = resrem[i];
      vr[i] = float4{x4.x + (a[0] + bq.x), x4.y + (a[1] + bq.y), x4.z + (a[2] + bq.z), x4.w + (a[3] + bq.w)};
    }
  }
}

// OP (fused form, cached_transformer.py:316-327 in one launch): X is the ATTENTION output and Rin the layer input; a
// tile starts with one extra ring slot -- the out-projection fragments (`ringp`) -- from which every wave computes
// x1 = LayerNorm1(Rin + Wo attn + bo) of its 32 rows on the same two MFMA forms; x1 never leaves the registers (it is
// GEMM1's B operand and LN2's residual).  Without OP, X is x1 itself (k_linear_res_ln wrote it).
// SL (sliced form, mid-size M; with OP only): workgroup u takes tile u / nslice and the hidden units of slice u % nslice
// only (its share of the chunk slots, after the out-projection slot every slice recomputes), and stores its partial rows
// raw -- slice 0 with x1 + b2 added -- to Y[slice][M][D]; k_rows_reduce_ln adds the slices in order and normalises.
// The grid is g x nslice workgroups (<= the CUs): workgroup (t0, slice) takes tiles t0, t0 + g, ... of its slice (round 4:
// where tiles x nslice exceeds the CUs the units go through them in rounds, e.g. ECG B = 384: 187 tiles x 4 slices on
// 64 x 4 workgroups in 3 rounds of a quarter tile instead of one round that leaves 69 CUs idle).
// ONE: at most one tile per workgroup (the grid covers the tiles: ECG B = 512 is 250 tiles of 384 rows): the tile loop
// is gone at compile time, and with it the loop-carried row registers that cost the fused form 68 B of scratch.
template <int D, int NW, int CPS, int NSLOT, int PR, bool OP, bool SL, bool ONE>
__global__ __launch_bounds__(64 * NW, 3) void k_ffn_rows(const float* __restrict__ X, const float* __restrict__ Rin,
                                                             const float* __restrict__ ring, const float* __restrict__ ringp,
                                                             const float* __restrict__ b2, const float* __restrict__ gam,
                                                             const float* __restrict__ bet, const float* __restrict__ bo,
                                                             const float* __restrict__ gam1, const float* __restrict__ bet1,
                                                             float* __restrict__ Y, int M, int F, int nslice,
                                                             unsigned long long* stamp) {
  // (SL without OP, round 4: X is x1 itself -- k_linear_res_ln ran once for all slices -- and slice 0 adds x1 + b2)
  // stamp (diagnostic launches of ffd_probe_ffn_clock only, nullptr otherwise): the record k_ffn_ln writes (8 x u64 per
  // workgroup), to memory nothing else reads
  using C = FfnRowsCfg<D, NW, CPS, NSLOT>;
  constexpr int KS2 = C::KS2, CT = C::CT, NG = C::NG, NQ1 = C::NQ1, SGC = C::SGC, NFC = C::NFC, PD = C::PD;
  constexpr int NGA = NG > 0 ? NG : 1, CTA = CT > 0 ? CT : 1;
  constexpr int NRH = (NG + 1) / 2;  // remainder groups per lane half
  constexpr int NRA = NRH > 0 ? NRH : 1;
  constexpr int R = 32 * NW;
  constexpr int NFS = CPS * NFC;  // fragment groups of a slot
  constexpr int RPG = cdiv(4, CTA);  // accumulator registers (k pairs) one GEMM2 fragment group covers
  constexpr int NPM = C::NPM, NFP = C::NFP;
  static_assert(CT == 0 || 4 % CT == 0, "relu placement assumes CT in {1, 2, 4}");
  static_assert(!OP || (CPS == 2 && C::AHEAD == 2 && NFP + PD <= NFS + PD && C::NFPP <= C::SLOT_G && CT > 0),
                "the fused form: two-chunk slots, ring one slot ahead of the barrier");
  __shared__ __align__(16) float lds[C::LDS_FLOATS];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, m = lane & 31;
  float* const ringl = lds;
  float* const lnp = lds + NSLOT * C::SLOT_FLOATS;
  const unsigned ring_base = __builtin_amdgcn_readfirstlane(lds_addr(ringl));

  const int ntiles = (M + R - 1) / R;
  const int slice = SL ? (int)blockIdx.x % nslice : 0;
  // chunk slots of a unit: slice s of n takes slots [s N / n, (s + 1) N / n) (uneven when n does not divide N)
  const int slot_first = SL ? slice * (F / (32 * CPS)) / nslice : 0;
  const int nchunk_slots = SL ? (slice + 1) * (F / (32 * CPS)) / nslice - slot_first : F / (32 * CPS);
  const int NSL = nchunk_slots + (OP ? 1 : 0);  // slots per tile (fused form: slot 0 = the out-projection)
  // (sliced form: the grid is g tiles x nslice workgroups; workgroup (t0, slice) walks tiles t0, t0 + g, ... of ITS slice,
  //  so the ring's slot stream is the same for every tile it takes)
  const int tstride = SL ? (int)gridDim.x / nslice : (int)gridDim.x;
  const int my_tiles = SL ? ((int)blockIdx.x / nslice < ntiles ? (ntiles - 1 - (int)blockIdx.x / nslice) / tstride + 1 : 0)
                       : ONE ? ((int)blockIdx.x < ntiles ? 1 : 0)
                             : ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int total = my_tiles * NSL;
  if (total == 0) return;  // (uniform over the workgroup)

  const float* const ring_s = ring + (size_t)slot_first * C::SLOT_FLOATS;  // this unit's chunk slots
  auto slot_src = [&](int wslot) -> const float* {  // packed slot `wslot` of a tile's stream
    return OP ? (wslot == 0 ? ringp : ring_s + (size_t)(wslot - 1) * C::SLOT_FLOATS) : ring_s + (size_t)wslot * C::SLOT_FLOATS;
  };

  float4 xv[CTA][4], xrem[NRA];  // X rows (B operands of GEMM1 and the residual)
  f32x16 yacc[CTA];
  f32x4 yrem[NGA];
  int tile = SL ? (int)blockIdx.x / nslice : (int)blockIdx.x;
  int wnext = C::AHEAD % NSL;  // packed slot the next ring DMA fetches
  // The wave's 32 rows of `src` in ACCUMULATOR layout: lane (row m, half) holds columns 32 ct + 8 t + 4 half + (0..3) as
  // one float4 per (ct, t) -- B operands under the pack's k permutation, and residual rows.
  auto load_rows = [&](const float* src, float4 (*q)[4], float4* qrem) {
    const int row = min(tile * R + wave * 32 + m, M - 1);  // rows past M repeat row M-1; they are never stored
    const float* xr = src + (size_t)row * D;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int t = 0; t < 4; ++t) q[ct][t] = *reinterpret_cast<const float4*>(xr + 32 * ct + 8 * t + 4 * half);
#pragma unroll
    for (int i = 0; i < NRH; ++i) {  // lane half h holds remainder groups g = 2 i + h
      qrem[i] = float4{0.f, 0.f, 0.f, 0.f};
      if (2 * i + half < NG) qrem[i] = *reinterpret_cast<const float4*>(xr + 32 * CT + 4 * (2 * i + half));
    }
  };
  // Retire loads HERE: left to itself hipcc waits for them with counted vmcnt in front of their first use in the loop
  // body, in EVERY iteration (it cannot know the loads are not re-issued), and those counts also drain the ring's
  // LDS-DMA pieces, which share the counter: the ring then runs one slot deep.
  auto retire_rows = [&](float4 (*q)[4], float4* qrem) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(q[ct][t].x), "+v"(q[ct][t].y), "+v"(q[ct][t].z), "+v"(q[ct][t].w));
#pragma unroll
    for (int i = 0; i < NRH; ++i) asm volatile("" : "+v"(qrem[i].x), "+v"(qrem[i].y), "+v"(qrem[i].z), "+v"(qrem[i].w));
  };
  auto xb_of = [&](int s) -> float {  // B operand of k-step s
    return s < 16 * CT ? f4e(xv[s / 16 < CTA ? s / 16 : 0][(s % 16) / 4], s % 4)
                       : f4e(xrem[(s - 16 * CT) / 4 < NRH ? (s - 16 * CT) / 4 : 0], (s - 16 * CT) % 4);
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int ct = 0; ct < CTA; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) yacc[ct][r] = 0.f;
#pragma unroll
    for (int g = 0; g < NGA; ++g) yrem[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- prologue: LN parameters; the first slot and the first tile's rows are waited for, the second slot's pieces
  //      stay in flight (certified by the first slot barrier: mid-slot in the out-projection slot / the chunk slot) ----
  static_assert(C::AHEAD == 2 || !OP, "prologue of the fused form");
  for (int i = threadIdx.x; i < (OP ? 6 : 3) * D; i += 64 * NW)
    lnp[i] = i < D ? b2[i] : i < 2 * D ? gam[i - D] : i < 3 * D ? bet[i - 2 * D]
           : i < 4 * D ? bo[i - 3 * D] : i < 5 * D ? gam1[i - 4 * D] : bet1[i - 5 * D];
  {
    const float* src = slot_src(0) + lane * 4;  // (the out-projection slot has NFP fragment groups, the rest is padding)
    for (int g = wave; g < (OP ? C::NFPP : C::SLOT_G); g += NW) dma_piece(src + g * 256, ring_base + g * 1024);
  }
  float4 rin[CTA][4], rinrem[NRA];  // fused form: the layer input rows (LN1's residual)
  load_rows(X, xv, xrem);
  if (OP) load_rows(Rin, rin, rinrem);
  wait_vm<0>();
  retire_rows(xv, xrem);
  if (OP) retire_rows(rin, rinrem);
#pragma unroll
  for (int j = 1; j < C::AHEAD; ++j)
    if (j < total) {  // NPW pieces per wave (a piece index past the slot wraps: fetched twice), like the slots after it
      const float* src = slot_src(j % NSL) + lane * 4;
      const unsigned dst = ring_base + (unsigned)(j % NSLOT) * (C::SLOT_FLOATS * 4);
#pragma unroll
      for (int i = 0; i < C::NPW; ++i) {
        const int g0 = wave + i * NW, g = g0 < C::SLOT_G ? g0 : g0 - C::SLOT_G;
        dma_piece(src + g * 256, dst + g * 1024);
      }
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LN parameter writes
  __builtin_amdgcn_s_barrier();

  // The slot is consumed as a stream of NFS fragment groups (one ds_read_b128 each = the A operands of 4 MFMAs);
  // group k is requested PD groups before its MFMAs, across chunk and slot boundaries too (the first PD groups and
  // the bias fragments of a chunk are requested under the previous chunk's last MFMAs).
  float4 hb[4], f[NFS + PD];
  {
    const float* slot = ringl;
    if (!OP) {
#pragma unroll
      for (int t = 0; t < 4; ++t) hb[t] = *reinterpret_cast<const float4*>(slot + (SGC - 1) * 256 + 4 * (2 * t + half));
    }
#pragma unroll
    for (int k = 0; k < PD; ++k) f[k] = *reinterpret_cast<const float4*>(slot + k * 256 + lane * 4);
  }
  unsigned long long st_clk = 0, st_rt = 0, st_acc = 0, st_acc_rt = 0, st_first_b = 0, st_first_e = 0, st_epi = 0;
  const unsigned long long st_entry = stamp ? __builtin_amdgcn_s_memrealtime() : 0ull;
  int st_tiles = 0;


  // Per ring slot: the lane index is re-derived (two VALU instructions, from inline asm so that it is not hoisted): as
  // a loop-invariant register it is the first thing hipcc spills at 168 VGPRs, and the reload of a scratch dword sits
  // behind an s_waitcnt vmcnt(0) that drains the DMA pieces just issued.  The DMA pieces of slot it + AHEAD go out one
  // per fragment group after the slot's barrier (an LDS-DMA issue holds the wave for 60-180 cycles: one fits in the
  // shadow of a 64-cycle MFMA, a burst of them does not).
#define FFD_SLOT_PRELUDE                                                                                               \
  int lane_i;                                                                                                          \
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_i));                       \
  const int half_i = lane_i >> 5;                                                                                      \
  const float* slot = ringl + (it % NSLOT) * C::SLOT_FLOATS;                                                           \
  const float* nslot = ringl + ((it + 1) % NSLOT) * C::SLOT_FLOATS;                                                    \
  const bool dma_on = it + C::AHEAD < total;                                                                           \
  const float* dma_src = slot_src(wnext) + lane_i * 4;                                                                 \
  const unsigned dma_dst = ring_base + (unsigned)((it + C::AHEAD) % NSLOT) * (C::SLOT_FLOATS * 4);                     \
  int dma_g = wave; /* next piece of the slot this wave issues */                                                      \
  auto issue_piece = [&]() { /* (a piece index past the slot wraps to a piece some other wave also fetches) */         \
    if (dma_on) {                                                                                                      \
      const int g = dma_g < C::SLOT_G ? dma_g : dma_g - C::SLOT_G;                                                     \
      dma_piece(dma_src + g * 256, dma_dst + g * 1024);                                                                \
      dma_g += NW;                                                                                                     \
    }                                                                                                                  \
  }

  int it = 0;  // ring slot counter over the whole launch
  for (int tl = 0; tl < my_tiles; ++tl) {
    if (stamp) {
      st_clk = __builtin_amdgcn_s_memtime(), st_rt = __builtin_amdgcn_s_memrealtime();
      if (st_tiles == 0) st_first_b = st_rt;
    }
    if (OP) {
      // ---- tile start, fused form: x1 = LN1(Rin + Wo attn + bo) of the wave's rows from the out-projection slot ----
      FFD_SLOT_PRELUDE;
      if (it > 0) {
        // the slot's barrier stands at its HEAD here (all waves have just finished a tile together; the rows of this
        // tile were requested and retired at the end of the previous one): slot it+1 is certified, slot it-1 free
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
      }
#pragma unroll
      for (int i = 0; i < C::NPW; ++i) issue_piece();  // (a burst: nothing of this wave's is there to be delayed yet)
      zero_acc();  // (the accumulators of the out-projection; the FFN's start from zero again below)
#pragma unroll
      for (int K = 0; K < NFP; ++K) {
        const float4 w = f[K];
        if (PR) {  // descending through the slot, like the chunk slots between their barriers
          if (K == 0) __builtin_amdgcn_s_setprio(3);
          else if (K == NFP / 4) __builtin_amdgcn_s_setprio(2);
          else if (K == NFP / 2) __builtin_amdgcn_s_setprio(1);
          else if (K == 3 * NFP / 4) __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (K < NPM) {  // main column tiles, CT independent chains
            const int idx = 4 * K + j, s = idx / CTA, ct = idx % CTA;
            if (s < KS2) yacc[ct] = mfma32(f4e(w, j), xb_of(s), yacc[ct]);
          } else {  // remainder columns on the 4x4x1 form
            const int idx = 4 * (K - NPM) + j, s = idx / NGA, g = idx % NGA;
            if (s < KS2) yrem[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(f4e(w, j), xb_of(s), yrem[g], 0, 0, 0);
          }
          if (j == 1) {  // request group K + PD (of the next slot, a chunk slot, once past the end)
            const int KP = K + PD;
            if (KP < NFP) f[KP] = *reinterpret_cast<const float4*>(slot + ring_frag_off(C::NPF, KP, lane_i));
            else f[KP] = *reinterpret_cast<const float4*>(nslot + ring_frag_off(C::NMAIN, KP - NFP, lane_i));
            if (K == NFP - 1) {
#pragma unroll
              for (int t = 0; t < 4; ++t) hb[t] = *reinterpret_cast<const float4*>(nslot + (SGC - 1) * 256 + 4 * (2 * t + half_i));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (K == NPM - 1) {
          // the next slot (the tile's first chunk slot) is read from here on: my pieces of it have landed (the burst
          // above, younger, stays in flight); after the first tile the head barrier has certified it already
          if (dma_on) wait_vm<C::NPW>(); else wait_vm<0>();
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int k = 0; k < PD; ++k) f[k] = f[NFP + k];
      {
        float4 v[CTA][4], vr[NRA];
        ln_rows<D>(yacc, yrem, rin, rinrem, lnp + 3 * D, half, v, vr);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int t = 0; t < 4; ++t) xv[ct][t] = v[ct][t];
#pragma unroll
        for (int i = 0; i < NRH; ++i) xrem[i] = (2 * i + half < NG) ? vr[i] : float4{0.f, 0.f, 0.f, 0.f};
      }
      if (dma_on && ++wnext == NSL) wnext = 0;
      ++it;
    }
    zero_acc();

    for (int sl = OP ? 1 : 0; sl < NSL; ++sl, ++it) {
      FFD_SLOT_PRELUDE;
      f32x16 h;
#pragma unroll
      for (int K = 0; K < NFS; ++K) {
        const int c = K / NFC, k = K % NFC;  // chunk of the slot, group of the chunk (compile-time after unrolling)
        const float4 w = f[K];
        if (PR) {
          // Descending priority through the barrier interval: a wave ahead of its SIMD's other waves yields to them
          // (issue arbitration is priority, then age: at equal priority the oldest wave takes every slot it can use and
          // the youngest runs the last third of the interval alone, at a lone wave's efficiency).
          constexpr int KB = (CPS - 1) * NFC + NQ1;  // first group after the slot's barrier
          const int rel = (K - KB + NFS) % NFS;
          if (rel == 0 || (rel * 4) / NFS != ((rel - 1) * 4) / NFS) {
            const int lvl = 3 - (rel * 4) / NFS;  // (the builtin wants a literal; the chain folds after unrolling)
            if (lvl == 3) __builtin_amdgcn_s_setprio(3);
            else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
            else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
          }
        }
        if (k == 0) {  // the bias is GEMM1's initial accumulator
#pragma unroll
          for (int t = 0; t < 4; ++t) h[4 * t] = hb[t].x, h[4 * t + 1] = hb[t].y, h[4 * t + 2] = hb[t].z, h[4 * t + 3] = hb[t].w;
        }
        // A group = 4 MFMAs (64 cycles each).  The wave's other work is placed in the gaps BETWEEN them, one kind per
        // gap (an in-order wave cannot issue past an MFMA the pipe has not accepted yet, so only what sits in a gap
        // hides under the preceding MFMA): gap 0 relu of the accumulator registers the next GEMM2 group reads, gap 1
        // the fragment read PD groups ahead, gap 2 one LDS-DMA piece.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (k < NQ1) {
            // ---- GEMM1: H^T chunk (32 hidden x 32 rows), K = D ----
            const int s = 4 * k + j;
            if (s < KS2) h = mfma32(f4e(w, j), xb_of(s), h);
          } else if (k < NQ1 + 4 * CT) {
            // ---- GEMM2: Y^T += W2[:, chunk] relu(H^T chunk); accumulator register r is the k pair (f_r, f_r + 4) ----
            const int idx = 4 * (k - NQ1) + j, r = idx / CTA, ct = idx % CTA;
            yacc[ct] = mfma32(f4e(w, j), h[r], yacc[ct]);
            if (j == 0) {  // relu (one v_med3_f32 (x, 0, +inf) per element)
#pragma unroll
              for (int rr = 0; rr < RPG; ++rr) {
                const int rn = (k - NQ1 + 1) * RPG + rr;
                if (rn < 16) h[rn] = __builtin_amdgcn_fmed3f(h[rn], 0.f, __builtin_inff());
              }
            }
          } else {
            // remainder columns on the 4x4x1 form
            const int idx = 4 * (k - NQ1 - 4 * CT) + j, r = idx / NGA, g = idx % NGA;
            yrem[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(f4e(w, j), h[r], yrem[g], 0, 0, 0);
          }
          if (j == 1) {
            // request group K + PD (of the next slot once past the end: certified by this iteration's barrier)
            const int KP = K + PD;
            const float* base = KP < NFS ? slot : nslot;
            const int kk = KP < NFS ? KP : KP - NFS;
            f[KP] = *reinterpret_cast<const float4*>(base + (kk / NFC) * SGC * 256 + ring_frag_off(C::NMAIN, kk % NFC, lane_i));
            if (k == NFC - 1) {  // bias fragments of the next chunk (of the next slot after the last chunk)
              const float* bsrc = (c + 1 < CPS) ? slot + ((c + 1) * SGC + SGC - 1) * 256 : nslot + (SGC - 1) * 256;
#pragma unroll
              for (int t = 0; t < 4; ++t) hb[t] = *reinterpret_cast<const float4*>(bsrc + 4 * (2 * t + half_i));
            }
          }
          {  // this wave's NPW pieces of slot it + AHEAD: one per group from the barrier on (a second one per group where
             // the groups left in the slot are fewer than the pieces: d_model 64 at four waves)
            constexpr int KB0 = (CPS - 1) * NFC + NQ1, W = NFC - NQ1;
            if (j == 2 && K >= KB0 && K - KB0 < C::NPW) issue_piece();
            if (j == 3 && K >= KB0 && K - KB0 + W < C::NPW) issue_piece();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (k == NQ1 - 1) {
          // relu of the accumulator registers the first GEMM2 group reads
#pragma unroll
          for (int r = 0; r < RPG; ++r) h[r] = __builtin_amdgcn_fmed3f(h[r], 0.f, __builtin_inff());
          if (c == CPS - 1) {
            // ---- the slot's barrier: my pieces of slot it+1 have landed; afterwards slot it+1 is readable by
            //      everyone and nobody reads slot it-1 any more.  In flight and allowed to stay so: the slots after
            //      it+1 that have been issued (it+2 .. it+AHEAD-1), and the previous tile's stores (younger) ----
            if (C::AHEAD > 2 && it + C::AHEAD - 1 < total) {
              if (sl == 0 && it > 0) wait_vm<(C::AHEAD - 2) * C::NPW + C::NST>(); else wait_vm<(C::AHEAD - 2) * C::NPW>();
            } else {
              wait_vm<0>();
            }
            __builtin_amdgcn_s_barrier();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int k = 0; k < PD; ++k) f[k] = f[NFS + k];
      if (dma_on && ++wnext == NSL) wnext = 0;
    }

    // ---- tile end: + b2, + residual, LayerNorm2, float4 stores ----
    if (stamp) {
      const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
      st_acc += __builtin_amdgcn_s_memtime() - st_clk;
      st_acc_rt += rt - st_rt;
      if (st_tiles == 0) st_first_e = rt;
    }
    {
      float4 v[CTA][4], vr[NRA];
      if (SL) raw_rows<D>(yacc, yrem, xv, xrem, lnp, half, slice == 0, v, vr);
      else ln_rows<D>(yacc, yrem, xv, xrem, lnp, half, v, vr);
      const int row = tile * R + wave * 32 + m;
      if (row < M) {
        float* yr = Y + ((size_t)(SL ? slice : 0) * M + row) * D;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(yr + 32 * ct + 8 * t + 4 * half) = v[ct][t];
#pragma unroll
        for (int i = 0; i < (NG + 1) / 2; ++i)
          if (2 * i + half < NG) *reinterpret_cast<float4*>(yr + 32 * CT + 4 * (2 * i + half)) = vr[i];
      }
    }
    if (stamp) {
      if (st_tiles == 0) st_epi = __builtin_amdgcn_s_memrealtime();
      ++st_tiles;
    }
    tile += tstride;
    if (tl + 1 < my_tiles) {  // the next tile's rows (the first tile's came with the ring fill)
      load_rows(X, xv, xrem);
      if (OP) load_rows(Rin, rin, rinrem);
      retire_rows(xv, xrem);
      if (OP) retire_rows(rin, rinrem);
    }
  }
#undef FFD_SLOT_PRELUDE
  if (stamp && threadIdx.x == 0) {
    unsigned long long* o = stamp + 8 * (size_t)blockIdx.x;
    o[0] = st_acc, o[1] = st_acc_rt, o[2] = st_entry, o[3] = st_first_b, o[4] = st_first_e, o[5] = st_epi;
    o[6] = __builtin_amdgcn_s_memrealtime();
    o[7] = (unsigned long long)st_tiles | ((unsigned long long)__smid() << 32);
  }
}

#ifndef FFD_ROWS_EXTRA_D
thread_local int g_ffn_rows = 1;     // 1: row-owning kernel for large M (ffd_tune "ffn_rows"); 0: k_ffn_ln; 2: at every M (tests)
thread_local int g_ffn_rows_nw = 0;  // 0 = heuristic; 4 / 8 / 12 waves per workgroup
thread_local int g_ffn_rows_cps = 0;  // 0 / 2: two chunks per ring slot; 1: one
thread_local int g_ffn_rows_fuse = 1;  // out-projection + LN1 inside the kernel (ffd_tune "ffn_rows_fuse"; two-chunk slots only)

// d_model values with an instance (round 4: 48, 60 -- the reference's class default, score_models.py:31 -- and 64 beside
// 72; the compacted remainder groups are what fits d_model 60's seven of them into the ring)
bool ffn_rows_supported(int D, int F) { return (D == 72 || D == 64 || D == 60 || D == 48) && F % 64 == 0 && F >= 64; }
// large M: where k_ffn_ln ran its 64-row persistent form
bool ffn_rows_selected(int M, int D, int F) {
  return g_ffn_rows && ffn_rows_supported(D, F) && (g_ffn_rows == 2 || cdiv(M, 64) >= 512);
}
// the fused form (out-proj + LN1 + FFN + LN2 in one launch) is taken where k_ffn_rows is, with two-chunk ring slots
bool ffn_rows_fused_selected(int M, int D, int F) {
  return g_ffn_rows_fuse && g_ffn_rows_cps != 1 && ffn_rows_selected(M, D, F);
}

#endif

// y = LayerNorm2(sum over slices of the partial rows, in slice order): one row per 32 lanes (18 of them hold a float4 at
// d_model 72), 8 rows per workgroup.  HBM-bound: (nslice + 1) D 4 bytes per row.
template <int D>
__global__ __launch_bounds__(256) void k_rows_reduce_ln(const float* __restrict__ P, int nslice, int M,
                                                        const float* __restrict__ gam, const float* __restrict__ bet,
                                                        float* __restrict__ Y) {
  static_assert(D % 4 == 0 && D / 4 <= 32, "one float4 per lane of a 32-lane group");
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int j = threadIdx.x & 31;
  if (row >= M) return;
  const bool on = j < D / 4;
  float4 a{0.f, 0.f, 0.f, 0.f};
  if (on) {
    a = *reinterpret_cast<const float4*>(P + (size_t)row * D + 4 * j);
    for (int sidx = 1; sidx < nslice; ++sidx) {
      const float4 q = *reinterpret_cast<const float4*>(P + ((size_t)sidx * M + row) * D + 4 * j);
      a.x += q.x, a.y += q.y, a.z += q.z, a.w += q.w;
    }
  }
  float sum = (a.x + a.y) + (a.z + a.w);
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum * (1.0f / D);
  const float dx = a.x - mean, dy = a.y - mean, dz = a.z - mean, dw = a.w - mean;
  float ss = on ? fmaf(dw, dw, fmaf(dz, dz, fmaf(dy, dy, dx * dx))) : 0.f;
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
  const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
  if (on) {
    const float4 g4 = *reinterpret_cast<const float4*>(gam + 4 * j), e4 = *reinterpret_cast<const float4*>(bet + 4 * j);
    *reinterpret_cast<float4*>(Y + (size_t)row * D + 4 * j) =
        float4{dx * rstd * g4.x + e4.x, dy * rstd * g4.y + e4.y, dz * rstd * g4.z + e4.z, dw * rstd * g4.w + e4.w};
  }
}

#ifndef FFD_ROWS_EXTRA_D
thread_local int g_rows_slices = 0;  // sliced form of the fused kernel: 0 heuristic, -1 off, 2 / 4 / 8 / 16 forced (ffd_tune "rows_slices")

// Mid-size M: (waves per workgroup, slices) of the sliced form, or false where another form is expected to be faster.
// Estimate per launch (us, tools/ffn_rows_sweep.py at d_model 72, F 2048): out-projection slot + chunk slots at the
// pace of NW / 4 waves per SIMD + launch / prologue / tile end, + the reduce launch; a unit per CU at most.
// *unfused_out (round 4): 1 where the slices should NOT recompute the out-projection (k_linear_res_ln once in front, no
// out-projection slot per unit: the better deal where the units are short, i.e. small M and many slices).
thread_local int g_rows_slices_fuse = 0;  // 0 heuristic, 1 fused only, 2 unfused only (ffd_tune "rows_slices_fuse")
bool rows_slice_plan(int M, int D, int F, int* nw_out, int* nslice_out, int* unfused_out) {
  if (g_rows_slices < 0 || !g_ffn_rows || !g_ffn_rows_fuse || g_ffn_rows_cps == 1 || !ffn_rows_supported(D, F)) return false;
  if (g_rows_slices == 0 && ffn_height_plan(M, D, F)) return false;  // (one 32- / 48-row k_ffn_ln tile per CU there)
  const int nslots = F / 64;
  double best = 1e30;
  int bnw = 0, bs = 0, bunf = 0;
  for (int unf = 0; unf <= 1; ++unf)
  for (int nw = 8; nw <= 12; nw += 4) {
    if ((unf == 1 && (g_rows_slices_fuse == 1 || unfused_out == nullptr)) || (unf == 0 && g_rows_slices_fuse == 2 && unfused_out)) continue;
    if (g_ffn_rows_nw && g_ffn_rows_nw != nw) continue;
    const int tiles = cdiv(M, 32 * nw);
    // (measured at d_model 72; the matrix cycles of a slot go with d_model)
    // p_us: what a unit of the fused form pays for its out-projection slot (slot + its barrier + LN1; fitted to the
    // round-4 sweep of fused-only against unfused-only, B = 96 ... 768)
    const double slot_us = (nw == 12 ? 392.0 : 271.0) / 32.0 * (F / 2048.0) * 32.0 / nslots * (D / 72.0), p_us = (nw == 12 ? 14.0 : 10.0) * (D / 72.0);
    const int smax = nslots < 16 ? nslots : 16;
    for (int sl = 2; sl <= smax; ++sl) {
      if (g_rows_slices > 0 && g_rows_slices != sl) continue;
      // g tiles in flight, the tiles x sl units in `rounds` rounds of (out-projection slot + the slice's chunk slots)
      const int g = num_cus() / sl < tiles ? num_cus() / sl : tiles;
      if (g < 1) continue;
      const int rounds = cdiv(tiles, g);
      // (unfused: + one k_linear_res_ln launch, 21.6 us at 95 744 rows, and a launch boundary; no out-projection slot)
      // (+ 5 us of handicap: within that the two forms measure alike, and the fused one is one launch fewer)
      const double t = rounds * ((unf ? 0.0 : p_us) + cdiv(nslots, sl) * slot_us) + 15.0 + 6.0 +
                       1.2e-4 * M * (sl + 1) * D * 4 / 1000.0 + (unf ? 5.0 + 8.0 + 0.15e-3 * M * (D / 72.0) : 0.0);
      if (t < best) best = t, bnw = nw, bs = sl, bunf = unf;
    }
  }
  if (!bs) return false;
  if (g_rows_slices == 0) {
    // what it competes with: the unsliced fused kernel where that is selected (full passes over the CUs), else the
    // F-sliced k_ffn_ln forms + k_linear_res_ln (tools/sweep_mid.py: ~ 30 us + 6 ns per row)
    double alt = 30.0 + 6.0e-3 * M;
    if (ffn_rows_selected(M, D, F)) {
      alt = 1e30;
      for (int nw = 4; nw <= 12; nw += 4) {
        const double loop = (nw == 12 ? 392.0 : nw == 8 ? 271.0 : 146.0) * (D / 72.0);
        alt = fmin(alt, cdiv(cdiv(M, 32 * nw), num_cus()) * (loop + 10.0) + 15.0);
      }
    }
    if (best > 0.97 * alt) return false;
  }
  *nw_out = bnw, *nslice_out = bs;
  if (unfused_out) *unfused_out = bunf;
  return true;
}
size_t rows_slice_floats(int M, int D, int nslice) { return (size_t)nslice * M * D; }

#endif

struct RowsArgs {
  const float *X, *Rin;  // fused: attention output + layer input; else the FFN input (Rin unused)
  const LayerWeights* w;
  float* Y;
  int M, F;
  bool fused;
  unsigned long long* stamp;
  int nslice = 0;  // > 0: the sliced form (Y = the partial rows [nslice][M][D])
};

template <int D, int NW, int CPS, int NSLOT>
static hipError_t launch_rows_cfg(const RowsArgs& a, hipStream_t s) {
  const int R = 32 * NW;
  const int ntiles = cdiv(a.M, R);
  // workgroups per CU: as many rings as fit the 160 KB of LDS and 12 waves
  constexpr int per_cu_lds = (160 * 1024) / (FfnRowsCfg<D, NW, CPS, NSLOT>::LDS_FLOATS * 4);
  constexpr int per_cu = per_cu_lds < 12 / NW ? per_cu_lds : 12 / NW;
  const int slots = per_cu * num_cus();
  const int gsl = a.nslice > 0 ? (slots / a.nslice < ntiles ? slots / a.nslice : ntiles) : 0;  // tiles in flight, sliced form
  const int grid = a.nslice > 0 ? gsl * a.nslice : ntiles < slots ? ntiles : slots;
  const LayerWeights& w = *a.w;
  constexpr int PRV = NW > 4 ? 1 : 0;  // descending wave priority through a barrier interval: with > 1 wave per SIMD
#define FFD_ROWS_LAUNCH2(PR, OP, SL, ONE)                                                                              \
  hipLaunchKernelGGL((k_ffn_rows<D, NW, CPS, NSLOT, PR, OP, SL, ONE>), dim3(grid), dim3(64 * NW), 0, s, a.X, a.Rin,     \
                     w.ring, w.ring_op, w.b2, w.n2w, w.n2b, w.out_b, w.n1w, w.n1b, a.Y, a.M, a.F, a.nslice, a.stamp)
#define FFD_ROWS_LAUNCH(PR, OP, SL)                                                                                    \
  do {                                                                                                                 \
    if constexpr (!(SL)) {                                                                                             \
      if (grid == ntiles) FFD_ROWS_LAUNCH2(PR, OP, false, true); else FFD_ROWS_LAUNCH2(PR, OP, false, false);          \
    } else {                                                                                                           \
      FFD_ROWS_LAUNCH2(PR, OP, true, false);                                                                           \
    }                                                                                                                  \
  } while (0)
  if constexpr (CPS == 2 && NSLOT == 3) {
    if (a.fused) {
      if constexpr (NW >= 8) {
        if (a.nslice > 0) {
          if (gsl < 1 || a.nslice > a.F / 64) return hipErrorInvalidValue;
          FFD_ROWS_LAUNCH(PRV, true, true);
          return hipGetLastError();
        }
      }
      if (a.nslice > 0) return hipErrorInvalidValue;
      FFD_ROWS_LAUNCH(PRV, true, false);
      return hipGetLastError();
    }
  }
  if (a.fused) return hipErrorInvalidValue;
  if (a.nslice > 0) {  // sliced form on x1 rows (k_linear_res_ln ran in front): no out-projection slot per unit
    if constexpr (CPS == 2 && NSLOT == 3 && NW >= 8) {
      if (gsl < 1 || a.nslice > a.F / 64) return hipErrorInvalidValue;
      FFD_ROWS_LAUNCH(PRV, false, true);
      return hipGetLastError();
    }
    return hipErrorInvalidValue;
  }
  FFD_ROWS_LAUNCH(PRV, false, false);
#undef FFD_ROWS_LAUNCH
#undef FFD_ROWS_LAUNCH2
  return hipGetLastError();
}

template <int D>
hipError_t launch_rows_d(const RowsArgs& a, hipStream_t s, int nw_forced) {
  // Waves per workgroup: a tile is 32 NW rows and every CU walks ceil(tiles / CUs) of them at NW / 4 waves per SIMD.
  // Pick the NW with the least estimated time = passes x waves per SIMD / measured main-loop efficiency
  // (tools/ffn_rows_sweep.py at the ECG B = 512 shape: 0.85 / 0.915 / 0.938 of the matrix pipe at 1 / 2 / 3 waves per
  // SIMD); ties go to more waves (the weights are then streamed fewer times).  ffd_tune "ffn_rows_nw" forces it.
  int nw = nw_forced ? nw_forced : g_ffn_rows_nw;
  if (nw != 4 && nw != 8 && nw != 12) {
    const double eff[3] = {0.85, 0.915, 0.938};
    double best = 0.0;
    for (int i = 2; i >= 0; --i) {
      const int cand = 4 * (i + 1);
      const double t = (double)cdiv(cdiv(a.M, 32 * cand), num_cus()) * (i + 1) / eff[i];
      if (best == 0.0 || t < best * 0.999) best = t, nw = cand;
    }
  }
  const int cps = g_ffn_rows_cps == 1 ? 1 : 2;  // 32-unit chunks per ring slot = per barrier (ffd_tune "ffn_rows_cps")
  if constexpr (D == 72) {  // (one-chunk slots: an experiment knob of the d_model 72 instances only)
    if (cps == 1) {
      switch (nw) {
        case 4: return launch_rows_cfg<72, 4, 1, 4>(a, s);
        case 8: return launch_rows_cfg<72, 8, 1, 4>(a, s);
        default: return launch_rows_cfg<72, 12, 1, 4>(a, s);
      }
    }
  }
  switch (nw) {
    case 4: return launch_rows_cfg<D, 4, 2, 3>(a, s);
    case 8: return launch_rows_cfg<D, 8, 2, 3>(a, s);
    default: return launch_rows_cfg<D, 12, 2, 3>(a, s);
  }
}

#ifdef FFD_ROWS_EXTRA_D
template hipError_t launch_rows_d<FFD_ROWS_EXTRA_D>(const RowsArgs&, hipStream_t, int);
#else
// (d_model 64 / 60 / 48: instantiated in ffd_ffn_rows_d64.hip / _d60 / _d48, which compile this file with FFD_ROWS_EXTRA_D set -- in parallel)
extern template hipError_t launch_rows_d<64>(const RowsArgs&, hipStream_t, int);
extern template hipError_t launch_rows_d<60>(const RowsArgs&, hipStream_t, int);
extern template hipError_t launch_rows_d<48>(const RowsArgs&, hipStream_t, int);
static hipError_t launch_rows_any(const RowsArgs& a, int D, hipStream_t s, int nw_forced = 0) {
  if (a.M <= 0) return hipSuccess;
  if (!ffn_rows_supported(D, a.F) || a.w->ring == nullptr || (a.fused && a.w->ring_op == nullptr)) return hipErrorInvalidValue;
  switch (D) {
    case 72: return launch_rows_d<72>(a, s, nw_forced);
    case 64: return launch_rows_d<64>(a, s, nw_forced);
    case 60: return launch_rows_d<60>(a, s, nw_forced);
    case 48: return launch_rows_d<48>(a, s, nw_forced);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_ffn_rows(const float* X, const LayerWeights& w, float* Y, int M, int D, int F, hipStream_t s,
                           unsigned long long* stamp) {
  return launch_rows_any(RowsArgs{X, nullptr, &w, Y, M, F, false, stamp}, D, s);
}

// Y = LN2(x1 + FFN(x1)), x1 = LN1(Rin + Wo attn + bo): cached_transformer.py:316-327 in one launch.  Y must not alias
// attn or Rin (a wave's stores and another wave's loads are not ordered).
hipError_t launch_oproj_ffn_rows(const float* attn, const float* Rin, const LayerWeights& w, float* Y, int M, int D,
                                 int F, hipStream_t s, unsigned long long* stamp) {
  return launch_rows_any(RowsArgs{attn, Rin, &w, Y, M, F, true, stamp}, D, s);
}

// The same for mid-size M as tiles x nslice units + the reduce / LN2 launch; P holds nslice x M x D floats.
hipError_t launch_oproj_ffn_rows_sliced(const float* attn, const float* Rin, const LayerWeights& w, float* P, float* Y,
                                        int M, int D, int F, int nw, int nslice, hipStream_t s) {
  if (!ffn_rows_supported(D, F) || nslice < 2) return hipErrorInvalidValue;
  RowsArgs a{attn, Rin, &w, P, M, F, true, nullptr};
  a.nslice = nslice;
  const hipError_t e = launch_rows_any(a, D, s, nw);
  if (e != hipSuccess) return e;
  switch (D) {
#define X(d) case d: hipLaunchKernelGGL(k_rows_reduce_ln<d>, dim3(cdiv(M, 8)), dim3(256), 0, s, P, nslice, M, w.n2w, w.n2b, Y); break;
    X(72) X(64) X(60) X(48)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// The same slicing on x1 rows (k_linear_res_ln has run): Y = LN2(x1 + FFN(x1)); P holds nslice x M x D floats.
hipError_t launch_ffn_rows_sliced(const float* X1, const LayerWeights& w, float* P, float* Y, int M, int D, int F, int nw,
                                  int nslice, hipStream_t s) {
  if (!ffn_rows_supported(D, F) || nslice < 2) return hipErrorInvalidValue;
  RowsArgs a{X1, nullptr, &w, P, M, F, false, nullptr};
  a.nslice = nslice;
  const hipError_t e = launch_rows_any(a, D, s, nw);
  if (e != hipSuccess) return e;
  switch (D) {
#define X(d) case d: hipLaunchKernelGGL(k_rows_reduce_ln<d>, dim3(cdiv(M, 8)), dim3(256), 0, s, P, nslice, M, w.n2w, w.n2b, Y); break;
    X(72) X(64) X(60) X(48)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
#endif

}  // namespace ffd
