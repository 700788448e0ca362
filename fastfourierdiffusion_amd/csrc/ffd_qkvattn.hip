// Fused in-projection + attention: one workgroup per (sample, head).
//
// Why: with separate kernels the (B,H,L,hd) q/k/v tensors make a round trip through HBM every layer
// (83 MB written + 83 MB read at B=512 ECG): the projection kernel's copy-out alone is write-bandwidth
// bound (21 of its 57 us) and the attention kernel spends ~28 us staging them back.  Here the head's
// q, k, v rows are produced by the matrix cores straight into the LDS images the attention phase reads
// (Q^T [dim][token], K^T [dim][token], V [token][8]) and never exist in global memory.
//
// Phase 1 (projection, v_mfma_f32_16x16x4_f32): D[token][feature] = x[token][:] . W_h[feature][:] for the
// head's 3*hd features (q | k | v rows of in_proj, score_models.py:61-66 -> nn.MultiheadAttention), in one or
// two 16-feature tiles.  x rows go from global memory directly into the A operand -- each lane holds 4
// consecutive k of its token per 16-wide chunk (float4), the weight pack uses the same k permutation --
// and the softmax scale log2(e)/sqrt(hd) is folded into the q rows of the pack.
// Phase 2 (attention): identical to k_attention_pk (ffd_attn.hip): S^T tiles on v_mfma_f32_32x32x2_f32 with the
// stale-max reference riding on contraction dim hd (skipped while it is zero), packed-fp32 softmax / P.V, V rows
// pipelined from LDS.
//
// E2-CRF cache modes (cached_transformer.py:237-305): `n_own` leading tokens take K/V from the projection,
// the rest from the shared tables (kt/vt != NULL); PURE (n_own = 0) runs with the q-only weight pack; in
// MIXED the workgroups of batch element 0 also write their recomputed rows back to the tables
// (caching.py:326-328).
#include <type_traits>

#include "ffd_internal.h"

namespace ffd {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// acc += p.lo * v  /  acc += p.hi * v  with the scalar broadcast done by the instruction's operand select (no v_mov to
// build a {p, p} pair: 16 of the ~93 vector instructions of a 32x32 score tile): p is a pair of adjacent accumulator
// registers of the S^T tile.  The same FMAs, bit for bit.
__device__ __forceinline__ f32x2 pk_fma_lo(f32x2 p, f32x2 v, f32x2 acc) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(p), "v"(v));
  return acc;
}
__device__ __forceinline__ f32x2 pk_fma_hi(f32x2 p, f32x2 v, f32x2 acc) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(p), "v"(v));
  return acc;
}

// Diagnostics (STAMP instances only; ffd_probe_attn): per-wave record of 16 x u64 in a buffer nothing else reads --
// [0] 100 MHz real time at entry, shader-clock stamps [1] entry, [2] projection begin (x requested, weights staged),
// [3] projection end, [4] attention begin (tables / norms done), [5] attention end, [6] exit, then shader cycles summed
// over the wave's key tiles: [7] K fragments + QK^T until the scores are readable, [8] mask + softmax, [9] P.V;
// [10] key tiles walked, [11] hardware id (HW_ID), [12] 100 MHz real time at exit.
#define FFD_STAMP_T() ((unsigned long long)__builtin_amdgcn_s_memtime())

// ---- weight pack ---------------------------------------------------------------------------------
// awp[h][ct][step4][lane][4]: the B operand of k-step (4*step4 + i) for lane (n = lane & 15, q = lane >> 4):
//   full 16-chunks j < D/16 : k = 16 j + 4 q + i          (step4 = j)
//   remainder (D % 16) / 4 steps : k = 16 (D/16) + 4 i + q   (step4 = D/16, i < rem steps)
// feature fi = 16 ct + n -> (reg, e) = (fi / hd, fi % hd) for fi < nf (= 3 hd, or hd for the q-only pack).
__global__ void k_pack_attn(const float* __restrict__ W, const float* __restrict__ b, float* __restrict__ awp,
                            float* __restrict__ abp, int D, int H, int hd, int hpw, int nct, int q_only, float qscale) {
  // one pack per head group (hpw consecutive heads): feature fi -> head hh = fi / fph, then (reg, e)
  const int steps4 = (D + 15) / 16;
  const int NG = H / hpw;
  // q_only is the pack MODE: 0 q | k | v of every head in feature order, 1 q only, 2 (one head per group) tile 0 = k | v,
  // tile 1 = q -- the split small-batch form projects tile 1 for its own q-tiles' tokens only
  const bool kvq = q_only == 2;
  const int fph = q_only == 1 ? hd : 3 * hd;  // features per head
  const int nf = kvq ? 32 : hpw * fph;
  const int total = NG * nct * steps4 * 64 * 4;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int i = idx & 3, lane = (idx >> 2) & 63;
    int rest = idx >> 8;
    const int j = rest % steps4;
    rest /= steps4;
    const int ct = rest % nct, hg = rest / nct;
    const int n = lane & 15, q = lane >> 4;
    const int fi = 16 * ct + n;
    int k;
    if (16 * j + 16 <= D) k = 16 * j + 4 * q + i;
    else k = (4 * i < D - 16 * j) ? 16 * j + 4 * i + q : -1;
    float v = 0.f;
    if (fi < nf && k >= 0 && k < D) {
      int hh = fi / fph, f = fi - hh * fph;
      int reg = f / hd, e = f - reg * hd;
      bool on = true;
      if (kvq) {  // tile 0: k[0 .. hd), v[0 .. hd); tile 1: q[0 .. hd)
        hh = 0, on = n < (ct == 0 ? 2 * hd : hd) && ct < 2;
        reg = ct == 0 ? (n < hd ? 1 : 2) : 0, e = ct == 0 ? (n < hd ? n : n - hd) : n;
      }
      if (on) v = W[(size_t)(reg * D + (hg * hpw + hh) * hd + e) * D + k] * (reg == 0 ? qscale : 1.f);
    }
    awp[idx] = v;
  }
  const int nb = NG * nct * 16;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nb; idx += gridDim.x * blockDim.x) {
    const int n = idx & 15, ct = (idx >> 4) % nct, hg = (idx >> 4) / nct;
    const int fi = 16 * ct + n;
    float v = 0.f;
    if (fi < nf) {
      int hh = fi / fph, f = fi - hh * fph;
      int reg = f / hd, e = f - reg * hd;
      bool on = true;
      if (kvq) {
        hh = 0, on = n < (ct == 0 ? 2 * hd : hd) && ct < 2;
        reg = ct == 0 ? (n < hd ? 1 : 2) : 0, e = ct == 0 ? (n < hd ? n : n - hd) : n;
      }
      if (on) v = b[reg * D + (hg * hpw + hh) * hd + e] * (reg == 0 ? qscale : 1.f);
    }
    abp[idx] = v;
  }
}

static int attn_nct(int hd, int hpw, int q_only) { return q_only == 2 ? 2 : cdiv(hpw * (q_only ? hd : 3 * hd), 16); }
// head dims for which the kv | q pack exists: a head's k and v fit one 16-wide tile and q | k | v need two anyway
bool attn_kvq_supported(int hd) { return 2 * hd <= 16 && 3 * hd > 16; }

size_t attn_pack_floats(int D, int H, int hpw, int q_only) {
  const int nct = attn_nct(D / H, hpw, q_only);
  return (size_t)(H / hpw) * nct * ((D + 15) / 16) * 256 + (size_t)(H / hpw) * nct * 16;
}

hipError_t launch_pack_attn(const float* in_w, const float* in_b, float* pack, int D, int H, int hpw, int q_only,
                            hipStream_t s) {
  const int hd = D / H;
  const int nct = attn_nct(hd, hpw, q_only);
  float* abp = pack + (size_t)(H / hpw) * nct * ((D + 15) / 16) * 256;
  const float qscale = 1.4426950408889634f / sqrtf((float)hd);
  hipLaunchKernelGGL(k_pack_attn, dim3(64), dim3(256), 0, s, in_w, in_b, pack, abp, D, H, hd, hpw, nct, q_only, qscale);
  return hipGetLastError();
}

// max over the wave of a non-negative value: four DPP steps inside each row of 16 lanes, two xor-shuffles across the rows
__device__ __forceinline__ float wave_max_nonneg(float v) {
#define FFD_DPP_MAX(ctrl) \
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true)))
  FFD_DPP_MAX(0xB1);   // quad_perm [1, 0, 3, 2]
  FFD_DPP_MAX(0x4E);   // quad_perm [2, 3, 0, 1]
  FFD_DPP_MAX(0x141);  // row_half_mirror
  FFD_DPP_MAX(0x140);  // row_mirror
#undef FFD_DPP_MAX
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}

// Largest squared norm of the head's (scaled) q rows and of its k rows: nrm[0] keys, nrm[1] queries (as the bit patterns
// of non-negative floats, which order like unsigned integers: reduced inside the wave, then ONE LDS atomic max per wave
// and value -- an atomic per lane serialises on the one address: 24 k cycles of the L = 512 kernel's 140 k when tried).
// |q . k| <= |q| |k| bounds every score of the head; while that bound is within the threshold T the online softmax
// needs no running maximum (its reference stays 0) and the key-tile loop skips the max reduction, the cross-half
// exchange and the refresh test.  The caller zeroes nrm[0..1] before the barrier that precedes this call.
template <int HD>
__device__ __forceinline__ void head_norms(const float* kts, const float* qts, unsigned* nrm, int Lp, int LS, int tid, int nthreads,
                                           int jq0 = 0, int jq1 = 1 << 30) {  // q rows [jq0, jq1) only
  float km = 0.f, qm = 0.f;
  for (int j = tid; j < Lp; j += nthreads) {
    float k2 = 0.f, q2 = 0.f;
    const bool qon = j >= jq0 && j < jq1;
#pragma unroll
    for (int e = 0; e < HD; ++e) {
      const float kv = kts[e * LS + j], qv = qon ? qts[e * LS + j] : 0.f;
      k2 = fmaf(kv, kv, k2), q2 = fmaf(qv, qv, q2);
    }
    km = fmaxf(km, k2), qm = fmaxf(qm, q2);
  }
  km = wave_max_nonneg(km), qm = wave_max_nonneg(qm);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(nrm, __float_as_uint(km));
    atomicMax(nrm + 1, __float_as_uint(qm));
  }
}

// ---- the kernel ----------------------------------------------------------------------------------
//
// SPLIT (small batches, e.g. the benchmark_cache.py harness at batch 1: B*H workgroups would leave most of the chip
// idle and every wave walking 3 q-tiles x 6 key tiles in sequence): `qsplit` workgroups per (sample, head), each
// projecting the whole head but attending only nwaves/kspl q-tiles, with the key range of a q-tile cut into `kspl`
// pieces over the waves (flash-decoding).  The pieces (reference exponent, row sum, unnormalised output) meet in
// LDS and are merged in piece order, so the result does not depend on timing.
// KVQ (split form with the kv | q pack, NCT = 2): tile 0 holds the head's k and v features, tile 1 its q features, and a
// workgroup projects tile 1 only for the tokens of its own q-tiles -- the other q-tiles' workgroups project theirs.
template <int D, int HD, int QG, int NCT, bool SPLIT = false, bool STAMP = false, bool KVQ = false>
__global__ __launch_bounds__(256, (SPLIT && NCT >= 2 ? 2 : QG <= 2 ? 4 : 3)) void k_qkv_attention(
    const float* __restrict__ x, const float* __restrict__ awp, const float* __restrict__ kt,
    const float* __restrict__ vt, float* __restrict__ kt_out, float* __restrict__ vt_out, float* __restrict__ out,
    int B, int L, int n_own, int q_only, int qsplit, int kspl, unsigned long long* __restrict__ stamp) {
  unsigned long long st_t[7] = {0, 0, 0, 0, 0, 0, 0}, st_qk = 0, st_sm = 0, st_pv = 0, st_n = 0;
  if constexpr (STAMP) st_t[0] = __builtin_amdgcn_s_memrealtime(), st_t[1] = FFD_STAMP_T();
  constexpr int H = D / HD;
  constexpr int KST = (HD + 1) / 2;
  constexpr int KSX = (HD + 2) / 2;
  constexpr int SX = HD / 2;
  constexpr int HX = HD & 1;
  constexpr int HP = (HD + 1) / 2;
  constexpr int C16 = D / 16;           // full 16-wide k chunks
  constexpr int REM = (D % 16) / 4;     // remaining k-steps (k = 16*C16 + 4 i + q)
  constexpr int S4 = (D + 15) / 16;     // float4 groups of packed weight per (h, ct)
  constexpr float T = 64.0f;  // scores (log2 domain) may sit this far from the reference before it is refreshed
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.
  // All H heads of a sample read the same x rows, so they are placed on one XCD: x is fetched into one L2
  // once instead of into (up to) eight.
  int b, h, qs = 0;
  {
    int pair = blockIdx.x;
    if constexpr (SPLIT) qs = pair % qsplit, pair /= qsplit;
    const int nmain = (B >> 3) * 8 * H;
    if (pair < nmain) {
      const int xcd = pair & 7, slot = pair >> 3;
      const int sb = slot / H;
      b = sb * 8 + xcd, h = slot - sb * H;
    } else {
      b = pair / H, h = pair - b * H;
    }
  }
  const int KT = (L + 31) >> 5;
  const int Lp = KT * 32;
  // row stride of the LDS images: a compile-time 516 in the four-q-tiles-per-wave instance (launched for KT = 16 only), so
  // that its LDS addresses are per-lane bases + immediate offsets; the run-time Lp elsewhere
  // (+ 4: a stride that is a multiple of 64 floats puts the 16 feature rows a projection tile stores into the same banks)
  const int LS = (QG == 4 && !SPLIT) ? 516 : Lp + 4;
  float* vs = lds;                               // V   [Lp][8]
  float* kts = vs + (size_t)LS * 8;              // K^T [2*KST][LS]
  float* qts = kts + (size_t)2 * KST * LS;       // Q^T [2*KST][LS]   (already scaled by log2(e)/sqrt(hd))
  const int half = lane >> 5, l31 = lane & 31;
  unsigned* nrm = reinterpret_cast<unsigned*>(lds + (size_t)LS * (8 + 4 * KST) + (SPLIT ? 4 * 32 * (2 + 2 * HP) : 0));  // see head_norms
  if (threadIdx.x < 2) nrm[threadIdx.x] = 0u;  // (ordered before head_norms by the barrier behind the projection)

  if constexpr (HD % 2 == 1) {  // odd head dims read one pad row / pad column: keep them zero
    for (int idx = threadIdx.x; idx < LS * (8 + 4 * KST); idx += blockDim.x) vs[idx] = 0.f;
    __syncthreads();
  }

  // ------------------------------------------------------------------ phase 1: projection
  {
    const int n = lane & 15, qq = lane >> 4;
    // One head of head_dim 6 is 18 features = one 16-wide tile + 2 (v[4], v[5]).  A second tile for those two would be
    // seven eighths padding (576 matrix cycles per 16 tokens); instead every lane, which holds 18 of its token's 72 x
    // values as A operand anyway, forms their partial dot products with the vector ALU (36 FMAs) and the four lanes of
    // a token add theirs up (two xor-shuffles): ~180 cycles.  k order differs from the MFMA chain: rounding-level, in
    // these two features only.
    // (instances with >= 3 q-tiles per wave only: the others are built for 4 waves per SIMD = 128 registers, where the
    //  36 weight registers of the two columns would spill)
    constexpr bool VREM = HD == 6 && NCT == 2 && QG >= 3;
    constexpr int NCTM = VREM ? 1 : NCT;  // feature tiles on the matrix core
    float4 wf[NCTM][S4];
    const float4* Wq = reinterpret_cast<const float4*>(awp) + (size_t)h * NCT * S4 * 64;
#pragma unroll
    for (int ct = 0; ct < NCTM; ++ct)
#pragma unroll
      for (int j = 0; j < S4; ++j) wf[ct][j] = Wq[((size_t)ct * S4 + j) * 64 + lane];
    float4 w16[VREM ? C16 : 1], w17[VREM ? C16 : 1];  // the second tile's columns n = 0, 1 for this lane's k subset (qq)
    float2 w16r = {0.f, 0.f}, w17r = {0.f, 0.f};      // ... and its <= 2 remainder k-steps
    const float* abp = awp + (size_t)H * NCT * S4 * 256 + (size_t)h * NCT * 16;
    float b16 = 0.f, b17 = 0.f;
    if constexpr (VREM) {
#pragma unroll
      for (int j = 0; j < C16; ++j) w16[j] = Wq[((size_t)S4 + j) * 64 + 16 * qq], w17[j] = Wq[((size_t)S4 + j) * 64 + 16 * qq + 1];
      static_assert(!VREM || REM <= 2, "remainder k-steps of the two vector-ALU columns");
      if constexpr (REM > 0) {
        w16r = *reinterpret_cast<const float2*>(&Wq[((size_t)S4 + S4 - 1) * 64 + 16 * qq]);
        w17r = *reinterpret_cast<const float2*>(&Wq[((size_t)S4 + S4 - 1) * 64 + 16 * qq + 1]);
      }
      b16 = abp[16], b17 = abp[17];
    }
    float bias[NCTM];
    // where this lane's feature (16 ct + n) goes, worked out once: float index of token 0 in the LDS images -- Q^T / K^T
    // rows take a lane's four tokens as one float4, V rows ([token][8]) as four scalars 8 floats apart; -1 = no feature
    int sbase[NCTM];
    bool sv[NCTM];
#pragma unroll
    for (int ct = 0; ct < NCTM; ++ct) {
      bias[ct] = abp[ct * 16 + n];
      const int fi = 16 * ct + n;
      int reg = fi / HD, e = fi - reg * HD;
      int kind = q_only ? (fi < HD ? 0 : 3) : (reg > 2 ? 3 : reg);  // 0 q, 1 k, 2 v, 3 none
      if constexpr (KVQ) {
        static_assert(!KVQ || (SPLIT && NCT == 2 && 2 * HD <= 16), "kv | q pack");
        kind = ct == 0 ? (n < HD ? 1 : n < 2 * HD ? 2 : 3) : (n < HD ? 0 : 3);
        e = ct == 0 ? (n < HD ? n : n - HD) : n;
      }
      sv[ct] = kind == 2;
      sbase[ct] = kind == 3 ? -1 : kind == 0 ? LS * 8 + 2 * KST * LS + e * LS : kind == 1 ? LS * 8 + e * LS : e;
    }
    const float* xb = x + (size_t)b * L * D;
    const int TT = Lp >> 4;  // every row of the LDS images gets a (finite) value
    // token tiles [tt0, tt1) this workgroup projects: all of them -- except on a pure cache hit in the split form, where
    // K and V come from the tables and a workgroup needs q for ITS q-tiles only (the other workgroups of the head
    // project theirs): 2 of 12 tiles at batch 1
    int tt0 = 0, tt1 = TT;
    int ot0 = 0, ot1 = TT;  // token tiles of this workgroup's own q-tiles (KVQ: only they get tile 1 = q)
    if constexpr (SPLIT) {
      const int qpw = nwaves / kspl;
      ot0 = min(2 * qs * qpw, TT), ot1 = min(ot0 + 2 * qpw, TT);
      if (q_only) tt0 = ot0, tt1 = ot1;
    }
    auto load_x = [&](int tt, float4(&xa)[C16 > 0 ? C16 : 1], float(&xr)[REM > 0 ? REM : 1]) {
      int tok = 16 * tt + n;
      if (tok >= L) tok = L - 1;  // padded tokens repeat the last row: finite values that are masked / never stored
      const float* xp = xb + (size_t)tok * D;
#pragma unroll
      for (int j = 0; j < C16; ++j) xa[j] = *reinterpret_cast<const float4*>(xp + 16 * j + 4 * qq);
#pragma unroll
      for (int i = 0; i < REM; ++i) xr[i] = xp[16 * C16 + 4 * i + qq];
    };
    // x tiles are requested PFX tiles ahead: one tile's MFMAs take ~0.5 us, a global load 1-2 us
    constexpr int PFX = VREM ? 3 : 4;  // (three with the two vector-ALU columns' weights in registers: no scratch)
    float4 xa[PFX][C16 > 0 ? C16 : 1];
    float xr[PFX][REM > 0 ? REM : 1];
#pragma unroll
    for (int u = 0; u < PFX; ++u)  // unconditional (tile index clamped): conditional loads would force vmcnt(0) waits
      load_x(min(tt0 + wave + u * nwaves, TT - 1), xa[u], xr[u]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STAMP) st_t[2] = FFD_STAMP_T();
    // Straight-line code (full unroll, forward exits only): a loop back-edge would make the compiler wait for
    // *all* outstanding loads at every tile (vmcnt(0)), which defeats the ring.  MAXT tiles per wave cover
    // L <= 512 with 4 waves.
    constexpr int MAXT = 8;
#pragma unroll
    for (int it = 0; it < MAXT; ++it) {
      const int u = it % PFX;
      const int tt = tt0 + wave + it * nwaves;
      if (tt >= tt1) break;
      f32x4 acc[NCTM];
#pragma unroll
      for (int ct = 0; ct < NCTM; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool own = !KVQ || (tt >= ot0 && tt < ot1);  // (wave-uniform) the q tile of the kv | q pack: own tokens only
      auto tile_mfmas = [&](auto CTC) {
        constexpr int ct = decltype(CTC)::value;
#pragma unroll
        for (int j = 0; j < C16; ++j) {
          acc[ct] = mfma16(xa[u][j].x, wf[ct][j].x, acc[ct]);
          acc[ct] = mfma16(xa[u][j].y, wf[ct][j].y, acc[ct]);
          acc[ct] = mfma16(xa[u][j].z, wf[ct][j].z, acc[ct]);
          acc[ct] = mfma16(xa[u][j].w, wf[ct][j].w, acc[ct]);
        }
#pragma unroll
        for (int i = 0; i < REM; ++i) {
          const float4 w4 = wf[ct][S4 - 1];
          const float wv = i == 0 ? w4.x : i == 1 ? w4.y : i == 2 ? w4.z : w4.w;
          acc[ct] = mfma16(xr[u][i], wv, acc[ct]);
        }
      };
      if constexpr (KVQ) {
        tile_mfmas(std::integral_constant<int, 0>{});
        if (own) tile_mfmas(std::integral_constant<int, 1>{});
      } else {
#pragma unroll
        for (int j = 0; j < C16; ++j) {
#pragma unroll
          for (int ct = 0; ct < NCTM; ++ct) {
            acc[ct] = mfma16(xa[u][j].x, wf[ct][j].x, acc[ct]);
            acc[ct] = mfma16(xa[u][j].y, wf[ct][j].y, acc[ct]);
            acc[ct] = mfma16(xa[u][j].z, wf[ct][j].z, acc[ct]);
            acc[ct] = mfma16(xa[u][j].w, wf[ct][j].w, acc[ct]);
          }
        }
#pragma unroll
        for (int i = 0; i < REM; ++i) {
#pragma unroll
          for (int ct = 0; ct < NCTM; ++ct) {
            const float4 w4 = wf[ct][S4 - 1];
            const float wv = i == 0 ? w4.x : i == 1 ? w4.y : i == 2 ? w4.z : w4.w;
            acc[ct] = mfma16(xr[u][i], wv, acc[ct]);
          }
        }
      }
      float p16 = 0.f, p17 = 0.f;
      if constexpr (VREM) {  // features 16, 17 of token n over this lane's k subset
#pragma unroll
        for (int j = 0; j < C16; ++j) {
          p16 = fmaf(xa[u][j].x, w16[j].x, p16), p17 = fmaf(xa[u][j].x, w17[j].x, p17);
          p16 = fmaf(xa[u][j].y, w16[j].y, p16), p17 = fmaf(xa[u][j].y, w17[j].y, p17);
          p16 = fmaf(xa[u][j].z, w16[j].z, p16), p17 = fmaf(xa[u][j].z, w17[j].z, p17);
          p16 = fmaf(xa[u][j].w, w16[j].w, p16), p17 = fmaf(xa[u][j].w, w17[j].w, p17);
        }
#pragma unroll
        for (int i = 0; i < REM; ++i) {
          p16 = fmaf(xr[u][i], i == 0 ? w16r.x : w16r.y, p16), p17 = fmaf(xr[u][i], i == 0 ? w17r.x : w17r.y, p17);
        }
      }
      if (it + PFX < MAXT) load_x(min(tt + PFX * nwaves, TT - 1), xa[u], xr[u]);  // refill this slot (clamped, unconditional)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (VREM) {  // the token's four lanes (k subsets qq = 0 .. 3, 16 lanes apart) add up; qq = 0 stores v[4], v[5]
        p16 += __shfl_xor(p16, 16), p17 += __shfl_xor(p17, 16);
        p16 += __shfl_xor(p16, 32), p17 += __shfl_xor(p17, 32);
        if (qq == 0) *reinterpret_cast<float2*>(vs + (size_t)(16 * tt + n) * 8 + 4) = float2{p16 + b16, p17 + b17};
      }
      const int t0 = 16 * tt + 4 * qq;  // D: lane holds tokens t0 .. t0+3 of feature 16 ct + n
#pragma unroll
      for (int ct = 0; ct < NCTM; ++ct) {
        const float4 o = float4{acc[ct][0] + bias[ct], acc[ct][1] + bias[ct], acc[ct][2] + bias[ct],
                                acc[ct][3] + bias[ct]};
        if (sbase[ct] >= 0 && (ct == 0 || own)) {
          if (!sv[ct]) {
            *reinterpret_cast<float4*>(lds + sbase[ct] + t0) = o;
          } else {
            float* vp = lds + sbase[ct] + t0 * 8;
            vp[0] = o.x, vp[8] = o.y, vp[16] = o.z, vp[24] = o.w;
          }
        }
      }
    }
  }
  if constexpr (STAMP) st_t[3] = FFD_STAMP_T();
  // rows served by the shared tables (PURE: all of them; MIXED: tokens >= n_own) overwrite / fill K^T and V
  if (kt != nullptr) {
    if (!q_only && n_own > 0) __syncthreads();  // MIXED: the projection wrote these rows first
    const float* ktab = kt + (size_t)h * L * HD;
    const float* vtab = vt + (size_t)h * L * HD;
    for (int j = n_own + threadIdx.x; j < L; j += blockDim.x) {
      float kx[HD], vx[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vx[e] = 0.f;
      const float* kp = ktab + (size_t)j * HD;
      const float* vp = vtab + (size_t)j * HD;
      if constexpr (HD % 2 == 0) {
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
#pragma unroll
      for (int e = 0; e < HD; ++e) kts[e * LS + j] = kx[e];
      *reinterpret_cast<float4*>(vs + (size_t)j * 8) = float4{vx[0], vx[1], vx[2], vx[3]};
      *reinterpret_cast<float4*>(vs + (size_t)j * 8 + 4) = float4{vx[4], vx[5], vx[6], vx[7]};
    }
    if (q_only) {  // PURE: key rows in [L, Lp) were never written; they are masked but must be finite
      for (int j = L + threadIdx.x; j < Lp; j += blockDim.x) {
#pragma unroll
        for (int e = 0; e < HD; ++e) kts[e * LS + j] = 0.f;
        *reinterpret_cast<float4*>(vs + (size_t)j * 8) = float4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<float4*>(vs + (size_t)j * 8 + 4) = float4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  __syncthreads();
  {  // (q rows outside the projected token range were never written: they stay out of the bound)
    int jq0 = 0, jq1 = Lp;
    if constexpr (SPLIT) {
      if (q_only || KVQ) {
        const int qpw = nwaves / kspl;
        jq0 = min(32 * qs * qpw, Lp), jq1 = min(jq0 + 32 * qpw, Lp);
      }
    }
    head_norms<HD>(kts, qts, nrm, Lp, LS, threadIdx.x, blockDim.x, jq0, jq1);
  }
  // MIXED: batch element 0 publishes its recomputed rows (caching.py:326-328, cached_transformer.py:301-305)
  if (kt_out != nullptr && b == 0 && qs == 0) {
    for (int idx = threadIdx.x; idx < n_own * HD; idx += blockDim.x) {
      const int j = idx / HD, e = idx - j * HD;
      kt_out[(size_t)h * L * HD + idx] = kts[e * LS + j];
      vt_out[(size_t)h * L * HD + idx] = vs[(size_t)j * 8 + e];
    }
  }

  __syncthreads();  // the tile norms
  if constexpr (STAMP) st_t[4] = FFD_STAMP_T();
  // every score of this head is within +- sqrt(max |q|^2 max |k|^2) (wave-uniform)
  const bool head_bounded =
      __builtin_amdgcn_ballot_w64(__uint_as_float(nrm[1]) * __uint_as_float(nrm[0]) <= T * T) != 0;
  // ------------------------------------------------------------------ phase 2: attention (see k_attention_pk)
  const bool xlane = half == HX;
  constexpr int PF = 4;
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
  const int d = D;
  // q-tiles of this wave: qt0 = q_first, q_first + q_step, ... < q_end; key tiles [t_lo, t_hi)
  int q_first = wave * QG, q_step = nwaves * QG, q_end = QT, t_lo = 0, t_hi = KT;
  if constexpr (SPLIT) {  // one (q-tile, key piece) per wave; exactly one trip so that every wave reaches the merge
    const int qpw = nwaves / kspl, kps = (KT + kspl - 1) / kspl;
    q_first = qs * qpw + wave / kspl, q_step = 1, q_end = q_first + 1;
    t_lo = (wave % kspl) * kps, t_hi = min(KT, t_lo + kps);
    if (q_first >= QT) t_hi = t_lo;  // ragged last workgroup: an empty piece
  }
  for (int qt0 = q_first; qt0 < q_end; qt0 += q_step) {
    float qf[QG][KSX], mref[QG];
    bool ref_on = false;  // wave-uniform: some lane of this wave carries a non-zero reference
    bool acc_empty[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) acc_empty[g] = true;
    f32x2 lsum[QG], acc[QG][HP];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      const int qtile = (qt0 + g < QT) ? qt0 + g : QT - 1;
#pragma unroll
      for (int s = 0; s < KSX; ++s) {
        const int e = 2 * s + half;
        qf[g][s] = (e < HD) ? qts[e * LS + 32 * qtile + l31] : 0.f;  // dim HD starts at -m_ref = 0
      }
      mref[g] = 0.f;
      lsum[g] = f32x2{0.f, 0.f};
#pragma unroll
      for (int e = 0; e < HP; ++e) acc[g][e] = f32x2{0.f, 0.f};
    }
#pragma unroll 1
    for (int t = t_lo; t < t_hi; ++t) {
      unsigned long long st_a = 0;
      if constexpr (STAMP) st_a = FFD_STAMP_T();
      float kf[KSX];
#pragma unroll
      for (int s = 0; s < KSX; ++s) {
        const int e = 2 * s + half;
        kf[s] = (s < KST && (2 * s + 1 < HD || half == 0)) ? kts[e * LS + 32 * t + l31] : 0.f;
      }
      if (xlane) kf[SX] = 1.0f;
      f32x16 sc[QG];
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KSX; ++s) {
          if (HD % 2 == 0 && s == SX && !ref_on) continue;  // even hd: that step carries nothing but -m_ref = 0
          z = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[g][s], z, 0, 0, 0);
        }
        sc[g] = z;
      }
      const int kbase = 32 * t + 4 * half;
      f32x2 vb[PF][4];
#pragma unroll
      for (int r = 0; r < PF; ++r) load_v(r, kbase, vb[r]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (STAMP) {  // the stamp follows an instruction that reads the last score tile: QK^T has retired
        asm volatile("v_mov_b32 %0, %0" : "+v"(sc[QG - 1][15]));
        const unsigned long long n = FFD_STAMP_T();
        st_qk += n - st_a, st_a = n, ++st_n;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (32 * t + 32 > L) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          if (32 * t + 8 * r4 + 8 > L) {  // (uniform) registers 4 r4 .. 4 r4 + 3 hold key rows 8 r4 .. 8 r4 + 7 of the tile
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const bool dead = kbase + rr + 8 * r4 >= L;
#pragma unroll
              for (int g = 0; g < QG; ++g) sc[g][4 * r4 + rr] = dead ? -INFINITY : sc[g][4 * r4 + rr];
            }
          }
        }
      }
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        // while the head's bound stays within T and no lane of the wave uses a reference, nothing below can trigger
        if (ref_on || !head_bounded) {
        float bm = __builtin_fmaxf(__builtin_fmaxf(sc[g][0], sc[g][1]), sc[g][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) bm = __builtin_fmaxf(__builtin_fmaxf(bm, sc[g][r]), sc[g][r + 1]);
        bm = __builtin_fmaxf(bm, sc[g][15]);
        const float bmx = fmaxf(bm, __shfl_xor(bm, 32));
        // m_ref starts at 0 and usually stays there: |scores| <= 64 (log2 domain) neither overflow nor lose the row
        // to underflow, and the factor 2^-m_ref cancels in the normalisation whatever it is.
        const bool first = acc_empty[g];  // nothing accumulated yet: this is the row's first key tile
        const bool refresh = first ? (fabsf(bmx) > T) : (bmx > T);
        if (__builtin_amdgcn_ballot_w64(refresh) != 0) ref_on = true;
        if (refresh) {
          const float delta = bmx;
          mref[g] += delta;
          if (!first) {
            const float corr = __builtin_amdgcn_exp2f(-delta);
            lsum[g] *= corr;
#pragma unroll
            for (int e = 0; e < HP; ++e) acc[g][e] *= corr;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[g][r] -= delta;
          if (xlane) qf[g][SX] = -mref[g];
        }
        }
        acc_empty[g] = false;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(sc[g][r]);
          const float p1 = __builtin_amdgcn_exp2f(sc[g][r + 1]);
          sc[g][r] = p0;
          sc[g][r + 1] = p1;
          lsum[g] += f32x2{p0, p1};
        }
      }
      if constexpr (STAMP) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long n = FFD_STAMP_T();
        st_sm += n - st_a, st_a = n;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f32x2 vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = vb[r % PF][e];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
          const f32x2 pp = f32x2{sc[g][r & ~1], sc[g][(r & ~1) + 1]};  // (adjacent registers of the accumulator)
#pragma unroll
          for (int e = 0; e < HP; ++e) acc[g][e] = (r & 1) ? pk_fma_hi(pp, vv[e], acc[g][e]) : pk_fma_lo(pp, vv[e], acc[g][e]);
        }
        if (r + PF < 16) {
          load_v(r + PF, kbase, vb[r % PF]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (STAMP) {
        __builtin_amdgcn_sched_barrier(0);
        st_pv += FFD_STAMP_T() - st_a;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (STAMP) st_t[5] = FFD_STAMP_T();
    if constexpr (SPLIT) {
      constexpr int PS = 2 + 2 * HP;  // per query row: reference exponent, row sum, unnormalised output
      float* part = lds + (size_t)LS * (8 + 4 * KST);  // [wave][32][PS]
      float l = lsum[0].x + lsum[0].y;
      l += __shfl_xor(l, 32);
      float o[2 * HP];
#pragma unroll
      for (int e = 0; e < HP; ++e) {
        float a0 = acc[0][e].x, a1 = acc[0][e].y;
        o[2 * e] = a0 + __shfl_xor(a0, 32), o[2 * e + 1] = a1 + __shfl_xor(a1, 32);
      }
      if (half == 0) {
        float* pw = part + (size_t)(wave * 32 + l31) * PS;
        pw[0] = (t_lo < t_hi) ? mref[0] : -INFINITY;  // an empty piece weighs 2^-inf = 0 in the merge
        pw[1] = l;
#pragma unroll
        for (int e = 0; e < 2 * HP; ++e) pw[2 + e] = o[e];
      }
      __syncthreads();
      const int qpw = nwaves / kspl;
      for (int idx = threadIdx.x; idx < qpw * 32 * HD; idx += blockDim.x) {
        const int e = idx % HD, ql = (idx / HD) & 31, qi = idx / (HD * 32);
        const int qtile = qs * qpw + qi, q = 32 * qtile + ql;
        if (qtile >= QT || q >= L) continue;
        const float* p0 = part + (size_t)(qi * kspl * 32 + ql) * PS;
        float mx = -INFINITY;
        for (int sp = 0; sp < kspl; ++sp) mx = fmaxf(mx, p0[(size_t)sp * 32 * PS]);
        float lt = 0.f, ot = 0.f;
        for (int sp = 0; sp < kspl; ++sp) {
          const float* pp = p0 + (size_t)sp * 32 * PS;
          const float wgt = __builtin_amdgcn_exp2f(pp[0] - mx);
          lt = fmaf(pp[1], wgt, lt), ot = fmaf(pp[2 + e], wgt, ot);
        }
        out[((size_t)b * L + q) * d + h * HD + e] = ot / lt;
      }
    } else {
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      float l = lsum[g].x + lsum[g].y;
      l += __shfl_xor(l, 32);
      const float inv = __builtin_amdgcn_rcpf(l);  // (1 ulp; the IEEE division sequence is ten vector instructions per q-tile)
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
  if constexpr (STAMP) {
    if (lane == 0 && stamp != nullptr) {
      unsigned long long* r = stamp + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 16;
      st_t[6] = FFD_STAMP_T();
#pragma unroll
      for (int i = 0; i < 7; ++i) r[i] = st_t[i];
      r[7] = st_qk, r[8] = st_sm, r[9] = st_pv, r[10] = st_n, r[11] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
      r[12] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// ---- multi-head variant ------------------------------------------------------------------------------
// HPW heads of one sample per workgroup (2 waves per head).  Why: with one head per workgroup every
// workgroup re-reads the sample's x rows and its own weight slice, and the kernel is paced by the L1 request
// rate at workgroup start (measured: ~20k cycles until a wave's second tile vs ~2.3k per tile afterwards).
// Here x is read once for HPW heads, the HPW*3*hd features are packed densely into 16-wide tiles
// (2 heads x 18 = 36 features -> 3 tiles, 75 % useful, vs 18 -> 2 tiles, 56 %), and the weight pack is
// staged once in LDS and read as the MFMA B operand from there.  Two heads = 4 waves, one per SIMD: a 6-wave
// workgroup (3 heads) leaves the CU with a single resident workgroup (two of its waves land on SIMDs 0 and 1,
// so a second one never fits at 3 waves per SIMD) and was slower.
template <int D, int HD, int HPW, int QG, int NCT, bool QO, bool STAMP = false>
__global__ __launch_bounds__(128 * HPW, 3) void k_qkv_attention_mh(
    const float* __restrict__ x, const float* __restrict__ awp, const float* __restrict__ kt,
    const float* __restrict__ vt, float* __restrict__ kt_out, float* __restrict__ vt_out, float* __restrict__ out,
    int B, int L, int n_own, int q_only, unsigned long long* __restrict__ stamp) {
  unsigned long long st_t[7] = {0, 0, 0, 0, 0, 0, 0}, st_qk = 0, st_sm = 0, st_pv = 0, st_n = 0;
  if constexpr (STAMP) st_t[0] = __builtin_amdgcn_s_memrealtime(), st_t[1] = FFD_STAMP_T();
  constexpr int H = D / HD;
  constexpr int NG = H / HPW;
  constexpr int KST = (HD + 1) / 2;
  constexpr int KSX = (HD + 2) / 2;
  constexpr int SX = HD / 2;
  constexpr int HX = HD & 1;
  constexpr int HP = (HD + 1) / 2;
  constexpr int C16 = D / 16;
  constexpr int REM = (D % 16) / 4;
  constexpr int S4 = (D + 15) / 16;
  constexpr int NW = 2 * HPW;           // waves per workgroup
  constexpr int MAXT = 3;               // token tiles per wave: Lp <= 192 -> 12 tiles over >= 4 waves
  // Two heads of head_dim 6 are 36 features: two 16-wide tiles + 4 (head 1's v[2..5]).  A third tile would be three
  // quarters padding; the four features are vector-ALU dot products instead (each lane holds 18 of its token's 72 x
  // values as A operand anyway: 4 x 18 FMAs + two xor-shuffles per 16 tokens, like k_qkv_attention's two).  Round 3 ran
  // them on v_mfma_f32_4x4x1_16b_f32 (lane = token), which needed every x row a second time in that layout (55 KB more
  // L2 reads per workgroup, 72 registers): 100.4 -> 99.8 us at ECG B = 512 without it.
  constexpr bool REMV = HD == 6 && HPW == 2 && NCT == 3;
  constexpr int NCTM = REMV ? 2 : NCT;  // 16-wide feature tiles on the 16x16x4 form
  constexpr float T = 64.0f;  // scores (log2 domain) may sit this far from the reference before it is refreshed
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int b, hg;
  {
    const int pair = blockIdx.x;
    const int nmain = (B >> 3) * 8 * NG;
    if (pair < nmain) {  // all head groups of a sample on one XCD (see k_qkv_attention)
      const int xcd = pair & 7, slot = pair >> 3;
      const int sb = slot / NG;
      b = sb * 8 + xcd, hg = slot - sb * NG;
    } else {
      b = pair / NG, hg = pair - b * NG;
    }
  }
  const int KT = (L + 31) >> 5;
  const int Lp = KT * 32;
  // the LDS images have a COMPILE-TIME row stride (a wave owns QG q-tiles, a head two waves: Lp <= LS): every LDS address
  // below is one per-lane base + immediate offsets instead of multiplications by a run-time Lp
  constexpr int LS = 64 * QG + 4;  // (+ 4: see k_qkv_attention)
  constexpr int RS = LS * (8 + 4 * KST);                       // floats per head region: V | K^T | Q^T
  float4* wl = reinterpret_cast<float4*>(lds + (size_t)HPW * RS);  // weight pack [NCT][S4][64] float4
  const int half = lane >> 5, l31 = lane & 31;
  constexpr int fph = QO ? HD : 3 * HD;                        // features per head in this pack (QO == q_only)
  if (threadIdx.x < 2 * HPW)  // the heads' norm words (ordered before head_norms by the barriers of the projection)
    reinterpret_cast<unsigned*>(lds + (size_t)HPW * RS + (size_t)NCT * S4 * 256)[threadIdx.x] = 0u;

  // ------------------------------------------------------------------ phase 1: projection
  {
    const int n = lane & 15, qq = lane >> 4;
    const float* xb = x + (size_t)b * L * D;
    const int TT = Lp >> 4;
    auto load_x = [&](int tt, float4(&xa)[C16 > 0 ? C16 : 1], float(&xr)[REM > 0 ? REM : 1]) {
      int tok = 16 * tt + n;
      if (tok >= L) tok = L - 1;
      const float* xp = xb + (size_t)tok * D;
#pragma unroll
      for (int j = 0; j < C16; ++j) xa[j] = *reinterpret_cast<const float4*>(xp + 16 * j + 4 * qq);
#pragma unroll
      for (int i = 0; i < REM; ++i) xr[i] = xp[16 * C16 + 4 * i + qq];
    };
    // every x tile of this wave is requested up front (unconditional, clamped), then the weights are staged
    float4 xa[MAXT][C16 > 0 ? C16 : 1];
    float xr[MAXT][REM > 0 ? REM : 1];
#pragma unroll
    for (int u = 0; u < MAXT; ++u) load_x(min(wave + u * NW, TT - 1), xa[u], xr[u]);
    {
      const float4* Wq = reinterpret_cast<const float4*>(awp) + (size_t)hg * NCT * S4 * 64;
      for (int i = threadIdx.x; i < NCT * S4 * 64; i += 64 * NW) wl[i] = Wq[i];
      if constexpr (HD % 2 == 1) {  // odd head dims read one pad row / pad column: keep them zero
        for (int idx = threadIdx.x; idx < HPW * RS; idx += 64 * NW) lds[idx] = 0.f;
      }
    }
    const float* abp = awp + (size_t)NG * NCT * S4 * 256 + (size_t)hg * NCT * 16;
    float bias[NCTM];
    // where this lane's feature (16 ct + n) goes, worked out once: float index of token 0 in the LDS images -- Q^T / K^T
    // rows take a lane's four tokens as one float4, V rows ([token][8]) as four scalars 8 floats apart; -1 = no feature
    int sbase[NCTM];
    bool sv[NCTM];
#pragma unroll
    for (int ct = 0; ct < NCTM; ++ct) {
      bias[ct] = abp[ct * 16 + n];
      const int fi = 16 * ct + n;
      const int hh = fi / fph, f = fi - hh * fph;
      const int reg = f / HD, e = f - reg * HD;
      sv[ct] = reg == 2;
      sbase[ct] = hh >= HPW ? -1
                  : hh * RS + (reg == 0 ? LS * 8 + 2 * KST * LS + e * LS : reg == 1 ? LS * 8 + e * LS : e);
    }
    __syncthreads();
    if constexpr (STAMP) st_t[2] = FFD_STAMP_T();
    // columns n = 0 .. 3 of the pack's third tile for this lane's k subset (qq), from the staged pack
    float4 wv4[REMV ? 4 : 1][REMV ? C16 : 1];
    float2 wvr[REMV ? 4 : 1];
    float bvv[REMV ? 4 : 1];
    if constexpr (REMV) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int j = 0; j < C16; ++j) wv4[f][j] = wl[(2 * S4 + j) * 64 + 16 * qq + f];
        const float4 r4 = wl[(2 * S4 + S4 - 1) * 64 + 16 * qq + f];
        wvr[f] = float2{r4.x, r4.y};
        bvv[f] = abp[2 * 16 + f];
      }
    }
#pragma unroll
    for (int it = 0; it < MAXT; ++it) {
      const int tt = wave + it * NW;
      if (tt >= TT) break;
      if constexpr (REMV) {
        float pv[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          float p = 0.f;
#pragma unroll
          for (int j = 0; j < C16; ++j) {
            p = fmaf(xa[it][j].x, wv4[f][j].x, p), p = fmaf(xa[it][j].y, wv4[f][j].y, p);
            p = fmaf(xa[it][j].z, wv4[f][j].z, p), p = fmaf(xa[it][j].w, wv4[f][j].w, p);
          }
#pragma unroll
          for (int i = 0; i < REM; ++i) p = fmaf(xr[it][i], i == 0 ? wvr[f].x : wvr[f].y, p);
          p += __shfl_xor(p, 16);
          p += __shfl_xor(p, 32);
          pv[f] = p + bvv[f];
        }
        if (qq == 0) {  // head 1 of the workgroup, v[2 .. 5] of token 16 tt + n
          float* vp = lds + RS + (16 * tt + n) * 8 + 2;
          *reinterpret_cast<float2*>(vp) = float2{pv[0], pv[1]};
          *reinterpret_cast<float2*>(vp + 2) = float2{pv[2], pv[3]};
        }
      }
      f32x4 acc[NCTM];
#pragma unroll
      for (int ct = 0; ct < NCTM; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < C16; ++j) {
#pragma unroll
        for (int ct = 0; ct < NCTM; ++ct) {
          const float4 w4 = wl[(ct * S4 + j) * 64 + lane];
          acc[ct] = mfma16(xa[it][j].x, w4.x, acc[ct]);
          acc[ct] = mfma16(xa[it][j].y, w4.y, acc[ct]);
          acc[ct] = mfma16(xa[it][j].z, w4.z, acc[ct]);
          acc[ct] = mfma16(xa[it][j].w, w4.w, acc[ct]);
        }
      }
      if constexpr (REM > 0) {
#pragma unroll
        for (int ct = 0; ct < NCTM; ++ct) {
          const float4 w4 = wl[(ct * S4 + S4 - 1) * 64 + lane];
#pragma unroll
          for (int i = 0; i < REM; ++i) {
            const float wv = i == 0 ? w4.x : i == 1 ? w4.y : i == 2 ? w4.z : w4.w;
            acc[ct] = mfma16(xr[it][i], wv, acc[ct]);
          }
        }
      }
      const int t0 = 16 * tt + 4 * qq;  // D: lane holds tokens t0 .. t0+3 of feature 16 ct + n
#pragma unroll
      for (int ct = 0; ct < NCTM; ++ct) {
        const float4 o = float4{acc[ct][0] + bias[ct], acc[ct][1] + bias[ct], acc[ct][2] + bias[ct],
                                acc[ct][3] + bias[ct]};
        if (sbase[ct] >= 0) {
          if (!sv[ct]) {
            *reinterpret_cast<float4*>(lds + sbase[ct] + t0) = o;
          } else {
            float* vp = lds + sbase[ct] + t0 * 8;
            vp[0] = o.x, vp[8] = o.y, vp[16] = o.z, vp[24] = o.w;
          }
        }
      }
    }
  }
  if constexpr (STAMP) st_t[3] = FFD_STAMP_T();
  // this wave's head for the rest of the kernel
  const int hh = wave >> 1, gw = wave & 1;
  const int h = hg * HPW + hh;
  float* vs = lds + (size_t)hh * RS;
  float* kts = vs + LS * 8;
  float* qts = kts + 2 * KST * LS;
  // rows served by the shared tables (PURE: all of them; MIXED: tokens >= n_own): the head's two waves fill them
  if (kt != nullptr) {
    if (!q_only && n_own > 0) __syncthreads();  // MIXED: the projection wrote these rows first
    const float* ktab = kt + (size_t)h * L * HD;
    const float* vtab = vt + (size_t)h * L * HD;
    const int tid2 = gw * 64 + lane;  // 0..127 inside the head's wave pair
    for (int j = n_own + tid2; j < L; j += 128) {
      float kx[HD], vx[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vx[e] = 0.f;
      const float* kp = ktab + (size_t)j * HD;
      const float* vp = vtab + (size_t)j * HD;
      if constexpr (HD % 2 == 0) {
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
#pragma unroll
      for (int e = 0; e < HD; ++e) kts[e * LS + j] = kx[e];
      *reinterpret_cast<float4*>(vs + (size_t)j * 8) = float4{vx[0], vx[1], vx[2], vx[3]};
      *reinterpret_cast<float4*>(vs + (size_t)j * 8 + 4) = float4{vx[4], vx[5], vx[6], vx[7]};
    }
    if (q_only) {  // PURE: key rows in [L, Lp) were never written; they are masked but must be finite
      for (int j = L + tid2; j < Lp; j += 128) {
#pragma unroll
        for (int e = 0; e < HD; ++e) kts[e * LS + j] = 0.f;
        *reinterpret_cast<float4*>(vs + (size_t)j * 8) = float4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<float4*>(vs + (size_t)j * 8 + 4) = float4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  __syncthreads();
  unsigned* nrm = reinterpret_cast<unsigned*>(lds + (size_t)HPW * RS + (size_t)NCT * S4 * 256) + 2 * hh;  // this head's pair, see head_norms
  head_norms<HD>(kts, qts, nrm, Lp, LS, gw * 64 + lane, 128);
  if (kt_out != nullptr && b == 0) {  // MIXED: batch element 0 publishes its recomputed rows
    const int tid2 = gw * 64 + lane;
    for (int idx = tid2; idx < n_own * HD; idx += 128) {
      const int j = idx / HD, e = idx - j * HD;
      kt_out[(size_t)h * L * HD + idx] = kts[e * LS + j];
      vt_out[(size_t)h * L * HD + idx] = vs[(size_t)j * 8 + e];
    }
  }

  __syncthreads();  // the tile norms
  if constexpr (STAMP) st_t[4] = FFD_STAMP_T();
  // every score of this head is within +- sqrt(max |q|^2 max |k|^2) (wave-uniform)
  const bool head_bounded =
      __builtin_amdgcn_ballot_w64(__uint_as_float(nrm[1]) * __uint_as_float(nrm[0]) <= T * T) != 0;
  // ------------------------------------------------------------------ phase 2: attention (see k_attention_pk)
  const bool xlane = half == HX;
  constexpr int PF = 4;
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
  for (int qt0 = gw * QG; qt0 < QT; qt0 += 2 * QG) {
    float qf[QG][KSX], mref[QG];
    bool ref_on = false;  // wave-uniform: some lane of this wave carries a non-zero reference
    bool acc_empty[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) acc_empty[g] = true;
    f32x2 lsum[QG], acc[QG][HP];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      const int qtile = (qt0 + g < QT) ? qt0 + g : QT - 1;
#pragma unroll
      for (int s = 0; s < KSX; ++s) {
        const int e = 2 * s + half;
        qf[g][s] = (e < HD) ? qts[e * LS + 32 * qtile + l31] : 0.f;
      }
      mref[g] = 0.f;
      lsum[g] = f32x2{0.f, 0.f};
#pragma unroll
      for (int e = 0; e < HP; ++e) acc[g][e] = f32x2{0.f, 0.f};
    }
#pragma unroll 1
    for (int t = 0; t < KT; ++t) {
      unsigned long long st_a = 0;
      if constexpr (STAMP) st_a = FFD_STAMP_T();
      float kf[KSX];
#pragma unroll
      for (int s = 0; s < KSX; ++s) {
        const int e = 2 * s + half;
        kf[s] = (s < KST && (2 * s + 1 < HD || half == 0)) ? kts[e * LS + 32 * t + l31] : 0.f;
      }
      if (xlane) kf[SX] = 1.0f;
      f32x16 sc[QG];
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KSX; ++s) {
          if (HD % 2 == 0 && s == SX && !ref_on) continue;  // even hd: that step carries nothing but -m_ref = 0
          z = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[g][s], z, 0, 0, 0);
        }
        sc[g] = z;
      }
      const int kbase = 32 * t + 4 * half;
      f32x2 vb[PF][4];
#pragma unroll
      for (int r = 0; r < PF; ++r) load_v(r, kbase, vb[r]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (STAMP) {  // the stamp follows an instruction that reads the last score tile: QK^T has retired
        asm volatile("v_mov_b32 %0, %0" : "+v"(sc[QG - 1][15]));
        const unsigned long long n = FFD_STAMP_T();
        st_qk += n - st_a, st_a = n, ++st_n;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (32 * t + 32 > L) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          if (32 * t + 8 * r4 + 8 > L) {  // (uniform) registers 4 r4 .. 4 r4 + 3 hold key rows 8 r4 .. 8 r4 + 7 of the tile
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const bool dead = kbase + rr + 8 * r4 >= L;
#pragma unroll
              for (int g = 0; g < QG; ++g) sc[g][4 * r4 + rr] = dead ? -INFINITY : sc[g][4 * r4 + rr];
            }
          }
        }
      }
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        // while the head's bound stays within T and no lane of the wave uses a reference, nothing below can trigger
        if (ref_on || !head_bounded) {
        float bm = __builtin_fmaxf(__builtin_fmaxf(sc[g][0], sc[g][1]), sc[g][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) bm = __builtin_fmaxf(__builtin_fmaxf(bm, sc[g][r]), sc[g][r + 1]);
        bm = __builtin_fmaxf(bm, sc[g][15]);
        const float bmx = fmaxf(bm, __shfl_xor(bm, 32));
        // m_ref starts at 0 and usually stays there: |scores| <= 64 (log2 domain) neither overflow nor lose the row
        // to underflow, and the factor 2^-m_ref cancels in the normalisation whatever it is.
        const bool first = acc_empty[g];  // nothing accumulated yet: this is the row's first key tile
        const bool refresh = first ? (fabsf(bmx) > T) : (bmx > T);
        if (__builtin_amdgcn_ballot_w64(refresh) != 0) ref_on = true;
        if (refresh) {
          const float delta = bmx;
          mref[g] += delta;
          if (!first) {
            const float corr = __builtin_amdgcn_exp2f(-delta);
            lsum[g] *= corr;
#pragma unroll
            for (int e = 0; e < HP; ++e) acc[g][e] *= corr;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[g][r] -= delta;
          if (xlane) qf[g][SX] = -mref[g];
        }
        }
        acc_empty[g] = false;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(sc[g][r]);
          const float p1 = __builtin_amdgcn_exp2f(sc[g][r + 1]);
          sc[g][r] = p0;
          sc[g][r + 1] = p1;
          lsum[g] += f32x2{p0, p1};
        }
      }
      if constexpr (STAMP) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long n = FFD_STAMP_T();
        st_sm += n - st_a, st_a = n;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f32x2 vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = vb[r % PF][e];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
          const f32x2 pp = f32x2{sc[g][r & ~1], sc[g][(r & ~1) + 1]};  // (adjacent registers of the accumulator)
#pragma unroll
          for (int e = 0; e < HP; ++e) acc[g][e] = (r & 1) ? pk_fma_hi(pp, vv[e], acc[g][e]) : pk_fma_lo(pp, vv[e], acc[g][e]);
        }
        if (r + PF < 16) {
          load_v(r + PF, kbase, vb[r % PF]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (STAMP) {
        __builtin_amdgcn_sched_barrier(0);
        st_pv += FFD_STAMP_T() - st_a;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (STAMP) st_t[5] = FFD_STAMP_T();
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      float l = lsum[g].x + lsum[g].y;
      l += __shfl_xor(l, 32);
      const float inv = __builtin_amdgcn_rcpf(l);  // (1 ulp; the IEEE division sequence is ten vector instructions per q-tile)
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
        float* orow = out + ((size_t)b * L + q) * D + h * HD;
#pragma unroll
        for (int e = 0; e < HD; ++e) orow[e] = o[e];
      }
    }
  }
  if constexpr (STAMP) {
    if (lane == 0 && stamp != nullptr) {
      unsigned long long* r = stamp + ((size_t)blockIdx.x * NW + wave) * 16;
      st_t[6] = FFD_STAMP_T();
#pragma unroll
      for (int i = 0; i < 7; ++i) r[i] = st_t[i];
      r[7] = st_qk, r[8] = st_sm, r[9] = st_pv, r[10] = st_n, r[11] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
      r[12] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

template <int D, int HD, int HPW, int QG, int NCT, bool QO>
static hipError_t launch_mh_t(const float* x, const float* awp, const float* kt, const float* vt, float* kt_out,
                              float* vt_out, float* out, int B, int L, int n_own, int q_only, hipStream_t s,
                              unsigned long long* stamp) {
  constexpr int KST = (HD + 1) / 2;
  constexpr int S4 = (D + 15) / 16;
  const int KT = (L + 31) / 32;
  const size_t lds = ((size_t)HPW * (64 * QG + 4) * (8 + 4 * KST) + (size_t)NCT * S4 * 256 + (size_t)HPW * 2) * sizeof(float);
  if (cdiv(2 * KT, 2 * HPW) > 3 || cdiv(KT, 2) > QG) return hipErrorInvalidValue;  // <= 3 token tiles, one q-group per wave
  if constexpr (D == 72 && HD == 6) {  // (the stamped twin exists for the headline shape only)
    if (stamp != nullptr) {
      hipLaunchKernelGGL((k_qkv_attention_mh<D, HD, HPW, QG, NCT, QO, true>), dim3(B * (D / HD / HPW)), dim3(128 * HPW), lds,
                         s, x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, q_only, stamp);
      return hipGetLastError();
    }
  }
  if (stamp != nullptr) return hipErrorInvalidValue;
  if (q_only != (QO ? 1 : 0)) return hipErrorInvalidValue;
  auto kern = k_qkv_attention_mh<D, HD, HPW, QG, NCT, QO>;
  hipLaunchKernelGGL(kern, dim3(B * (D / HD / HPW)), dim3(128 * HPW), lds, s, x, awp, kt, vt, kt_out, vt_out, out, B, L,
                     n_own, q_only, (unsigned long long*)nullptr);
  return hipGetLastError();
}

// 2 heads per workgroup, 2 waves per head: L <= 192 (<= 3 q-tiles per wave, 12 token tiles over 4 waves)
template <int D, int HD>
static hipError_t launch_mh2(const float* x, const float* awp, int q_only, const float* kt, const float* vt,
                             float* kt_out, float* vt_out, float* out, int B, int L, int n_own, hipStream_t s,
                             unsigned long long* stamp) {
  constexpr int NF = (2 * 3 * HD + 15) / 16, NQ = (2 * HD + 15) / 16;
  const int QG = cdiv((L + 31) / 32, 2);
#define FFD_MH(qg)                                                                                                   \
  if (QG == qg)                                                                                                      \
    return q_only ? launch_mh_t<D, HD, 2, qg, NQ, true>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 1, s, stamp)     \
                  : launch_mh_t<D, HD, 2, qg, NF, false>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 0, s, stamp);
  FFD_MH(1) FFD_MH(2) FFD_MH(3)
#undef FFD_MH
  return hipErrorInvalidValue;
}

template <int D, int HD, int QG, int NCT>
static hipError_t launch_t(const float* x, const float* awp, const float* kt, const float* vt, float* kt_out,
                           float* vt_out, float* out, int B, int L, int n_own, int q_only, hipStream_t s,
                           unsigned long long* stamp = nullptr) {
  constexpr int KST = (HD + 1) / 2;
  const int KT = (L + 31) / 32;
  const size_t lds = ((size_t)(KT * 32 + 4) * (8 + 4 * KST) + (size_t)2 * KT) * sizeof(float);
  int nwaves = cdiv(KT, QG);
  if (nwaves > 4) nwaves = 4;
  if (cdiv(2 * KT, nwaves) > 8) return hipErrorInvalidValue;  // the projection loop is unrolled for <= 8 token tiles per wave
  if constexpr (D == 72 && HD == 6 && NCT == 2) {  // (the stamped twin exists for the headline model's full pack only)
    if (stamp != nullptr) {
      hipLaunchKernelGGL((k_qkv_attention<D, HD, QG, NCT, false, true>), dim3(B * (D / HD)), dim3(64 * nwaves), lds, s, x,
                         awp, kt, vt, kt_out, vt_out, out, B, L, n_own, q_only, 1, 1, stamp);
      return hipGetLastError();
    }
  }
  if (stamp != nullptr) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_qkv_attention<D, HD, QG, NCT>), dim3(B * (D / HD)), dim3(64 * nwaves), lds, s, x, awp, kt, vt,
                     kt_out, vt_out, out, B, L, n_own, q_only, 1, 1, (unsigned long long*)nullptr);
  return hipGetLastError();
}

// small batches: 4 waves per workgroup, 4 / kspl q-tiles per workgroup, the key range of each cut into kspl pieces
template <int D, int HD, int NCT, bool KVQ = false>
static hipError_t launch_split_t(const float* x, const float* awp, const float* kt, const float* vt, float* kt_out,
                                 float* vt_out, float* out, int B, int L, int n_own, int q_only, int kspl, hipStream_t s) {
  constexpr int KST = (HD + 1) / 2, HP = (HD + 1) / 2;
  const int KT = (L + 31) / 32;
  if (cdiv(2 * KT, 4) > 8 || (kspl != 1 && kspl != 2 && kspl != 4)) return hipErrorInvalidValue;
  const int qsplit = cdiv(KT, 4 / kspl);
  const size_t lds = ((size_t)(KT * 32 + 4) * (8 + 4 * KST) + (size_t)4 * 32 * (2 + 2 * HP) + (size_t)2 * KT) * sizeof(float);
  hipLaunchKernelGGL((k_qkv_attention<D, HD, 1, NCT, true, false, KVQ>), dim3(B * (D / HD) * qsplit), dim3(256), lds, s, x, awp, kt,
                     vt, kt_out, vt_out, out, B, L, n_own, q_only, qsplit, kspl, (unsigned long long*)nullptr);
  return hipGetLastError();
}

thread_local int g_attn_small = 1;  // 0 never, 1 by batch size, 2 / 4 force that many key pieces (ffd_tune "attn_small")

// Key pieces per q-tile of the small-batch split form, 0 when the one-workgroup-per-head(-pair) kernels run: the
// split form projects a head once per q-split, which only pays while the chip is not full.
int qkv_attention_small_split(int B, int H, int L) {
  if (g_attn_small == 0 || L > 512) return 0;
  if (g_attn_small == 2 || g_attn_small == 4) return g_attn_small;
  const int QT = (L + 31) / 32;
  // tools/sweep_small.py (ECG, H = 12, 6 q-tiles): four key pieces win while the grid stays under ~5/8 of the CUs
  // (B <= 2), two pieces up to ~5/4 (B <= 8); beyond that the repeated projection costs more than the idle CUs
  if (8L * B * H * QT <= 5L * num_cus()) return 4;
  if (4L * B * H * cdiv(QT, 2) <= 5L * num_cus()) return 2;
  return 0;
}

template <int D, int HD>
static hipError_t launch_dh(const float* x, const float* awp, int q_only, const float* kt, const float* vt,
                            float* kt_out, float* vt_out, float* out, int B, int L, int n_own, hipStream_t s,
                            unsigned long long* stamp) {
  constexpr int NCTF = (3 * HD + 15) / 16;
  const int QT = (L + 31) / 32;
  if (const int kspl = stamp ? 0 : qkv_attention_small_split(B, D / HD, L)) {
    if constexpr (2 * HD <= 16 && 3 * HD > 16) {  // (q_only == 2: the caller handed over the kv | q pack)
      if (q_only == 2) return launch_split_t<D, HD, 2, true>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 0, kspl, s);
    }
    if (q_only == 2) return hipErrorInvalidValue;
    return q_only ? launch_split_t<D, HD, 1>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 1, kspl, s)
                  : launch_split_t<D, HD, NCTF>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 0, kspl, s);
  }
  if (q_only == 2) return hipErrorInvalidValue;  // the kv | q pack is the split form's only
  if (q_only) {
    if (stamp != nullptr) return hipErrorInvalidValue;
    if (QT == 1) return launch_t<D, HD, 1, 1>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 1, s);
    if (QT % 3 == 0) return launch_t<D, HD, 3, 1>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 1, s);
    if constexpr (HD <= 6) {
      if (QT % 4 == 0 && QT >= 16) return launch_t<D, HD, 4, 1>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 1, s);
    }
    return launch_t<D, HD, 2, 1>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 1, s);
  }
  if (g_attn_qg == 2) return launch_t<D, HD, 2, NCTF>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 0, s, stamp);
  if (g_attn_qg == 1) return launch_t<D, HD, 1, NCTF>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 0, s, stamp);
  if (QT == 1) return launch_t<D, HD, 1, NCTF>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 0, s, stamp);
  if (QT % 3 == 0) return launch_t<D, HD, 3, NCTF>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 0, s, stamp);
  // long sequences: four q-tiles per wave share every K^T / V read (L = 512: 2422 -> 2286 us against two per wave)
  if constexpr (HD <= 6) {  // (hd = 8 would spill at four query groups)
    if (QT % 4 == 0 && QT >= 16) return launch_t<D, HD, 4, NCTF>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 0, s, stamp);
  }
  return launch_t<D, HD, 2, NCTF>(x, awp, kt, vt, kt_out, vt_out, out, B, L, n_own, 0, s, stamp);
}

thread_local int g_attn_fused = 1;  // 1: fused in-projection + attention where a kernel exists (ffd_tune "attn_fused")

// (d_model, head_dim) pairs with a fused kernel; anything else keeps the two-kernel path.
bool qkv_attention_supported(int D, int hd) {
  return (D == 72 && hd == 6) || (D == 60 && hd == 5) || (D == 24 && hd == 6) || (D == 8 && hd == 2) ||
         (D == 64 && hd == 8) || (D == 48 && hd == 4) || (D == 32 && hd == 8) || (D == 16 && hd == 4) ||
         (D == 24 && hd == 3);
}

thread_local int g_attn_hpw = 0;  // 0 heuristic, 1 / 2 force heads per workgroup (ffd_tune "attn_hpw")

// heads per workgroup the fused kernel uses for this shape (the caller passes the matching weight pack)
int qkv_attention_hpw(int D, int hd, int L, int B) {
  if (qkv_attention_small_split(B, D / hd, L)) return 1;
  const bool mh2 = ((D == 72 && hd == 6) || (D == 60 && hd == 5) || (D == 48 && hd == 4)) && L <= 192;
  return (g_attn_hpw == 1 || !mh2) ? 1 : 2;
}

hipError_t launch_qkv_attention(const float* x, const float* awp, int hpw, int q_only, const float* kt,
                                const float* vt, float* kt_out, float* vt_out, float* out, int B, int L, int D, int hd,
                                int n_own, hipStream_t s, unsigned long long* stamp) {
  if (B <= 0) return hipSuccess;
  if (hpw == 2) {
    if (D == 72 && hd == 6) return launch_mh2<72, 6>(x, awp, q_only, kt, vt, kt_out, vt_out, out, B, L, n_own, s, stamp);
    if (D == 60 && hd == 5) return launch_mh2<60, 5>(x, awp, q_only, kt, vt, kt_out, vt_out, out, B, L, n_own, s, stamp);
    if (D == 48 && hd == 4) return launch_mh2<48, 4>(x, awp, q_only, kt, vt, kt_out, vt_out, out, B, L, n_own, s, stamp);
    return hipErrorInvalidValue;
  }
#define FFD_QA(dd, hh) \
  if (D == dd && hd == hh) return launch_dh<dd, hh>(x, awp, q_only, kt, vt, kt_out, vt_out, out, B, L, n_own, s, stamp);
  FFD_QA(72, 6)
  FFD_QA(60, 5)
  FFD_QA(24, 6)
  FFD_QA(8, 2)
  FFD_QA(64, 8)
  FFD_QA(48, 4)
  FFD_QA(32, 8)
  FFD_QA(16, 4)
  FFD_QA(24, 3)
#undef FFD_QA
  return hipErrorInvalidValue;
}

}  // namespace ffd
