// Opt-in (ffd_tune "ffn_split" = 1, off by default): the fused FFN + residual + LN2 of k_ffn_ln
//     y = LN2(x + W2 relu(W1 x + b1) + b2)                                            (cached_transformer.py:325-327)
// with both products on the bf16 matrix cores as an fp32-equivalent SPLIT: every fp32 operand is the sum of three
// bf16 parts, a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 24 significant bits), and
// a product keeps the six terms a1 b1, a1 b2, a2 b1, a1 b3, a3 b1, a2 b2 -- everything down to 2^-24 relative.
// bf16 x bf16 is exact in fp32 and the MFMA accumulates in fp32, so the result differs from the fp32 kernel by a
// few fp32 ulp of the row norm (tools/split_bf16_error.py: 2.0e-7 against 3.2e-7 for the plain fp32 GEMM, both
// relative to fp64).  v_mfma_f32_16x16x32_bf16 runs at 16x the rate of v_mfma_f32_16x16x4_f32: six terms leave
// 16 / 6 = 2.7x, less the K padding of GEMM1 (d = 72 -> 96) and the operand splitting.
//
// It is NOT the arithmetic of the reference (an fp32 FMA chain), so it never runs unless asked for, and bench.py
// reports it on its own line (dtype "bf16x3-split, fp32 accumulate").
//
// Structure (a tile = 64 rows; 4 waves, one workgroup per CU walking tiles blockIdx.x, += gridDim.x -- the kernel uses
// the whole register file):
//   * the X tile is split once into three bf16 planes in LDS (B operand of GEMM1, 8 consecutive k per lane);
//   * wave w owns hidden chunks c = w, w + 4, ... (32 units each) for all four 16-row subtiles, its W1 / W2 fragments
//     (pre-split and pre-permuted by k_pack_w*_split, coalesced 16-byte loads) in registers, each set reloaded
//     under the other product;
//   * GEMM1 gives H^T tiles whose accumulator registers, bias added, relu'd and split, ARE the B operand of GEMM2 (the
//     k order inside a 32-wide step is a permutation that the W2 pack mirrors): the hidden never leaves registers;
//   * the four waves' partial Y tiles meet in LDS ((w0 + w2) + (w1 + w3), fixed order), then residual + LN2.
#include "ffd_internal.h"

namespace ffd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {  // v_cvt_pk_bf16_f32 (round to nearest even)
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// (a, b) -> three packed bf16 pairs with a = a1 + a2 + a3 (+ O(2^-24 |a|))
__device__ __forceinline__ void split3(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = pack_bf16(a, b);
  const float ra = a - bf16_lo(p1), rb = b - bf16_hi(p1);  // exact
  p2 = pack_bf16(ra, rb);
  p3 = pack_bf16(ra - bf16_lo(p2), rb - bf16_hi(p2));
}
__device__ __forceinline__ float plane_of(float a, int p) {  // the p-th bf16 part of a, as a float
  uint32_t p1, p2, p3;
  split3(a, 0.f, p1, p2, p3);
  return bf16_lo(p == 0 ? p1 : p == 1 ? p2 : p3);
}

__device__ __forceinline__ f32x4 mfma_bf16(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// GEMM1's K = d_model in 32-wide bf16 steps.  A remainder of 4, 8 or 12 columns (d = 72: 8) is NOT padded to a 32-wide
// step (three quarters zeros, 6 MFMAs of 16 cycles) but runs as d%32/4 exact-fp32 16x16x4 steps (32 cycles each) into
// the same accumulators: fewer matrix cycles, a third of the fragment bytes, and those terms are exact.
constexpr __host__ __device__ int split_rf(int D) { return (D >= 32 && D % 32 != 0 && D % 32 <= 12) ? (D % 32) / 4 : 0; }
constexpr __host__ __device__ int split_ks(int D) { return split_rf(D) ? D / 32 : cdiv(D, 32); }

// ---- weight packs ------------------------------------------------------------------------------------
// w1s: linear1.weight (F x D) as the A operand of H^T = W1 X^T, K in KS = split_ks(D) steps of 32 (zero past D):
//   [c = F/32][t 2][s KS][p 3][lane 64][j 8] bf16 = part_p(W1[32 c + 16 t + (lane & 15)][32 s + 8 (lane >> 4) + j])
// w1r: its fp32 K remainder (split_rf(D) steps of 4), A operand of v_mfma_f32_16x16x4_f32:
//   [c][lane 64][t 2][step 4] float = W1[32 c + 16 t + (lane & 15)][32 KS + 4 step + (lane >> 4)]   (0 for step >= RF)
// w2s: linear2.weight (D x F) as the A operand of Y^T += W2 H^T with the GEMM1 accumulators as B:
//   [c][ct = ceil(D/16)][p 3][lane][j] = part_p(W2[16 ct + (lane & 15)][32 c + hid(lane >> 4, j)]),
//   hid(q, j) = 4 q + j (j < 4: tile t = 0, register j) | 16 + 4 q + j - 4 (tile t = 1)
__global__ void k_pack_w1_split(const float* __restrict__ W1, uint16_t* __restrict__ out, int D, int F) {
  const int KS = split_ks(D);
  const size_t total = (size_t)(F / 32) * 2 * KS * 3 * 64 * 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63;
    size_t r = i >> 9;
    const int p = r % 3;
    r /= 3;
    const int s = r % KS;
    r /= KS;
    const int t = r & 1, c = (int)(r >> 1);
    const int f = 32 * c + 16 * t + (lane & 15), k = 32 * s + 8 * (lane >> 4) + j;
    const float v = k < D ? plane_of(W1[(size_t)f * D + k], p) : 0.f;
    out[i] = (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);
  }
}
__global__ void k_pack_w1_rem(const float* __restrict__ W1, float* __restrict__ out, int D, int F) {
  const int KS = split_ks(D), RF = split_rf(D);
  const size_t total = (size_t)(F / 32) * 64 * 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int step = i & 3, t = (i >> 2) & 1, lane = (i >> 3) & 63, c = (int)(i >> 9);
    const int f = 32 * c + 16 * t + (lane & 15), k = 32 * KS + 4 * step + (lane >> 4);
    out[i] = (step < RF && k < D) ? W1[(size_t)f * D + k] : 0.f;
  }
}
__global__ void k_pack_w2_split(const float* __restrict__ W2, uint16_t* __restrict__ out, int D, int F) {
  const int CT = cdiv(D, 16);
  const size_t total = (size_t)(F / 32) * CT * 3 * 64 * 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63;
    size_t r = i >> 9;
    const int p = r % 3;
    r /= 3;
    const int ct = r % CT, c = (int)(r / CT);
    const int q = lane >> 4;
    const int o = 16 * ct + (lane & 15), hid = 32 * c + (j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4));
    const float v = o < D ? plane_of(W2[(size_t)o * F + hid], p) : 0.f;
    out[i] = (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);
  }
}

size_t w1split_bytes(int D, int F) {  // the bf16 fragments, then (16-byte aligned) the fp32 remainder fragments
  return (size_t)(F / 32) * 2 * split_ks(D) * 3 * 64 * 16 + (size_t)(F / 32) * 64 * 32;
}
static size_t w1rem_offset_bytes(int D, int F) { return (size_t)(F / 32) * 2 * split_ks(D) * 3 * 64 * 16; }
size_t w2split_bytes(int D, int F) { return (size_t)(F / 32) * cdiv(D, 16) * 3 * 64 * 16; }

hipError_t launch_pack_ffn_split(const float* W1, const float* W2, void* w1s, void* w2s, int D, int F, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_w1_split, dim3(256), dim3(256), 0, s, W1, (uint16_t*)w1s, D, F);
  hipLaunchKernelGGL(k_pack_w1_rem, dim3(64), dim3(256), 0, s, W1, (float*)((char*)w1s + w1rem_offset_bytes(D, F)), D, F);
  hipLaunchKernelGGL(k_pack_w2_split, dim3(256), dim3(256), 0, s, W2, (uint16_t*)w2s, D, F);
  return hipGetLastError();
}

// ---- the kernel ----------------------------------------------------------------------------------------
template <int D>
struct SplitGeom {
  static constexpr int KS = split_ks(D);         // 32-wide bf16 k-steps of GEMM1
  static constexpr int RF = split_rf(D);         // + exact-fp32 remainder steps of 4
  static constexpr int KP = 32 * KS;
  static constexpr int XRS = 4 * RF + 1;         // floats per row of the fp32 remainder image (odd: conflict-free)
  static constexpr int CT = cdiv(D, 16);         // 16-wide column tiles of Y
  static constexpr int XS = 2 * KP + 16;         // bytes per row of an X plane (odd multiple of 16: conflict-free b128 reads)
  static constexpr int PLANE = 64 * XS;
  static constexpr int RS = 16 * CT + 4;         // floats per row of a reduction image
  static constexpr size_t lds = (size_t)3 * PLANE + (size_t)2 * 64 * RS * sizeof(float) + (size_t)64 * XRS * sizeof(float);
};

template <int D>
__global__ __launch_bounds__(256, 1) void k_ffn_ln_split(const float* __restrict__ X, const uint4* __restrict__ W1s,
                                                         const float* __restrict__ b1, const uint4* __restrict__ W2s,
                                                         const float* __restrict__ b2, const float* __restrict__ g2,
                                                         const float* __restrict__ e2, float* __restrict__ Y, int M,
                                                         int F, unsigned long long* __restrict__ stamp) {
  // stamp (ffd_probe_ffn_clock only, nullptr otherwise): the 8 x u64 record of k_ffn_ln (ffd_ffn.hip)
  using G = SplitGeom<D>;
  constexpr int KS = G::KS, KP = G::KP, CT = G::CT, XS = G::XS, PLANE = G::PLANE, RS = G::RS, RF = G::RF, XRS = G::XRS;
  constexpr int DB = RF ? KP : D;  // columns that live in the bf16 planes
  constexpr int D4 = D / 4;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* ximg = smem;                                        // [p 3][row 64][XS bytes]
  float* red = reinterpret_cast<float*>(smem + 3 * PLANE);           // [2][row 64][RS]
  float* xrem = red + 2 * 64 * RS;                                   // [row 64][XRS]: fp32 columns DB .. D - 1
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int ntiles = (M + 63) >> 6;
  const int ncw = F / 128;  // chunks per wave
  const float4* W1r = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(W1s) + (size_t)(F / 32) * 2 * KS * 3 * 64 * 16);
  const unsigned long long st_entry = stamp ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // ---- this wave's fragments of a chunk: W1 [t][s][p] + the bias quads of the two hidden tiles; W2 [ct][p] ----
  // One copy of each in registers.  W1 of chunk i + 1 is requested when GEMM1 of chunk i has issued (it arrives under
  // GEMM2 of chunk i), W2 of chunk i + 1 when GEMM2 of chunk i has issued (it arrives under GEMM1 of chunk i + 1).
  // (A second copy of W1 requested a whole chunk ahead was slower, 45 against 41 us per tile: the waits are not
  // latency -- every CU streams 57 GB/s of weights out of L2 here -- and 464 registers brought AGPR shuffling.)
  uint4 w1[2][KS][3], w2[CT][3];
  float4 w1r[2];  // fp32 remainder fragments: [t] . {step 0..3}
  float4 ba, bb;
  float4 ban, bbn;  // the next chunk's bias quads (the current ones are still needed when W1 is reloaded)
  auto load_w1_next = [&](int i) {
    const int c = 4 * i + wave;
    const uint4* p1 = W1s + (size_t)c * (2 * KS * 3 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) w1[t][s][p] = p1[((t * KS + s) * 3 + p) * 64];
    if constexpr (RF > 0) {
      const float4* pr = W1r + ((size_t)c * 64 + lane) * 2;
      w1r[0] = pr[0], w1r[1] = pr[1];
    }
    ban = *reinterpret_cast<const float4*>(b1 + 32 * c + 4 * q);
    bbn = *reinterpret_cast<const float4*>(b1 + 32 * c + 16 + 4 * q);
  };
  auto load_w2 = [&](int i) {
    const int c = 4 * i + wave;
    const uint4* p2 = W2s + (size_t)c * (CT * 3 * 64) + lane;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int p = 0; p < 3; ++p) w2[ct][p] = p2[(ct * 3 + p) * 64];
  };
  load_w1_next(0);
  ba = ban, bb = bbn;
  load_w2(0);

  // ---- X tile -> three bf16 planes in LDS (rows past M repeat the last row; k in [D, KP) is zero) ----
  // Persistent over tiles (one workgroup per CU): the next tile's rows are requested into registers when the main loop
  // ends and converted into the (single) image after the epilogue, so a tile boundary costs the epilogue + the
  // conversion instead of a launch-time prologue with its exposed load latency (3.1 us per tile before).
  constexpr int NXP = cdiv(64 * D4, 256);
  float4 xpre[NXP];
  auto stage_load = [&](int t) {
    const int rv = min(64, M - t * 64);
#pragma unroll
    for (int u = 0; u < NXP; ++u) {
      const int f = threadIdx.x + 256 * u;
      const int row = f / D4, c4 = f - row * D4;
      if (f < 64 * D4) xpre[u] = *reinterpret_cast<const float4*>(X + (size_t)(t * 64 + min(row, rv - 1)) * D + 4 * c4);
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int u = 0; u < NXP; ++u) {
      const int f = threadIdx.x + 256 * u;
      const int row = f / D4, c4 = f - row * D4;
      if (f >= 64 * D4) continue;
      const float4 x = xpre[u];
      if (RF > 0 && 4 * c4 >= DB) {  // the fp32 remainder columns
        float* d = xrem + row * XRS + (4 * c4 - DB);
        d[0] = x.x, d[1] = x.y, d[2] = x.z, d[3] = x.w;
        continue;
      }
      uint32_t a1, a2, a3, c1, c2, c3;
      split3(x.x, x.y, a1, a2, a3);
      split3(x.z, x.w, c1, c2, c3);
      unsigned char* dst = ximg + row * XS + c4 * 8;
      *reinterpret_cast<uint2*>(dst) = uint2{a1, c1};
      *reinterpret_cast<uint2*>(dst + PLANE) = uint2{a2, c2};
      *reinterpret_cast<uint2*>(dst + 2 * PLANE) = uint2{a3, c3};
    }
  };
  int tile = blockIdx.x;
  stage_load(tile);
  if constexpr (RF == 0 && KP > D) {
    constexpr int PZ = (KP - D) / 4;  // 8-byte units of padding per row
    for (int f = threadIdx.x; f < 3 * 64 * PZ; f += 256) {
      const int p = f / (64 * PZ), rem = f - p * 64 * PZ, row = rem / PZ, u = rem - row * PZ;
      *reinterpret_cast<uint2*>(ximg + p * PLANE + row * XS + 2 * D + 8 * u) = uint2{0u, 0u};
    }
  }

  // the six kept terms, smallest first: (weight part, activation part)
  constexpr int TW[6] = {2, 0, 1, 1, 0, 0};
  constexpr int TX[6] = {0, 2, 1, 0, 1, 0};
  // One chunk = for each of the four row subtiles: A(st) = GEMM1 (12 KS MFMAs), B(st) = bias + ReLU + three-way split
  // of its 8 hidden values per lane (~70 vector instructions), C(st) = GEMM2 (6 CT MFMAs).  One wave per SIMD, so the
  // vector work has to ride in the issue slots the wave's own MFMAs leave (8 of every 16 cycles): the instruction
  // stream is  A0 | A1+B0 | A2+B1 | A3+B2 | C0 C1 + B3 | C2 C3.  Each "+" is written out: B is cut into 20 stages of
  // 2-4 instructions (split_stage) that are placed between the MFMAs with a scheduling fence after each, because the
  // scheduler on its own (and with sched_group_barrier hints) leaves the vector work in runs of 25-55 instructions
  // during which the matrix pipe idles.  The X fragments of k-step s are re-read from LDS for the next subtile as soon
  // as the current subtile's MFMAs of that step have issued.
  int xoff = n * XS + 16 * q;  // this lane's byte offset inside a 16-row block of an X plane
  uint4 xf[KS][3];
  auto read_x = [&](int st, int s) {
#pragma unroll
    for (int p = 0; p < 3; ++p) xf[s][p] = *reinterpret_cast<const uint4*>(ximg + xoff + p * PLANE + 16 * st * XS + 64 * s);
  };
  float xr[RF > 0 ? RF : 1];  // fp32 remainder B fragments of the current subtile: X[row n][DB + 4 step + q]
  int xro = 0;                // (opaque zero, like xoff: keeps these reads inside the loop)
  auto read_xr = [&](int st) {
#pragma unroll
    for (int u = 0; u < RF; ++u) xr[u] = xrem[(16 * st + n) * XRS + 4 * u + q + xro];
  };
  unsigned long long st_b = 0, st_b_rt = 0, st_acc = 0, st_acc_rt = 0, st_first_b = 0, st_first_e = 0, st_epi = 0;
  int st_tiles = 0;
  constexpr int NSTG = 20;  // 4 pairs x 5 stages
  for (;;) {  // tiles of this workgroup
  const int m0 = tile * 64;
  const int rows_valid = min(64, M - m0);
  stage_store();  // (every wave left the previous tile's main loop -- the only reader of the image -- two barriers ago)
  __syncthreads();
#pragma unroll
  for (int s = 0; s < KS; ++s) read_x(0, s);
  read_xr(0);
  f32x4 yacc[4][CT];
#pragma unroll
  for (int st = 0; st < 4; ++st)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) yacc[st][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (stamp) {
    st_b = __builtin_amdgcn_s_memtime(), st_b_rt = __builtin_amdgcn_s_memrealtime();
    if (st_tiles == 0) st_first_b = st_b_rt;
  }
  for (int i = 0; i < ncw; ++i) {
    // (opaque per iteration: otherwise all 12 KS X fragments are hoisted out of the loop into AGPRs and every MFMA
    // pays three v_accvgpr_read to get them back)
    asm volatile("" : "+v"(xoff));
    asm volatile("" : "+v"(xro));
    f32x4 h[4][2];
    uint4 hf[4][3];
    const float bav[4] = {ba.x, ba.y, ba.z, ba.w}, bbv[4] = {bb.x, bb.y, bb.z, bb.w};
    // stage u of B(st): pair j = u / 5 is (register 2 (j & 1), + 1) of hidden tile j >> 1; the accumulator registers
    // become component j of GEMM2's B fragments hf[st][part]
    float e0, e1;
    auto split_stage = [&](int st, int u) {
      const int j = u / 5, ph = u % 5, t = j >> 1, r = 2 * (j & 1);
      uint32_t& p1 = j == 0 ? hf[st][0].x : j == 1 ? hf[st][0].y : j == 2 ? hf[st][0].z : hf[st][0].w;
      uint32_t& p2 = j == 0 ? hf[st][1].x : j == 1 ? hf[st][1].y : j == 2 ? hf[st][1].z : hf[st][1].w;
      uint32_t& p3 = j == 0 ? hf[st][2].x : j == 1 ? hf[st][2].y : j == 2 ? hf[st][2].z : hf[st][2].w;
      if (ph == 0) {
        e0 = __builtin_amdgcn_fmed3f(h[st][t][r] + (t ? bbv[r] : bav[r]), 0.f, __builtin_inff());
        e1 = __builtin_amdgcn_fmed3f(h[st][t][r + 1] + (t ? bbv[r + 1] : bav[r + 1]), 0.f, __builtin_inff());
      } else if (ph == 1) {
        p1 = pack_bf16(e0, e1);
      } else if (ph == 2) {
        e0 -= bf16_lo(p1), e1 -= bf16_hi(p1);  // exact
      } else if (ph == 3) {
        p2 = pack_bf16(e0, e1);
      } else {
        p3 = pack_bf16(e0 - bf16_lo(p2), e1 - bf16_hi(p2));
      }
    };
#pragma unroll
    for (int st = 0; st < 4; ++st) {  // A(st) + B(st - 1)
      constexpr int NM = 12 * KS + 2 * RF;
      h[st][0] = h[st][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (RF > 0) {  // the exact fp32 remainder steps first
#pragma unroll
        for (int u = 0; u < RF; ++u)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float a = u == 0 ? w1r[t].x : u == 1 ? w1r[t].y : u == 2 ? w1r[t].z : w1r[t].w;
            h[st][t] = mfma16(a, xr[u], h[st][t]);
            if (st > 0) {
              const int mi = u * 2 + t;
#pragma unroll
              for (int v = mi * NSTG / NM; v < (mi + 1) * NSTG / NM; ++v) split_stage(st - 1, v);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        read_xr((st + 1) & 3);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            h[st][t] = mfma_bf16(w1[t][s][TW[k]], xf[s][TX[k]], h[st][t]);
            if (st > 0) {
              const int mi = 2 * RF + (s * 6 + k) * 2 + t;
#pragma unroll
              for (int u = mi * NSTG / NM; u < (mi + 1) * NSTG / NM; ++u) split_stage(st - 1, u);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        read_x((st + 1) & 3, s);  // next subtile (after the last one: the next chunk's first)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // (unconditional: a branch here would split the loop body into basic blocks)
    load_w1_next(i + 1 < ncw ? i + 1 : 0);  // the last chunk requests chunk 0: the next tile's first
    __builtin_amdgcn_sched_barrier(0);
    {
      constexpr int NM = 12 * CT;  // C0 C1 + B3
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            yacc[st][ct] = mfma_bf16(w2[ct][TW[k]], hf[st][TX[k]], yacc[st][ct]);
            const int mi = (st * 6 + k) * CT + ct;
#pragma unroll
            for (int u = mi * NSTG / NM; u < (mi + 1) * NSTG / NM; ++u) split_stage(3, u);
            __builtin_amdgcn_sched_barrier(0);
          }
    }
#pragma unroll
    for (int st = 2; st < 4; ++st)
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) yacc[st][ct] = mfma_bf16(w2[ct][TW[k]], hf[st][TX[k]], yacc[st][ct]);
    __builtin_amdgcn_sched_barrier(0);
    load_w2(i + 1 < ncw ? i + 1 : 0);
    ba = ban, bb = bbn;
  }

  if (stamp) {
    const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
    st_acc += __builtin_amdgcn_s_memtime() - st_b, st_acc_rt += rt - st_b_rt;
    if (st_tiles == 0) st_first_e = rt;
  }
  const int tile_next = tile + (int)gridDim.x;
  if (tile_next < ntiles) stage_load(tile_next);  // lands under the epilogue
  // ---- (w0 + w2) + (w1 + w3) through two LDS images, then residual + b2 + LN2 with 4 threads per row ----
  auto img_at = [&](int img, int st, int ct) { return red + ((size_t)img * 64 + 16 * st + n) * RS + 16 * ct + 4 * q; };
  if (wave < 2) {
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        *reinterpret_cast<float4*>(img_at(wave, st, ct)) = float4{yacc[st][ct][0], yacc[st][ct][1], yacc[st][ct][2], yacc[st][ct][3]};
  }
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float4* p = reinterpret_cast<float4*>(img_at(wave - 2, st, ct));
        const float4 a = *p;
        *p = float4{a.x + yacc[st][ct][0], a.y + yacc[st][ct][1], a.z + yacc[st][ct][2], a.w + yacc[st][ct][3]};
      }
  }
  __syncthreads();
  {
    constexpr int NV = cdiv(D4, 4);  // float4 columns per thread
    const int row = threadIdx.x >> 2, t4 = threadIdx.x & 3;
    const int m = m0 + min(row, rows_valid - 1);
    float4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = t4 + 4 * i;
      v[i] = float4{0.f, 0.f, 0.f, 0.f};
      if (c4 < D4) {
        const float4 a = *reinterpret_cast<const float4*>(red + (size_t)row * RS + 4 * c4);
        const float4 b = *reinterpret_cast<const float4*>(red + (size_t)(64 + row) * RS + 4 * c4);
        const float4 x = *reinterpret_cast<const float4*>(X + (size_t)m * D + 4 * c4);
        const float4 bo = *reinterpret_cast<const float4*>(b2 + 4 * c4);
        v[i] = float4{(x.x + (a.x + b.x)) + bo.x, (x.y + (a.y + b.y)) + bo.y, (x.z + (a.z + b.z)) + bo.z,
                      (x.w + (a.w + b.w)) + bo.w};
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    const float mean = sum * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (t4 + 4 * i < D4) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        ss = fmaf(a, a, ss), ss = fmaf(b, b, ss), ss = fmaf(c, c, ss), ss = fmaf(d, d, ss);
      }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
    if (row < rows_valid) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c4 = t4 + 4 * i;
        if (c4 < D4) {
          const float4 g4 = *reinterpret_cast<const float4*>(g2 + 4 * c4), e4 = *reinterpret_cast<const float4*>(e2 + 4 * c4);
          *reinterpret_cast<float4*>(Y + (size_t)(m0 + row) * D + 4 * c4) =
              float4{(v[i].x - mean) * rstd * g4.x + e4.x, (v[i].y - mean) * rstd * g4.y + e4.y,
                     (v[i].z - mean) * rstd * g4.z + e4.z, (v[i].w - mean) * rstd * g4.w + e4.w};
        }
      }
    }
  }
  if (stamp) {
    if (st_tiles == 0) st_epi = __builtin_amdgcn_s_memrealtime();
    ++st_tiles;
  }
  if (tile_next >= ntiles) break;
  tile = tile_next;
  }  // tiles
  if (stamp && threadIdx.x == 0) {
    unsigned long long* o = stamp + 8 * (size_t)blockIdx.x;
    o[0] = st_acc, o[1] = st_acc_rt, o[2] = st_entry, o[3] = st_first_b, o[4] = st_first_e, o[5] = st_epi;
    o[6] = __builtin_amdgcn_s_memrealtime();
    o[7] = (unsigned long long)st_tiles | ((unsigned long long)__smid() << 32);
  }
}

thread_local int g_ffn_split = 0;  // ffd_tune "ffn_split": 1 = the bf16x3-split FFN (opt-in, not the reference's fp32 arithmetic)

bool ffn_split_supported(int D, int F) { return D % 4 == 0 && D <= 96 && F % 128 == 0; }

template <int D>
static hipError_t launch_split_d(const float* X, const LayerWeights& w, float* Y, int M, int F, hipStream_t s,
                                 unsigned long long* stamp) {
  auto kern = k_ffn_ln_split<D>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SplitGeom<D>::lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int ntiles = cdiv(M, 64);
  hipLaunchKernelGGL(kern, dim3(ntiles < num_cus() ? ntiles : num_cus()), dim3(256), SplitGeom<D>::lds, s, X, (const uint4*)w.w1s, w.b1,
                     (const uint4*)w.w2s, w.b2, w.n2w, w.n2b, Y, M, F, stamp);
  return hipGetLastError();
}

hipError_t launch_ffn_ln_split(const float* X, const LayerWeights& w, float* Y, int M, int D, int F, hipStream_t s,
                               unsigned long long* stamp) {
  if (M <= 0) return hipSuccess;
  if (!ffn_split_supported(D, F) || w.w1s == nullptr || w.w2s == nullptr) return hipErrorInvalidValue;
  switch (D) {
#define X(d) \
  case d: return launch_split_d<d>(X, w, Y, M, F, s, stamp);
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ffd
