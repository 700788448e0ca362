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
// in a fixed order through the LDS copy of the X tile (deterministic), which also
// provides the residual; LayerNorm2 runs on that tile and rows are written back
// fully coalesced.
//
// Large M (MB >= 4) runs PERSISTENT: the grid is the number of workgroups the chip holds
// (two per CU at MB = 4) and each workgroup walks tiles blockIdx.x, += gridDim.x.  The next
// tile's X rows are fetched by LDS-DMA (global_load_lds_dwordx4: no VGPRs -- the kernel sits
// at 252 of its 256 registers) into the second half of a double-buffered X image while the
// current tile's main loop runs, and the weight prefetch of the last chunk wraps around to
// chunk 0 of the next tile.  In-kernel stamps (ffd_probe_ffn_clock) had shown 57 us of a
// workgroup's 157 us per tile outside the main loop (dispatch, X staging, first weight
// fetch, epilogue) while the chip held 2.38 GHz: the matrix pipe, not the clock, was idle.
#include "ffd_internal.h"

namespace ffd {

// REM = true: the d % 16 remainder rows of GEMM2 (8 of 72) run on v_mfma_f32_4x4x1_16b_f32
// (8 cycles per instruction) instead of a zero-padded 16-row tile: block b = lane>>2 multiplies
// A[lane 4b+i] by B[lane 4b+j], and with a GEMM1 accumulator register as B that is
//   W2[c0 + 4g + i][16 fc + 4q + r] * H^T[16 fc + 4q + r][16 mb + 4 mq + j],  q = lane>>4, mq = (lane>>2)&3,
// so each lane-quarter q accumulates the partial sum over its hidden units and the four
// partials are added with two cross-lane xor-adds once per tile.
// OP (round 4, the one-tile forms MB <= 3 only): X is the ATTENTION OUTPUT and the workgroup first forms
// x1 = LN1(Rres + X Wo^T + bo) of its rows (cached_transformer.py:316-322; the prologue of k_oproj_ffn_split at 16 MB rows:
// 90 MB MFMAs over the four waves) instead of reading x1 from a k_linear_res_ln launch.  Y may be Rres (a workgroup reads
// and writes its own rows only).
struct FfnOprojArgs {
  const float *Rres, *Wop, *bo, *g1, *e1;
};
template <int D, int MB, bool REM, bool OP = false>
__global__ __launch_bounds__(256, (MB <= 4 ? 2 : 1)) void k_ffn_ln(const float* __restrict__ X, const float* __restrict__ W1p,
                                                  const float* __restrict__ b1, const float* __restrict__ W2p,
                                                  const float* __restrict__ W2r, const float* __restrict__ b2,
                                                  const float* __restrict__ gam, const float* __restrict__ bet,
                                                  float* __restrict__ Y, int M, int F,
                                                  unsigned long long* __restrict__ stamp, FfnOprojArgs op) {
  // stamp (diagnostic launches of ffd_probe_ffn_clock only, nullptr otherwise): shader-clock and 100 MHz real-time
  // deltas around the main loops, written to memory nothing else reads (MI355X_MICROARCH.md, DVFS item 6)
  constexpr int S = lds_stride(D);
  constexpr int KS = D / 4;
  constexpr int G = dpack_groups(D);
  constexpr int CTP = cdiv(D, 16);                       // tiles in the packed w2 image
  constexpr int NG = (REM && D >= 16) ? w2rem_groups(D) : 0;  // 4-row remainder groups on the 4x4x1 path
  constexpr int CT = (NG > 0) ? D / 16 : cdiv(D, 16);     // 16-row tiles on the 16x16x4 path
  constexpr int NGA = NG > 0 ? NG : 1;
  constexpr int R = 16 * MB;
  constexpr int S2 = ((D + 3) / 4) * 4 + 4;  // partial-sum row stride (16-byte aligned rows)
  constexpr bool DMA = MB >= 4;              // persistent over tiles, X image filled by LDS-DMA
  // X image row stride: the LDS-DMA image is lane-linear (wave-uniform base + lane * 16 B), so its rows are whole
  // float4 slots (S2, one pad slot per row; the fragment reads are then 2-way bank conflicted, once per tile); the
  // register-staged small-M form keeps the conflict-free stride S.
  constexpr int SX = DMA ? S2 : S;
  constexpr int S4 = S2 / 4;
  constexpr int NBUF = DMA ? 2 : 1;
  constexpr int NRED = DMA ? 2 : 3;  // partial-sum buffers (the persistent form trades one for the second X image)
  __shared__ __align__(16) float xsb[NBUF * R * SX];
  __shared__ __align__(16) float red[NRED * R * S2];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntiles = (M + R - 1) / R;

  // a tile's X rows -> image `b`, asynchronously (rows past M repeat the last valid row; they are never stored)
  auto issue_dma = [&](int t, int b) {
    const int rows_valid = min(R, M - t * R);
    const float* Xt = X + (size_t)t * R * D;
    constexpr int NPC = (R * S4 + 63) / 64;
    for (int pc = wave; pc < NPC; pc += 4) {
      const int p = pc * 64 + lane;  // float4 slot of the image
      const int r = p / S4, c4 = p - r * S4;
      const int rr = min(r, rows_valid - 1), cc = min(c4, D / 4 - 1);  // pad slot: any valid address
      if (p < R * S4)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(Xt + (size_t)rr * D + 4 * cc),
                                         (lds_ptr_t)(xsb + b * R * SX + pc * 256), 16, 0, 0);
    }
  };

  int tile = blockIdx.x;
  int buf = 0;
  if (DMA) {
    issue_dma(tile, 0);
  } else if constexpr (OP) {
    static_assert(!OP || (!DMA && D % 4 == 0), "fused out-projection: the one-tile forms");
    constexpr int D4 = D / 4;
    float* pre = red;  // [R][S2]: residual rows -> pre-LN1 rows (the partial-sum buffers are free until the tile end)
    const int rows_valid = min(R, M - tile * R);
    const int n = lane & 15, q = lane >> 4;
    // this wave's out-projection column tiles ct = wave, wave + 4 (requested under the staging)
    const float4* Woq = reinterpret_cast<const float4*>(op.Wop);
    float4 wo[2][G];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ct = wave + 4 * i;
#pragma unroll
      for (int g = 0; g < G; ++g) wo[i][g] = ct < CTP ? Woq[((size_t)ct * G + g) * 64 + lane] : float4{0.f, 0.f, 0.f, 0.f};
    }
    for (int f = threadIdx.x; f < R * D4; f += 256) {
      const int rr = f / D4, c4 = f - rr * D4;
      const size_t off = (size_t)(tile * R + min(rr, rows_valid - 1)) * D + 4 * c4;
      const float4 a = *reinterpret_cast<const float4*>(X + off);
      const float4 x = *reinterpret_cast<const float4*>(op.Rres + off);
      float2* d2 = reinterpret_cast<float2*>(&xsb[rr * SX + 4 * c4]);
      d2[0] = float2{a.x, a.y}, d2[1] = float2{a.z, a.w};
      *reinterpret_cast<float4*>(&pre[rr * S2 + 4 * c4]) = x;
    }
    __syncthreads();
    // out-projection + bias + residual: lane holds columns c .. c+3 of row 16 mb + n
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float af[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) af[s] = xsb[(16 * mb + n) * SX + 4 * s + q];
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float4 w4 = wo[i][s >> 2];
          const float a = (s & 3) == 0 ? w4.x : (s & 3) == 1 ? w4.y : (s & 3) == 2 ? w4.z : w4.w;
          acc[i] = mfma16(a, af[s], acc[i]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = 16 * (wave + 4 * i) + 4 * q;
        if (wave + 4 * i < CTP && c < D) {
          const float4 b4 = *reinterpret_cast<const float4*>(op.bo + c);
          float4* p4 = reinterpret_cast<float4*>(&pre[(16 * mb + n) * S2 + c]);
          const float4 x4 = *p4;
          *p4 = float4{acc[i][0] + b4.x + x4.x, acc[i][1] + b4.y + x4.y, acc[i][2] + b4.z + x4.z, acc[i][3] + b4.w + x4.w};
        }
      }
    }
    __syncthreads();
    // LayerNorm1: 16 threads per row, 16 rows per pass; x1 -> the X image (the FFN's input and LN2's residual)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int row = 16 * mb + (threadIdx.x >> 4), sub = threadIdx.x & 15;
      float4 v[2];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = sub + 16 * i;
        v[i] = c4 < D4 ? *reinterpret_cast<const float4*>(&pre[row * S2 + 4 * c4]) : float4{0.f, 0.f, 0.f, 0.f};
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
      const float mean = sum * (1.0f / D);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (sub + 16 * i < D4) {
          const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
          ss = fmaf(a, a, ss), ss = fmaf(b, b, ss), ss = fmaf(c, c, ss), ss = fmaf(d, d, ss);
        }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 16);
      const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = sub + 16 * i;
        if (c4 < D4) {
          const float4 g4 = *reinterpret_cast<const float4*>(op.g1 + 4 * c4), e4 = *reinterpret_cast<const float4*>(op.e1 + 4 * c4);
          const float4 o = {(v[i].x - mean) * rstd * g4.x + e4.x, (v[i].y - mean) * rstd * g4.y + e4.y,
                            (v[i].z - mean) * rstd * g4.z + e4.z, (v[i].w - mean) * rstd * g4.w + e4.w};
          float2* d2 = reinterpret_cast<float2*>(&xsb[row * SX + 4 * c4]);
          d2[0] = float2{o.x, o.y}, d2[1] = float2{o.z, o.w};
        }
      }
    }
    // (the barrier at the top of the tile loop orders these x1 writes before the fragment reads)
  } else {
    // ---- stage the X tile (coalesced float4) ----
    const float4* X4 = reinterpret_cast<const float4*>(X + (size_t)tile * R * D);
    const int rows_valid = min(R, M - tile * R);
    for (int i4 = threadIdx.x; i4 < R * D / 4; i4 += 256) {
      const int r = (4 * i4) / D, k = 4 * i4 - r * D;
      float4 v = (r < rows_valid) ? X4[i4] : float4{0.f, 0.f, 0.f, 0.f};
      float2* dst = reinterpret_cast<float2*>(&xsb[r * SX + k]);  // S even: 8-byte aligned
      dst[0] = float2{v.x, v.y};
      dst[1] = float2{v.z, v.w};
    }
  }

  // ---- this wave's quarter of F, 16 hidden units per chunk ----
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
  // stamp record of a workgroup (8 x u64): [0] sum of shader-clock deltas over its main loops, [1] the same in 10 ns
  // real-time ticks, [2] real time at entry, [3] / [4] begin / end of its first main loop, [5] end of its first
  // epilogue, [6] exit, [7] tiles processed.  s_memrealtime is one chip-wide 100 MHz counter: a launch timeline.
  unsigned long long st_clk = 0, st_rt = 0, st_acc = 0, st_acc_rt = 0, st_first_b = 0, st_first_e = 0, st_epi = 0;
  const unsigned long long st_entry = stamp ? __builtin_amdgcn_s_memrealtime() : 0ull;
  int st_tiles = 0;

  for (;;) {  // tiles of this workgroup (a single iteration unless DMA)
    float* xs = xsb + buf * R * SX;
    const int m0 = tile * R;
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's image has landed
    __syncthreads();
    // ---- this wave's B fragments; then the next tile's rows start streaming into the other image ----
    float xf[MB][KS];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int s = 0; s < KS; ++s) xf[mb][s] = xs[(16 * mb + (lane & 15)) * SX + 4 * s + (lane >> 4)];
    const int tile_next = tile + (int)gridDim.x;
    if (DMA && tile_next < ntiles) issue_dma(tile_next, buf ^ 1);

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

    if (stamp) {  // scalar: stays in SGPRs
      st_clk = __builtin_amdgcn_s_memtime(), st_rt = __builtin_amdgcn_s_memrealtime();
      if (st_tiles == 0) st_first_b = st_rt;
    }
    for (int ci = 0; ci < nchunk; ++ci) {
      const int nx = (ci + 1 < nchunk) ? ci + 1 : 0;  // the last prefetch wraps to chunk 0 = the next tile's first chunk
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
    if (stamp) {
      const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
      st_acc += __builtin_amdgcn_s_memtime() - st_clk;
      st_acc_rt += rt - st_rt;
      if (st_tiles == 0) st_first_e = rt;
    }

    // ---- deterministic cross-wave reduction through LDS, fixed order of additions:
    //   3 buffers (small M): wave 0 adds its partial into the X tile (= the residual), waves 1..3 park theirs;
    //                        row sum = ((((x + p0) + p1) + p2) + p3) + b2
    //   2 buffers (DMA):     wave 0 -> X tile, waves 1, 2 park; barrier; wave 3 adds into wave 1's buffer;
    //                        row sum = (((x + p0) + (p1 + p3)) + p2) + b2
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
    // The epilogue's index arithmetic is recomputed per tile from an opaque copy of the thread index: hoisted out of
    // the tile loop as loop invariants it lived in VGPRs across the main loop, which sits at the register limit, and
    // was spilled (24 dwords per lane: 12 MB of scratch writes per launch at the ECG shape).
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63;
    // acc = true: add to what the buffer holds (row stride st), else overwrite
    // (a lane's four accumulator registers are four consecutive columns of one row: whole float4 where the row stride
    //  allows -- the persistent form's S2 images -- so the epilogue, which crawls beside the partner workgroup's
    //  MFMA stream, is a quarter of the LDS instructions)
    auto put_partial = [&](float* dst, int st, bool acc) {
      const bool vec = (st & 3) == 0;  // compile-time after inlining (st is SX or S2)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int c0 = 16 * ct + 4 * (lane_e >> 4);
          float* q0 = &dst[(16 * mb + (lane_e & 15)) * st + c0];
          if (vec && c0 + 3 < D) {
            float4* q4 = reinterpret_cast<float4*>(q0);
            float4 o = float4{yacc[ct][mb][0], yacc[ct][mb][1], yacc[ct][mb][2], yacc[ct][mb][3]};
            if (acc) {
              const float4 a = *q4;
              o = float4{a.x + o.x, a.y + o.y, a.z + o.z, a.w + o.w};
            }
            *q4 = o;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (c0 + r < D) q0[r] = acc ? q0[r] + yacc[ct][mb][r] : yacc[ct][mb][r];
          }
        }
      if (NG > 0 && (lane_e >> 4) < NGA) {  // quarter q handles remainder group g = q
        const int g = lane_e >> 4;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = 0.f;
#pragma unroll
            for (int gg = 0; gg < NGA; ++gg) v[i] = (gg == g) ? yrem[gg][mb][i] : v[i];
          }
          float* q0 = &dst[(16 * mb + (lane_e & 15)) * st + 16 * CT + 4 * g];
          if (vec) {
            float4* q4 = reinterpret_cast<float4*>(q0);
            float4 o = float4{v[0], v[1], v[2], v[3]};
            if (acc) {
              const float4 a = *q4;
              o = float4{a.x + o.x, a.y + o.y, a.z + o.z, a.w + o.w};
            }
            *q4 = o;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) q0[i] = acc ? q0[i] + v[i] : v[i];
          }
        }
      }
    };
    if (wave == 0) put_partial(xs, SX, true);
    else if (wave <= NRED) put_partial(red + (size_t)(wave - 1) * R * S2, S2, false);
    __syncthreads();
    if (NRED == 2) {
      if (wave == 3) put_partial(red, S2, true);
      __syncthreads();
    }

    // ---- + b2, LayerNorm2, coalesced store ----
    // threads per row, a power of two (tile heights of 48 / 80 / 96 rows leave the threads past the last row idle)
    constexpr int TPR = R <= 16 ? 16 : R <= 32 ? 8 : R <= 64 ? 4 : 2;
    static_assert(R * TPR <= 256, "a thread group per row");
    const bool row_on = tid_e / TPR < R;
    const int row = row_on ? tid_e / TPR : R - 1, sub = tid_e % TPR;
    const int m = row_on ? m0 + row : M;  // (idle threads: in-range LDS reads, no store)
    if constexpr (DMA && D % 4 == 0) {
      // whole float4 columns c4 = sub, sub + TPR, ... (images with 16-byte aligned rows)
      constexpr int NV = cdiv(D / 4, TPR);
      float4 v4[NV];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c4 = sub + i * TPR;
        v4[i] = float4{0.f, 0.f, 0.f, 0.f};
        if (c4 < D / 4) {
          float4 v = *reinterpret_cast<const float4*>(&xs[row * SX + 4 * c4]);
#pragma unroll
          for (int k = 0; k < NRED; ++k) {
            const float4 p = *reinterpret_cast<const float4*>(&red[(k * R + row) * S2 + 4 * c4]);
            v.x += p.x, v.y += p.y, v.z += p.z, v.w += p.w;
          }
          const float4 bq = *reinterpret_cast<const float4*>(b2 + 4 * c4);
          v.x += bq.x, v.y += bq.y, v.z += bq.z, v.w += bq.w;
          sum += (v.x + v.y) + (v.z + v.w);
          v4[i] = v;
        }
      }
#pragma unroll
      for (int o = TPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float mean = sum * (1.0f / D);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (sub + i * TPR < D / 4) {
          const float a = v4[i].x - mean, b = v4[i].y - mean, c = v4[i].z - mean, d = v4[i].w - mean;
          ss = fmaf(a, a, ss), ss = fmaf(b, b, ss), ss = fmaf(c, c, ss), ss = fmaf(d, d, ss);
        }
#pragma unroll
      for (int o = TPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
      const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
      if (m < M) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int c4 = sub + i * TPR;
          if (c4 < D / 4) {
            const float4 g4 = *reinterpret_cast<const float4*>(gam + 4 * c4), e4 = *reinterpret_cast<const float4*>(bet + 4 * c4);
            *reinterpret_cast<float4*>(Y + (size_t)m * D + 4 * c4) =
                float4{(v4[i].x - mean) * rstd * g4.x + e4.x, (v4[i].y - mean) * rstd * g4.y + e4.y,
                       (v4[i].z - mean) * rstd * g4.z + e4.z, (v4[i].w - mean) * rstd * g4.w + e4.w};
          }
        }
      }
    } else {
    float vals[cdiv(D, TPR)];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < cdiv(D, TPR); ++i) {
      const int c = sub + i * TPR;
      float v = 0.f;
      if (c < D) {
        v = xs[row * SX + c];
#pragma unroll
        for (int k = 0; k < NRED; ++k) v += red[(k * R + row) * S2 + c];
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
    if (stamp) {
      if (st_tiles == 0) st_epi = __builtin_amdgcn_s_memrealtime();
      ++st_tiles;
    }
    if (!DMA || tile_next >= ntiles) break;
    tile = tile_next;
    buf ^= 1;
    // (the barrier at the top of the next iteration orders this tile's LDS reads before the next reduction's
    //  writes: the images alternate, and `red` is next written after that barrier)
  }
  if (stamp && threadIdx.x == 0) {
    unsigned long long* o = stamp + 8 * (size_t)blockIdx.x;
    o[0] = st_acc, o[1] = st_acc_rt, o[2] = st_entry, o[3] = st_first_b, o[4] = st_first_e, o[5] = st_epi;
    o[6] = __builtin_amdgcn_s_memrealtime();
    o[7] = (unsigned long long)st_tiles | ((unsigned long long)__smid() << 32);  // + which CU (xcc, se, cu) it ran on
  }
}

thread_local int g_ffn_persist = 1;      // 1: persistent grid for MB >= 4 (n > 1: n x the resident workgroups); 0: one workgroup per tile
int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
thread_local int g_ffn_height = 1;  // ffd_tune "ffn_height": tile heights of 32 / 48 rows where the 16-row tiles are 1.4 - 3 per CU: 1 with the out-projection + LN1 inside, 2 behind a k_linear_res_ln launch; 0 off
// Between the small-batch pair and the sliced row-owning kernel (ECG: B = 12 ... 21 and 28 ... 65, the reference's default
// sample_batch_size of 50 among them): 16 MB rows per workgroup with MB = the 16-row tiles per CU, so that every CU takes
// one tile and its four waves split F -- 141 ... 256 16-row tiles: one each (B = 16: 0.479 -> 0.430 ms per step, B = 20:
// 0.485 -> 0.432; below 0.55 tiles per CU the F-split pair's finer units win), 320 ... 512: 32-row tiles, 513 ... 768:
// 48-row tiles, with the out-projection + LN1 as the tile's prologue (tools/sweep_tile_heights.py, ms per step before -> behind a
// k_linear_res_ln launch -> in one launch: B = 32 0.688 -> 0.661 -> 0.627, B = 40 0.760 -> 0.671 -> 0.631, B = 50
// 0.958 -> 0.903 -> 0.874, B = 64 1.086 -> 0.920 -> 0.885; from 769 tiles on the 64-row forms were as fast already: a tile
// costs ~ 13 us + 16.2 us per 16 rows, 15.6 of them matrix time).  0 = another form.
int ffn_height_plan(int M, int D, int F) {
  if (!g_ffn_height || g_ffn_mb_override != 0 || g_ffn_split || D % 4 != 0 || D > 72 || F % 64 != 0) return 0;
  const int t16 = cdiv(M, 16), cus = num_cus();
  if (20 * t16 >= 11 * cus && t16 <= cus) return 1;
  if (4 * t16 >= 5 * cus && t16 <= 2 * cus) return 2;
  if (t16 > 2 * cus && t16 <= 3 * cus) return 3;
  return 0;
}
thread_local int g_ffn_rem = 1;          // 1: remainder rows of GEMM2 on the 4x4x1 MFMA (ffd_tune "ffn_rem")
thread_local int g_ffn_mb_override = 0;  // 0 = heuristic; 1 / 2 / 4 forces the tile height (ffd_tune "ffn_mb"; the 128-row MB = 8 instances -- never
                            // selected, 400-656 B of scratch at d_model >= 48 -- were retired in round 4)

template <int D>
static hipError_t launch_ffn_d(const float* X, const LayerWeights& w, float* Y, int M, int F, hipStream_t s,
                               unsigned long long* stamp) {
  // Tile height 16*MB rows.  MB = 4 keeps two workgroups (two waves per SIMD) resident per CU
  // and is the default once the grid fills the chip; smaller tiles for small batches.
  int mb = g_ffn_mb_override;
  if (mb < 1 || mb > 4) {
    const int target = 2 * 256;
    mb = cdiv(M, 64) >= target ? 4 : cdiv(M, 32) >= target ? 2 : 1;
    if (const int hp = ffn_height_plan(M, D, F)) mb = hp;
  }
  dim3 block(256);
  // MB >= 4 is persistent: as many workgroups as the chip holds (two per CU at MB = 4, one at MB = 8)
  const int resident = num_cus() * (mb == 4 ? 2 : 1) * (g_ffn_persist > 0 ? g_ffn_persist : 1);
  auto grid_of = [&](int mbv) {
    const int ntiles = cdiv(M, 16 * mbv);
    return dim3((mbv >= 4 && g_ffn_persist != 0 && ntiles > resident) ? resident : ntiles);
  };
#define FFD_LAUNCH_FFN(MBV)                                                                                       \
  do {                                                                                                            \
    if (g_ffn_rem && MBV >= 3 && D >= 16 && w2rem_groups(D) > 0)                                                                       \
      hipLaunchKernelGGL((k_ffn_ln<D, MBV, true>), grid_of(MBV), block, 0, s, X, w.w1p, w.b1, w.w2p,             \
                         w.w2r, w.b2, w.n2w, w.n2b, Y, M, F, stamp, FfnOprojArgs{});                      \
    else                                                                                                          \
      hipLaunchKernelGGL((k_ffn_ln<D, MBV, false>), grid_of(MBV), block, 0, s, X, w.w1p, w.b1, w.w2p,            \
                         w.w2r, w.b2, w.n2w, w.n2b, Y, M, F, stamp, FfnOprojArgs{});                      \
  } while (0)
  switch (mb) {
    case 4: FFD_LAUNCH_FFN(4); break;
    case 2: FFD_LAUNCH_FFN(2); break;
    case 3: FFD_LAUNCH_FFN(3); break;
    default: FFD_LAUNCH_FFN(1); break;
  }
#undef FFD_LAUNCH_FFN
  return hipGetLastError();
}

// y = LN2(x1 + FFN(x1)), x1 = LN1(Rres + attn Wo^T + bo) in ONE launch where ffn_height_plan applies (one 32- / 48-row
// tile per CU; cached_transformer.py:316-327).  Y may be Rres.
template <int D>
static hipError_t launch_oproj_ffn_d(const float* attn, const float* Rres, const LayerWeights& w, float* Y, int M, int F,
                                     int mb, hipStream_t s) {
  const FfnOprojArgs op{Rres, w.out_wp, w.out_b, w.n1w, w.n1b};
  const dim3 block(256), grid(cdiv(M, 16 * mb));
  if (mb == 3) {
    if (g_ffn_rem && D >= 16 && w2rem_groups(D) > 0)
      hipLaunchKernelGGL((k_ffn_ln<D, 3, true, true>), grid, block, 0, s, attn, w.w1p, w.b1, w.w2p, w.w2r, w.b2, w.n2w, w.n2b,
                         Y, M, F, nullptr, op);
    else
      hipLaunchKernelGGL((k_ffn_ln<D, 3, false, true>), grid, block, 0, s, attn, w.w1p, w.b1, w.w2p, w.w2r, w.b2, w.n2w,
                         w.n2b, Y, M, F, nullptr, op);
  } else if (mb == 2) {
    hipLaunchKernelGGL((k_ffn_ln<D, 2, false, true>), grid, block, 0, s, attn, w.w1p, w.b1, w.w2p, w.w2r, w.b2, w.n2w, w.n2b,
                       Y, M, F, nullptr, op);
  } else if (mb == 1) {
    hipLaunchKernelGGL((k_ffn_ln<D, 1, false, true>), grid, block, 0, s, attn, w.w1p, w.b1, w.w2p, w.w2r, w.b2, w.n2w, w.n2b,
                       Y, M, F, nullptr, op);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_oproj_ffn_ln(const float* attn, const float* Rres, const LayerWeights& w, float* Y, int M, int D, int F,
                               hipStream_t s) {
  const int mb = ffn_height_plan(M, D, F);
  if (M <= 0) return hipSuccess;
  if (!mb || w.out_wp == nullptr) return hipErrorInvalidValue;
  switch (D) {
#define X(d) \
    case d: return launch_oproj_ffn_d<d>(attn, Rres, w, Y, M, F, mb, s);
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

int ffn_tile_rows(int M) {  // rows per workgroup launch_ffn_ln picks (one stamp pair per workgroup)
  int mb = g_ffn_mb_override;
  if (mb < 1 || mb > 4) mb = cdiv(M, 64) >= 512 ? 4 : cdiv(M, 32) >= 512 ? 2 : 1;  // (+ ffn_height_plan: the callers that stamp pass large M)
  return 16 * mb;
}

hipError_t launch_ffn_ln(const float* X, const LayerWeights& w, float* Y, int M, int D, int F, hipStream_t s,
                         unsigned long long* stamp) {
  if (M <= 0) return hipSuccess;
  if (F % 64 != 0) return hipErrorInvalidValue;
  if (w.ring != nullptr && ffn_rows_selected(M, D, F)) return launch_ffn_rows(X, w, Y, M, D, F, s, stamp);
  switch (D) {
#define X(d) \
    case d: return launch_ffn_d<d>(X, w, Y, M, F, s, stamp);
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ffd
