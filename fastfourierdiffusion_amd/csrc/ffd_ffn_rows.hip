// Fused feed-forward block, large M (cached_transformer.py:325-327, nn.TransformerEncoderLayer._ff_block):
//     Y = LayerNorm2( X + W2 relu(W1 X + b1) + b2 )
// Row-owning waves + a CU-shared weight ring (round 3; replaces the F-split workgroup of k_ffn_ln at large M).
//
//   * Every wave owns 16*MB rows and walks the WHOLE hidden dimension: its Y^T accumulators hold complete rows, so
//     b2 + residual + LayerNorm2 happen in registers (a lane has 4 consecutive columns of one row per 16-column tile;
//     a row is spread over the 4 lane quarters: two xor-shuffles per statistic) and rows leave as float4 stores.
//     No partial tiles through LDS, no reduction barriers, nothing that crawls beside another wave's MFMA stream.
//   * The weights are streamed ONCE PER CU: a ring of NSLOT slots in LDS, one slot = the packed fragments of 32 hidden
//     units (W1 rows, W2 columns, b1) in exactly the order the waves consume them, filled by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPRs) three slots ahead and read by all NW waves with
//     ds_read_b128 (one read = the A operands of 4 MFMAs per row block).
//   * One s_barrier per slot, placed BETWEEN the two products of the slot: the fragments the first MFMAs after the
//     barrier need are already in registers, and the slot boundary itself has no barrier (the next slot's first
//     fragments are requested under the current slot's last MFMAs).  The barrier certifies slot i+1 (every wave waits
//     for its own DMA pieces with a counted vmcnt first) and frees slot i-1.
//   * GEMM1 computes H^T chunks for two 16-wide chunks at once (two independent accumulator chains per row block);
//     relu'd accumulators are GEMM2's B operand as in k_ffn_ln; the d % 16 remainder columns run on
//     v_mfma_f32_4x4x1_16b_f32.
//   * Persistent: one workgroup per CU walks tiles of 16*MB*NW rows; the weight stream simply wraps around.  The
//     next tile's X rows arrive by LDS-DMA into the wave's own image while the current tile computes.
// Summation order per output element is fixed (hidden units ascending), independent of grid and tile assignment.
#include "ffd_internal.h"

namespace ffd {

// ---- ring pack: [F/32 slots][SLOT_G groups][64 lanes][4], groups in the order the waves consume them ------------
//   groups 0 .. NQ1-1             W1 stream of the chunk pair, item idx = 4 g + j: k-step s = idx / 2, chunk ch = idx % 2
//                                 = W1[32 p + 16 ch + (lane & 15)][4 s + (lane >> 4)]
//   CT groups per chunk ch = 0, 1 (chunk 0's first), item idx: r = idx / CT, ct = idx % CT
//                                 = W2[16 ct + (lane & 15)][32 p + 16 ch + 4 (lane >> 4) + r]
//   NG groups per chunk ch = 0, 1 (4x4x1 A operands), item idx: r = idx / NG, g = idx % NG
//                                 = W2[16 CT + 4 g + (lane & 3)][32 p + 16 ch + 4 (lane >> 4) + r]
//   last group                    lanes 0..7: b1[32 p + 4 lane + j]
constexpr __host__ __device__ int ring_ct(int D) { return D / 16; }
constexpr __host__ __device__ int ring_ng(int D) { return (D % 16) / 4; }
constexpr __host__ __device__ int ring_nq1(int D) { return cdiv(2 * (D / 4), 4); }
constexpr __host__ __device__ int ring_slot_groups(int D) { return ring_nq1(D) + 2 * (ring_ct(D) + ring_ng(D)) + 1; }
size_t ffn_ring_floats(int D, int F) { return (size_t)(F / 32) * ring_slot_groups(D) * 256; }

__global__ void k_pack_ffn_ring(const float* __restrict__ W1, const float* __restrict__ b1,
                                const float* __restrict__ W2, float* __restrict__ out, int D, int F) {
  const int KS = D / 4, CT = ring_ct(D), NG = ring_ng(D), NQ1 = ring_nq1(D), SG = ring_slot_groups(D);
  const size_t total = (size_t)(F / 32) * SG * 256;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(o & 3), lane = (int)((o >> 2) & 63);
    const int g = (int)((o >> 8) % SG), p = (int)((o >> 8) / SG);
    float v = 0.f;
    if (g < NQ1) {
      const int idx = 4 * g + j;
      if (idx < 2 * KS) {
        const int s = idx >> 1, ch = idx & 1;
        v = W1[(size_t)(32 * p + 16 * ch + (lane & 15)) * D + 4 * s + (lane >> 4)];
      }
    } else if (g < NQ1 + 2 * CT) {
      const int gg = g - NQ1, ch = gg / CT, w = gg % CT;
      const int idx = 4 * w + j, r = idx / CT, ct = idx % CT;
      v = W2[(size_t)(16 * ct + (lane & 15)) * F + 32 * p + 16 * ch + 4 * (lane >> 4) + r];
    } else if (g < SG - 1) {
      const int gg = g - NQ1 - 2 * CT, ch = gg / NG, w = gg % NG;
      const int idx = 4 * w + j, r = idx / NG, gq = idx % NG;
      v = W2[(size_t)(16 * CT + 4 * gq + (lane & 3)) * F + 32 * p + 16 * ch + 4 * (lane >> 4) + r];
    } else if (lane < 8) {
      v = b1[32 * p + 4 * lane + j];
    }
    out[o] = v;
  }
}

hipError_t launch_pack_ffn_ring(const float* W1, const float* b1, const float* W2, float* out, int D, int F,
                                hipStream_t s) {
  if (F % 32 != 0 || D % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pack_ffn_ring, dim3(512), dim3(256), 0, s, W1, b1, W2, out, D, F);
  return hipGetLastError();
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS bytes [lds_byte, lds_byte + 1024).
// Issued from inline asm on purpose: for the builtin form hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of the
// next ds_read of the same __shared__ array (it cannot tell the ring's slots apart), which drains the ring every slot.
// The kernel orders DMA and reads itself: counted vmcnt + s_barrier (see the slot barrier below).
__device__ __forceinline__ void dma_piece(const float* g, unsigned lds_byte) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte), "v"(g) : "memory");  // (m0 is a reserved register: hipcc keeps nothing in it across statements)
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(unsigned long)((const __attribute__((address_space(3))) float*)p);
}

__device__ __forceinline__ float f4e(const float4& q, int j) { return j == 0 ? q.x : j == 1 ? q.y : j == 2 ? q.z : q.w; }

template <int D, int MB, int NW, int NSLOT>
struct FfnRowsCfg {
  static constexpr int KS = D / 4;
  static constexpr int CT = ring_ct(D), NG = ring_ng(D), NQ1 = ring_nq1(D);
  static constexpr int SG = ring_slot_groups(D);
  static constexpr int SLOT_FLOATS = SG * 256;
  static constexpr int SX4 = D / 4 + 1;                      // float4 slots per row of the X image (one pad slot)
  static constexpr int SX = 4 * SX4;
  static constexpr int NPX = cdiv(16 * MB * SX4, 64);        // 1 KiB DMA pieces per wave image
  static constexpr int XIMG_FLOATS = NPX * 256;
  static constexpr int LNP = 3 * D;                          // b2, gamma, beta
  static constexpr int LNP_PAD = cdiv(LNP, 4) * 4;
  static constexpr int LDS_FLOATS = NSLOT * SLOT_FLOATS + NW * XIMG_FLOATS + LNP_PAD;
  static constexpr int NST = MB * (CT + (NG > 0 ? 1 : 0));   // float4 stores per lane in a tile epilogue
  static constexpr int AHEAD = NSLOT - 1;                    // a slot's DMA is issued AHEAD slots before its use
  static constexpr int PD = 3;                               // fragment groups requested ahead of their MFMAs
  static constexpr int NPHI = cdiv(SG, NW), NPLO = SG / NW;  // DMA pieces of a slot per wave (waves < SG % NW: NPHI)
  static_assert(NPHI + NST <= 63, "vmcnt range");
};

template <int D, int MB, int NW, int NSLOT>
__global__ __launch_bounds__(64 * NW, NW / 4) void k_ffn_rows(const float* __restrict__ X, const float* __restrict__ ring,
                                                             const float* __restrict__ b2, const float* __restrict__ gam,
                                                             const float* __restrict__ bet, float* __restrict__ Y, int M,
                                                             int F, int dbg, unsigned long long* __restrict__ stamp) {
  // stamp (diagnostic launches of ffd_probe_ffn_clock only, nullptr otherwise): the record k_ffn_ln writes (8 x u64 per
  // workgroup), to memory nothing else reads
  using C = FfnRowsCfg<D, MB, NW, NSLOT>;
  constexpr int KS = C::KS, CT = C::CT, NG = C::NG, NQ1 = C::NQ1, SG = C::SG;
  constexpr int NGA = NG > 0 ? NG : 1, CTA = CT > 0 ? CT : 1;
  constexpr int R = 16 * MB * NW;
  __shared__ __align__(16) float lds[C::LDS_FLOATS];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, m = lane & 15;
  float* const ringl = lds;
  float* const ximg = lds + NSLOT * C::SLOT_FLOATS + wave * C::XIMG_FLOATS;
  float* const lnp = lds + NSLOT * C::SLOT_FLOATS + NW * C::XIMG_FLOATS;
  const unsigned ring_base = __builtin_amdgcn_readfirstlane(lds_addr(ringl));
  const unsigned ximg_base = __builtin_amdgcn_readfirstlane(lds_addr(ximg));

  const int ntiles = (M + R - 1) / R;
  const int NSL = F / 32;  // slots per tile
  const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int total = my_tiles * NSL;
  if (total == 0) return;  // (uniform over the workgroup)
  int np = 0;  // DMA pieces of a slot this wave issues
  for (int g = wave; g < SG; g += NW) ++np;

  auto issue_ring = [&](int wslot, int rslot) {  // packed slot `wslot` of the layer -> ring slot `rslot`
    const float* src = ring + (size_t)wslot * C::SLOT_FLOATS + lane * 4;
    const unsigned dst = ring_base + (unsigned)rslot * (C::SLOT_FLOATS * 4);
    for (int g = wave; g < SG; g += NW) dma_piece(src + g * 256, dst + g * 1024);
  };
  auto issue_x = [&](int tile) {  // this wave's rows of `tile` -> its image (rows past M repeat row M-1; never stored)
    const int row0 = tile * R + wave * 16 * MB;
#pragma unroll
    for (int pc = 0; pc < C::NPX; ++pc) {
      const int p = pc * 64 + lane;
      const int r = min(p / C::SX4, 16 * MB - 1), c4 = min(p % C::SX4, D / 4 - 1);
      const int rr = min(row0 + r, M - 1);
      dma_piece(X + (size_t)rr * D + 4 * c4, ximg_base + pc * 1024);
    }
  };

  // ---- prologue: LN parameters, first X image, first AHEAD slots ----
  for (int i = threadIdx.x; i < C::LNP; i += 64 * NW) lnp[i] = i < D ? b2[i] : i < 2 * D ? gam[i - D] : bet[i - 2 * D];
  issue_x(blockIdx.x);
#pragma unroll
  for (int j = 0; j < C::AHEAD; ++j)
    if (j < total) issue_ring(j % NSL, j % NSLOT);
  // slot 0 landed (and the X image, which is older); slots 1 .. AHEAD-1 may still be in flight
  if (total >= (int)C::AHEAD && C::AHEAD == 3) {
    if (np == C::NPHI) wait_vm<2 * C::NPHI>(); else wait_vm<2 * C::NPLO>();
  } else {
    wait_vm<0>();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LN parameter writes
  __builtin_amdgcn_s_barrier();

  float xf[MB][KS];
  float4 xres[MB][CTA], xrem[MB];
  f32x4 yacc[CTA][MB], yrem[NGA][MB];
  int tile = blockIdx.x, sl = 0;
  int wnext = C::AHEAD % NSL;  // packed slot the next ring DMA fetches
  // The slot is consumed as a stream of NFR fragment groups (one ds_read_b128 each = the A operands of 4 MFMAs per
  // row block); group k is requested PD groups before its MFMAs, across the slot boundary too (the first PD groups and
  // the two bias fragments of a slot are requested under the previous slot's last MFMAs).
  constexpr int NFR = SG - 1, PD = C::PD;
  float4 hb[2], f[NFR + PD];
  {
    const float* slot = ringl;
    hb[0] = *reinterpret_cast<const float4*>(slot + NFR * 256 + 4 * q);
    hb[1] = *reinterpret_cast<const float4*>(slot + NFR * 256 + 16 + 4 * q);
#pragma unroll
    for (int k = 0; k < PD; ++k) f[k] = *reinterpret_cast<const float4*>(slot + k * 256 + lane * 4);
  }
  unsigned long long st_clk = 0, st_rt = 0, st_acc = 0, st_acc_rt = 0, st_first_b = 0, st_first_e = 0, st_epi = 0;
  const unsigned long long st_entry = stamp ? __builtin_amdgcn_s_memrealtime() : 0ull;
  int st_tiles = 0;
  const bool late_dma = wave >= NW / 2;  // the second wave of each SIMD issues its DMA pieces a few MFMA groups later

  for (int it = 0; it < total; ++it) {
    const float* slot = ringl + (it % NSLOT) * C::SLOT_FLOATS + lane * 4;
    const float* nslot = ringl + ((it + 1) % NSLOT) * C::SLOT_FLOATS + lane * 4;
    if (sl == 0) {  // ---- tile start: B fragments + residual from the wave's image ----
      if (stamp) {
        st_clk = __builtin_amdgcn_s_memtime(), st_rt = __builtin_amdgcn_s_memrealtime();
        if (st_tiles == 0) st_first_b = st_rt;
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[mb][s] = ximg[(16 * mb + m) * C::SX + 4 * s + q];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          xres[mb][ct] = *reinterpret_cast<const float4*>(&ximg[(16 * mb + m) * C::SX + 16 * ct + 4 * q]);
        xrem[mb] = float4{0.f, 0.f, 0.f, 0.f};
        if (NG > 0 && q < NG) xrem[mb] = *reinterpret_cast<const float4*>(&ximg[(16 * mb + m) * C::SX + 16 * CT + 4 * q]);
#pragma unroll
        for (int ct = 0; ct < CTA; ++ct) yacc[ct][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < NGA; ++g) yrem[g][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x4 h[2][MB];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) h[ch][mb] = f32x4{hb[ch].x, hb[ch].y, hb[ch].z, hb[ch].w};

    auto issue_dma = [&]() {
      if (sl == 0 && tile + (int)gridDim.x < ntiles) issue_x(tile + gridDim.x);  // (this tile's fragments are in registers)
      if (it + C::AHEAD < total && !(dbg & 4)) {
        issue_ring(wnext, (it + C::AHEAD) % NSLOT);
        if (++wnext == NSL) wnext = 0;
      }
    };
#pragma unroll
    for (int k = 0; k < NFR; ++k) {
      // request group k + PD (of the next slot once past the end: certified by this iteration's barrier)
      if (k + PD < NFR) f[k + PD] = *reinterpret_cast<const float4*>(slot + (k + PD) * 256);
      else f[k + PD] = *reinterpret_cast<const float4*>(nslot + (k + PD - NFR) * 256);
      if (k == NFR - 1) {
        hb[0] = *reinterpret_cast<const float4*>(nslot - lane * 4 + NFR * 256 + 4 * q);
        hb[1] = *reinterpret_cast<const float4*>(nslot - lane * 4 + NFR * 256 + 16 + 4 * q);
      }
      __builtin_amdgcn_sched_barrier(0);
      const float4 w = f[k];
      if (k < NQ1) {
        // ---- GEMM1: H^T chunks a, b (16 hidden x 16 MB rows each), K = D; the bias is the initial accumulator ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = 4 * k + j;
          if (idx < 2 * KS) {
            const int s = idx >> 1, ch = idx & 1;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) h[ch][mb] = mfma16(f4e(w, j), xf[mb][s], h[ch][mb]);
          }
        }
        if (k == NQ1 - 1) {
          // relu as one v_med3_f32 (x, 0, +inf) per element
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
              for (int r = 0; r < 4; ++r) h[ch][mb][r] = __builtin_amdgcn_fmed3f(h[ch][mb][r], 0.f, __builtin_inff());
          // ---- the slot's barrier: my pieces of slot it+1 have landed; afterwards slot it+1 is readable by everyone
          //      and nobody reads slot it-1 any more ----
          const bool steady = C::AHEAD == 3 && it + 2 < total;  // slot it+2 is in flight and may stay so
          const bool stores = sl == 0 && it > 0;  // the previous tile's stores are younger than every DMA in flight
          if (dbg & 2) {
          } else if (steady) {
            if (np == C::NPHI) {
              if (stores) wait_vm<C::NPHI + C::NST>(); else wait_vm<C::NPHI>();
            } else {
              if (stores) wait_vm<C::NPLO + C::NST>(); else wait_vm<C::NPLO>();
            }
          } else {
            wait_vm<0>();
          }
          if (!(dbg & 2)) __builtin_amdgcn_s_barrier();
          if (!late_dma) issue_dma();
        }
      } else if (k < NQ1 + 2 * CT) {
        // ---- GEMM2: Y^T += W2[:, chunk] relu(H^T chunk); accumulator register r is the k-step ----
        const int ch = (k - NQ1) / CTA, wi = (k - NQ1) % CTA;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = 4 * wi + j, r = idx / CTA, ct = idx % CTA;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) yacc[ct][mb] = mfma16(f4e(w, j), h[ch][mb][r], yacc[ct][mb]);
        }
        if (k == NQ1 + 1 && late_dma) issue_dma();
      } else {
        // remainder columns on the 4x4x1 form, all of the slot's together (switching between the two MFMA forms is
        // expensive: interleaved one by one with the 16x16x4 ones they cost 47 cycles each instead of 8)
        const int ch = (k - NQ1 - 2 * CT) / NGA, wi = (k - NQ1 - 2 * CT) % NGA;
        if (!(dbg & 1)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int idx = 4 * wi + j, r = idx / NGA, g = idx % NGA;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
              yrem[g][mb] = __builtin_amdgcn_mfma_f32_4x4x1f32(f4e(w, j), h[ch][mb][r], yrem[g][mb], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < PD; ++k) f[k] = f[NFR + k];

    if (++sl == NSL) {  // ---- tile end: + b2, + residual, LayerNorm2, float4 stores ----
      sl = 0;
      if (stamp) {
        const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
        st_acc += __builtin_amdgcn_s_memtime() - st_clk;
        st_acc_rt += rt - st_rt;
        if (st_tiles == 0) st_first_e = rt;
      }
      const int row0 = tile * R + wave * 16 * MB;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        float4 v[CTA], vr = float4{0.f, 0.f, 0.f, 0.f};
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float4 bq = *reinterpret_cast<const float4*>(&lnp[16 * ct + 4 * q]);
          const float4 x4 = xres[mb][ct];
          v[ct] = float4{x4.x + (yacc[ct][mb][0] + bq.x), x4.y + (yacc[ct][mb][1] + bq.y), x4.z + (yacc[ct][mb][2] + bq.z),
                         x4.w + (yacc[ct][mb][3] + bq.w)};
          sum += (v[ct].x + v[ct].y) + (v[ct].z + v[ct].w);
        }
        if (NG > 0) {
          // the four lane quarters hold partial sums over their own hidden units: add them (all quarters get the total)
          float t[NGA][4];
#pragma unroll
          for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float a = yrem[g][mb][i];
              a += __shfl_xor(a, 16);
              a += __shfl_xor(a, 32);
              t[g][i] = a;
            }
          if (q < NG) {  // quarter q finishes remainder group g = q
            float a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              a[i] = 0.f;
#pragma unroll
              for (int g = 0; g < NG; ++g) a[i] = (g == q) ? t[g][i] : a[i];
            }
            const float4 bq = *reinterpret_cast<const float4*>(&lnp[16 * CT + 4 * q]);
            const float4 x4 = xrem[mb];
            vr = float4{x4.x + (a[0] + bq.x), x4.y + (a[1] + bq.y), x4.z + (a[2] + bq.z), x4.w + (a[3] + bq.w)};
            sum += (vr.x + vr.y) + (vr.z + vr.w);
          }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / D);
        float ss = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float a = v[ct].x - mean, b = v[ct].y - mean, c = v[ct].z - mean, d = v[ct].w - mean;
          ss = fmaf(a, a, ss), ss = fmaf(b, b, ss), ss = fmaf(c, c, ss), ss = fmaf(d, d, ss);
        }
        if (NG > 0 && q < NG) {
          const float a = vr.x - mean, b = vr.y - mean, c = vr.z - mean, d = vr.w - mean;
          ss = fmaf(a, a, ss), ss = fmaf(b, b, ss), ss = fmaf(c, c, ss), ss = fmaf(d, d, ss);
        }
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
        const int row = row0 + 16 * mb + m;
        if (row < M) {
          float* yr = Y + (size_t)row * D;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const float4 g4 = *reinterpret_cast<const float4*>(&lnp[D + 16 * ct + 4 * q]);
            const float4 e4 = *reinterpret_cast<const float4*>(&lnp[2 * D + 16 * ct + 4 * q]);
            *reinterpret_cast<float4*>(yr + 16 * ct + 4 * q) =
                float4{(v[ct].x - mean) * rstd * g4.x + e4.x, (v[ct].y - mean) * rstd * g4.y + e4.y,
                       (v[ct].z - mean) * rstd * g4.z + e4.z, (v[ct].w - mean) * rstd * g4.w + e4.w};
          }
          if (NG > 0 && q < NG) {
            const float4 g4 = *reinterpret_cast<const float4*>(&lnp[D + 16 * CT + 4 * q]);
            const float4 e4 = *reinterpret_cast<const float4*>(&lnp[2 * D + 16 * CT + 4 * q]);
            *reinterpret_cast<float4*>(yr + 16 * CT + 4 * q) =
                float4{(vr.x - mean) * rstd * g4.x + e4.x, (vr.y - mean) * rstd * g4.y + e4.y,
                       (vr.z - mean) * rstd * g4.z + e4.z, (vr.w - mean) * rstd * g4.w + e4.w};
          }
        }
      }
      if (stamp) {
        if (st_tiles == 0) st_epi = __builtin_amdgcn_s_memrealtime();
        ++st_tiles;
      }
      tile += gridDim.x;
    }
  }
  if (stamp && threadIdx.x == 0) {
    unsigned long long* o = stamp + 8 * (size_t)blockIdx.x;
    o[0] = st_acc, o[1] = st_acc_rt, o[2] = st_entry, o[3] = st_first_b, o[4] = st_first_e, o[5] = st_epi;
    o[6] = __builtin_amdgcn_s_memrealtime();
    o[7] = (unsigned long long)st_tiles | ((unsigned long long)__smid() << 32);
  }
}

int g_ffn_rows = 1;     // 1: row-owning kernel for large M (ffd_tune "ffn_rows"); 0: k_ffn_ln
int g_ffn_rows_dbg = 0;  // timing experiments only (results are wrong when set)
int g_ffn_rows_nw = 0;
int g_ffn_rows_mb = 1;  // 0 = heuristic; 8 / 12 / 16 waves per workgroup

bool ffn_rows_supported(int D, int F) { return D == 72 && F % 32 == 0 && F >= 64; }
// large M: where k_ffn_ln ran its 64-row persistent form
bool ffn_rows_selected(int M, int D, int F) { return g_ffn_rows && ffn_rows_supported(D, F) && cdiv(M, 64) >= 512; }

template <int D, int MB, int NW, int NSLOT>
static hipError_t launch_rows_cfg(const float* X, const LayerWeights& w, float* Y, int M, int F, hipStream_t s,
                                  unsigned long long* stamp) {
  const int R = 16 * MB * NW;
  const int ntiles = cdiv(M, R);
  const int grid = ntiles < num_cus() ? ntiles : num_cus();
  hipLaunchKernelGGL((k_ffn_rows<D, MB, NW, NSLOT>), dim3(grid), dim3(64 * NW), 0, s, X, w.ring, w.b2, w.n2w, w.n2b, Y, M, F, g_ffn_rows_dbg, stamp);
  return hipGetLastError();
}

hipError_t launch_ffn_rows(const float* X, const LayerWeights& w, float* Y, int M, int D, int F, hipStream_t s,
                           unsigned long long* stamp) {
  if (M <= 0) return hipSuccess;
  if (!ffn_rows_supported(D, F) || w.ring == nullptr) return hipErrorInvalidValue;
  const int cfg = g_ffn_rows_nw * 10 + g_ffn_rows_mb;
  switch (cfg) {
    case 82: return launch_rows_cfg<72, 2, 8, 3>(X, w, Y, M, F, s, stamp);
    case 121: return launch_rows_cfg<72, 1, 12, 4>(X, w, Y, M, F, s, stamp);
    case 161: return launch_rows_cfg<72, 1, 16, 3>(X, w, Y, M, F, s, stamp);
    default: return launch_rows_cfg<72, 1, 8, 4>(X, w, Y, M, F, s, stamp);
  }
}

}  // namespace ffd
