// d-contraction GEMMs on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32):
//   Y^T[n][m] = sum_k W[n][k] X[m][k]     (A = packed W fragment, B = X fragment)
// K is always d_model (<= 80) on this path, so the X fragments of a 64-row tile live
// in registers for the whole kernel and only packed weights stream (L2-resident,
// one coalesced float4 per lane = 4 k-steps).
//   k_linear         : Y = X W^T + b                       (QKV / KV / LSTM gates)
//   k_linear_res_ln  : Y = LN(R + X W^T + b)               (out-proj + residual + LN1)
#include "ffd_internal.h"

namespace ffd {

// Stage a 64-row tile (contiguous 64*D floats of a row-major (M x D) matrix) into LDS with
// coalesced float4 loads; rows >= M are zero.
template <int D>
__device__ __forceinline__ void load_x_tile(const float* __restrict__ X, float* xs, int m0, int M) {
  constexpr int S = lds_stride(D);
  const float4* X4 = reinterpret_cast<const float4*>(X + (size_t)m0 * D);
  const int rows_valid = min(64, M - m0);
  for (int i4 = threadIdx.x; i4 < 64 * D / 4; i4 += blockDim.x) {
    const int r = (4 * i4) / D, k = 4 * i4 - r * D;
    const float4 v = (r < rows_valid) ? X4[i4] : float4{0.f, 0.f, 0.f, 0.f};
    float2* dst = reinterpret_cast<float2*>(&xs[r * S + k]);  // S even: 8-byte aligned
    dst[0] = float2{v.x, v.y};
    dst[1] = float2{v.z, v.w};
  }
}

template <int D>
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ X, const float* __restrict__ Wp,
                                                const float* __restrict__ bias, float* __restrict__ Y, int M, int N,
                                                int ldy) {
  constexpr int S = lds_stride(D);
  constexpr int KS = D / 4;
  constexpr int G = dpack_groups(D);
  __shared__ float xs[64 * S];
  const int m0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  load_x_tile<D>(X, xs, m0, M);
  __syncthreads();
  float xf[4][KS];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[mb][s] = xs[(16 * mb + (lane & 15)) * S + 4 * s + (lane >> 4)];

  const int NT = cdiv(N, 16);
  const float4* Wq = reinterpret_cast<const float4*>(Wp);
  const bool vec_ok = ((ldy & 3) == 0) && ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
  for (int nt = wave; nt < NT; nt += 4) {
    float4 wq[G];
#pragma unroll
    for (int g = 0; g < G; ++g) wq[g] = Wq[((size_t)nt * G + g) * 64 + lane];
    f32x4 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 q = wq[s >> 2];
      const float a = (s & 3) == 0 ? q.x : (s & 3) == 1 ? q.y : (s & 3) == 2 ? q.z : q.w;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) acc[mb] = mfma16(a, xf[mb][s], acc[mb]);
    }
    const int n = 16 * nt + 4 * (lane >> 4);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int m = m0 + 16 * mb + (lane & 15);
      if (m >= M) continue;
      float* yr = Y + (size_t)m * ldy + n;
      if (vec_ok && n + 3 < N) {
        float4 b4 = *reinterpret_cast<const float4*>(bias + n);
        float4 o = {acc[mb][0] + b4.x, acc[mb][1] + b4.y, acc[mb][2] + b4.z, acc[mb][3] + b4.w};
        *reinterpret_cast<float4*>(yr) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) yr[r] = acc[mb][r] + bias[n + r];
      }
    }
  }
}

// Head-major projection (Q / K / V): column n = reg*d + h*hd + e of row m = b*L + l goes to
// out[reg][((b*H + h)*L + l)*hd + e].  32-row tiles; the (32 x N) result is staged in LDS in
// destination order -- [(reg, h)][row][e] -- so every (reg, h) slice leaves as one contiguous
// 32*hd-float run (rows of a sample are contiguous in the head-major layout).
template <int D>
__global__ __launch_bounds__(256) void k_linear_hm(const float* __restrict__ X, const float* __restrict__ Wp,
                                                   const float* __restrict__ bias, float* __restrict__ out0,
                                                   float* __restrict__ out1, float* __restrict__ out2, int M, int N,
                                                   int L, int H, int hd, unsigned hd_inv) {
  constexpr int S = lds_stride(D);
  constexpr int KS = D / 4;
  constexpr int G = dpack_groups(D);
  constexpr int R = 32;
  __shared__ __align__(16) float xs[R * S];
  __shared__ __align__(16) float stage[R * 3 * D];
  __shared__ unsigned rowbase[R];
  const int m0 = blockIdx.x * R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NT = cdiv(N, 16);
  const float4* Wq = reinterpret_cast<const float4*>(Wp);
  float4 wq[G], wn[G];
  if (wave < NT) {
#pragma unroll
    for (int g = 0; g < G; ++g) wq[g] = Wq[((size_t)wave * G + g) * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    const float4* X4 = reinterpret_cast<const float4*>(X + (size_t)m0 * D);
    const int rows_valid = min(R, M - m0);
    for (int i4 = threadIdx.x; i4 < R * D / 4; i4 += 256) {
      const int r = (4 * i4) / D, k = 4 * i4 - r * D;
      const float4 v = (r < rows_valid) ? X4[i4] : float4{0.f, 0.f, 0.f, 0.f};
      float2* dst = reinterpret_cast<float2*>(&xs[r * S + k]);
      dst[0] = float2{v.x, v.y};
      dst[1] = float2{v.z, v.w};
    }
    if (threadIdx.x < R) {
      const int m = m0 + threadIdx.x;
      const int b = m / L, l = m - b * L;
      rowbase[threadIdx.x] = (m < M) ? (unsigned)((b * H * L + l) * hd) : 0xFFFFFFFFu;
    }
  }
  __syncthreads();
  float xf[2][KS];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[mb][s] = xs[(16 * mb + (lane & 15)) * S + 4 * s + (lane >> 4)];
  for (int nt = wave; nt < NT; nt += 4) {
    if (nt + 4 < NT) {
#pragma unroll
      for (int g = 0; g < G; ++g) wn[g] = Wq[((size_t)(nt + 4) * G + g) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 q = wq[s >> 2];
      const float a = (s & 3) == 0 ? q.x : (s & 3) == 1 ? q.y : (s & 3) == 2 ? q.z : q.w;
      acc[0] = mfma16(a, xf[0][s], acc[0]);
      acc[1] = mfma16(a, xf[1][s], acc[1]);
    }
    const int n = 16 * nt + 4 * (lane >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nn = n + r;
      if (nn < N) {
        const int rh = (int)(((unsigned)nn * hd_inv) >> 16), e = nn - rh * hd;  // rh = reg*H + h  (d = H*hd)
        const float bv = bias[nn];
        stage[(rh * R + (lane & 15)) * hd + e] = acc[0][r] + bv;
        stage[(rh * R + 16 + (lane & 15)) * hd + e] = acc[1][r] + bv;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) wq[g] = wn[g];
  }
  __syncthreads();
  // copy-out: one (reg*H + h, row) pair per thread iteration = hd contiguous floats
  const int npairs = (N / hd) * R;
  for (int id = threadIdx.x; id < npairs; id += 256) {
    const int rh = id >> 5, row = id & 31;
    const unsigned rb = rowbase[row];
    if (rb == 0xFFFFFFFFu) continue;
    const int reg = rh >= 2 * H ? 2 : rh >= H ? 1 : 0;
    const int h = rh - reg * H;
    float* dst = (reg == 0 ? out0 : reg == 1 ? out1 : out2) + rb + (unsigned)(h * L * hd);
    const float* src = &stage[(rh * R + row) * hd];
    if ((hd & 1) == 0) {
      for (int e = 0; e < hd; e += 2) *reinterpret_cast<float2*>(dst + e) = *reinterpret_cast<const float2*>(src + e);
    } else {
      for (int e = 0; e < hd; ++e) dst[e] = src[e];
    }
  }
}

// out-proj + residual + LayerNorm1.  HBM-bound: 12 d bytes per row (attention output and residual in, LN1 output
// out) against 2 d^2 FLOP.  Persistent (grid = the workgroups the chip holds, two per CU), and every WAVE is its own
// stream of 16-row tiles: the weight fragments stay in registers for all tiles, the NEXT tile's X and R rows stream
// into the second half of the wave's double-buffered LDS images by LDS-DMA (global_load_lds_dwordx4, no VGPRs) while
// the current tile is computed, and the finished rows leave through the X image as whole 1 KiB runs (the accumulator
// layout would store sixteen 64-byte pieces per instruction).  A wave touches only its own images, so there is no
// workgroup barrier anywhere: the four waves of a workgroup drift apart and a stall of one (an image that has not
// landed) does not hold the others.  (The 64-row workgroup tile with a barrier per tile took 25.3 us on the ECG
// shape, one workgroup per tile 28 us.)
template <int D>
__global__ __launch_bounds__(256, 2) void k_linear_res_ln(const float* __restrict__ X, const float* __restrict__ Wp,
                                                          const float* __restrict__ bias, const float* __restrict__ R,
                                                          const float* __restrict__ gam, const float* __restrict__ bet,
                                                          float* __restrict__ Y, int M) {
  constexpr int KS = D / 4;
  constexpr int G = dpack_groups(D);
  constexpr int CT = cdiv(D, 16);
  constexpr int SX = ((D + 3) / 4) * 4 + 4;  // image row stride: whole float4 slots (lane-linear LDS-DMA image), one pad slot
  constexpr int S4 = SX / 4;
  constexpr int TR = 16;                     // rows per tile (one wave)
  constexpr int NPC = (TR * S4 + 63) / 64;   // 1 KiB DMA pieces per image
  __shared__ __align__(16) float imgs[4][2][2][TR * SX];  // [wave][buffer][X | R][row][SX] (the last DMA piece is partial: lanes past the image do not write)
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntiles = (M + TR - 1) / TR;
  // every weight fragment up front (CT*G float4 per lane), resident for all tiles
  const float4* Wq = reinterpret_cast<const float4*>(Wp);
  float4 wq[CT][G];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int g = 0; g < G; ++g) wq[ct][g] = Wq[((size_t)ct * G + g) * 64 + lane];

  // bias / LN scale / shift of this lane's columns too: an ordinary global load inside the tile loop would make hipcc
  // drain the in-flight LDS-DMA (vmcnt(0)) at its first use
  float4 bq[CT], gq[CT], eq[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int n = 16 * ct + 4 * (lane >> 4);
    const bool in = n < D;
    bq[ct] = in ? *reinterpret_cast<const float4*>(bias + n) : float4{0.f, 0.f, 0.f, 0.f};
    gq[ct] = in ? *reinterpret_cast<const float4*>(gam + n) : float4{0.f, 0.f, 0.f, 0.f};
    eq[ct] = in ? *reinterpret_cast<const float4*>(bet + n) : float4{0.f, 0.f, 0.f, 0.f};
  }
  // The DMA goes out from inline asm (dma_piece) and the kernel orders it against its LDS reads itself: for the
  // builtin form hipcc put s_waitcnt vmcnt(0) in front of every ds_read of `imgs` -- also the residual reads of the
  // CURRENT tile, right after the NEXT tile's DMA had been issued -- so nothing but the MFMAs overlapped the DMA
  // latency, and the top of the loop waited for the previous tile's stores to retire as well.
  const unsigned img_base = __builtin_amdgcn_readfirstlane(lds_addr(&imgs[0][0][0][0]));
  constexpr unsigned IMG_BYTES = TR * SX * 4;
  auto issue_dma = [&](int t, int b) {  // rows past M repeat the last valid row (never stored)
    const int rows_valid = min(TR, M - t * TR);
    const size_t base = (size_t)t * TR * D;
    const unsigned dx = img_base + (unsigned)((wave * 2 + b) * 2) * IMG_BYTES;
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) {
      const int p = pc * 64 + lane;
      const int r = p / S4, c4 = p - r * S4;
      const size_t off = base + (size_t)min(r, rows_valid - 1) * D + 4 * min(c4, D / 4 - 1);
      if (p < TR * S4) {
        dma_piece(X + off, dx + pc * 1024);
        dma_piece(R + off, dx + IMG_BYTES + pc * 1024);
      }
    }
  };
  // The parameter loads retire HERE: left to itself hipcc waits for them in front of their first use inside the
  // loop, in every iteration, with a count that also drains the DMA pieces issued just before (they share vmcnt).
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
    for (int g = 0; g < G; ++g)
      asm volatile("" : "+v"(wq[ct][g].x), "+v"(wq[ct][g].y), "+v"(wq[ct][g].z), "+v"(wq[ct][g].w));
    asm volatile("" : "+v"(bq[ct].x), "+v"(bq[ct].y), "+v"(bq[ct].z), "+v"(bq[ct].w));
    asm volatile("" : "+v"(gq[ct].x), "+v"(gq[ct].y), "+v"(gq[ct].z), "+v"(gq[ct].w));
    asm volatile("" : "+v"(eq[ct].x), "+v"(eq[ct].y), "+v"(eq[ct].z), "+v"(eq[ct].w));
  }

  constexpr int NSTI = cdiv(TR * (D / 4), 64);  // store instructions of a full tile
  int buf = 0;
  const int tstep = (int)gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  if (tile >= ntiles) return;  // (no workgroup barrier anywhere in this kernel)
  issue_dma(tile, 0);
  wait_vm<0>();
  for (; tile < ntiles; tile += tstep, buf ^= 1) {
    // this tile's images have landed: certified before the loop / at the end of the previous iteration
    float* xs = imgs[wave][buf][0];
    const float* rs = imgs[wave][buf][1];
    float xf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[s] = xs[(lane & 15) * SX + 4 * s + (lane >> 4)];
    const bool more = tile + tstep < ntiles;
    if (more) issue_dma(tile + tstep, buf ^ 1);

    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {  // k outer: CT independent accumulation chains
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float4 q = wq[ct][s >> 2];
        const float a = (s & 3) == 0 ? q.x : (s & 3) == 1 ? q.y : (s & 3) == 2 ? q.z : q.w;
        acc[ct] = mfma16(a, xf[s], acc[ct]);
      }
    }
    // epilogue: lane holds row (lane & 15) of the wave's 16, columns n = 16 ct + 4 (lane >> 4) + r
    const int lr = lane & 15;
    const float* rrow = rs + lr * SX;
    float v[CT][4];
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int n = 16 * ct + 4 * (lane >> 4);
      if (n < D) {  // D % 4 == 0: whole quads
        const float4 r4 = *reinterpret_cast<const float4*>(rrow + n);
        const float4 b4 = bq[ct];
        v[ct][0] = acc[ct][0] + b4.x + r4.x, v[ct][1] = acc[ct][1] + b4.y + r4.y;
        v[ct][2] = acc[ct][2] + b4.z + r4.z, v[ct][3] = acc[ct][3] + b4.w + r4.w;
        sum += (v[ct][0] + v[ct][1]) + (v[ct][2] + v[ct][3]);
      } else {
        v[ct][0] = v[ct][1] = v[ct][2] = v[ct][3] = 0.f;
      }
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int n = 16 * ct + 4 * (lane >> 4);
      if (n < D) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dlt = v[ct][r] - mean;
          ss = fmaf(dlt, dlt, ss);
        }
      }
    }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + 1e-5f);
    // finished rows -> the X image (its fragments are in registers)
    float* orow = xs + lr * SX;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int n = 16 * ct + 4 * (lane >> 4);
      if (n < D) {
        const float4 g4 = gq[ct], b4 = eq[ct];
        *reinterpret_cast<float4*>(orow + n) =
            float4{(v[ct][0] - mean) * rstd * g4.x + b4.x, (v[ct][1] - mean) * rstd * g4.y + b4.y,
                   (v[ct][2] - mean) * rstd * g4.z + b4.z, (v[ct][3] - mean) * rstd * g4.w + b4.w};
      }
    }
    // ... and out as contiguous runs: the wave's 16 rows are 16 * D consecutive floats of Y
    const int row0 = tile * TR;
    float4* y4 = reinterpret_cast<float4*>(Y + (size_t)row0 * D);
    const int nvalid = min(16, M - row0);
    float4 o4[NSTI];
#pragma unroll
    for (int i = 0; i < NSTI; ++i) {
      const int f = min(lane + 64 * i, 16 * (D / 4) - 1);
      const int r = f / (D / 4), c4 = f - r * (D / 4);
      o4[i] = *reinterpret_cast<const float4*>(xs + r * SX + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < NSTI; ++i) {
      const int f = lane + 64 * i;
      if (f < 16 * (D / 4) && f / (D / 4) < nvalid) y4[f] = o4[i];
    }
    // The next tile's images (issued before this tile's MFMAs) have landed; this tile's NSTI stores, younger, stay in
    // flight.  (A tile with a successor is a full tile: every one of its store instructions has active lanes.)
    if (more) wait_vm<NSTI>();
  }
}

// Row-major linear with LDS-staged output (the LSTM input gates: N = 4 d): 32-row tile, n-tiles split over the
// 4 waves with double-buffered weight fragments, the (32 x N) result staged in LDS and written as whole rows --
// every store instruction covers 1 KiB of consecutive addresses instead of sixteen 64-byte pieces (k_linear).
template <int D>
__global__ __launch_bounds__(256) void k_linear_rm(const float* __restrict__ X, const float* __restrict__ Wp,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int M, int N) {
  constexpr int S = lds_stride(D);
  constexpr int KS = D / 4;
  constexpr int G = dpack_groups(D);
  constexpr int R = 32;
  __shared__ __align__(16) float xs[R * S];
  __shared__ __align__(16) float stage[R * 4 * D];  // N <= 4 D
  const int m0 = blockIdx.x * R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NT = cdiv(N, 16);
  const float4* Wq = reinterpret_cast<const float4*>(Wp);
  float4 wq[G], wn[G];
  if (wave < NT) {
#pragma unroll
    for (int g = 0; g < G; ++g) wq[g] = Wq[((size_t)wave * G + g) * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    const float4* X4 = reinterpret_cast<const float4*>(X + (size_t)m0 * D);
    const int rows_valid = min(R, M - m0);
    for (int i4 = threadIdx.x; i4 < R * D / 4; i4 += 256) {
      const int r = (4 * i4) / D, k = 4 * i4 - r * D;
      const float4 v = (r < rows_valid) ? X4[i4] : float4{0.f, 0.f, 0.f, 0.f};
      float2* dst = reinterpret_cast<float2*>(&xs[r * S + k]);
      dst[0] = float2{v.x, v.y};
      dst[1] = float2{v.z, v.w};
    }
  }
  __syncthreads();
  float xf[2][KS];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[mb][s] = xs[(16 * mb + (lane & 15)) * S + 4 * s + (lane >> 4)];
  for (int nt = wave; nt < NT; nt += 4) {
    if (nt + 4 < NT) {
#pragma unroll
      for (int g = 0; g < G; ++g) wn[g] = Wq[((size_t)(nt + 4) * G + g) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 q = wq[s >> 2];
      const float a = (s & 3) == 0 ? q.x : (s & 3) == 1 ? q.y : (s & 3) == 2 ? q.z : q.w;
      acc[0] = mfma16(a, xf[0][s], acc[0]);
      acc[1] = mfma16(a, xf[1][s], acc[1]);
    }
    // D^T: lane holds columns n .. n+3 of rows (lane & 15) and 16 + (lane & 15)
    const int n = 16 * nt + 4 * (lane >> 4);
    if (n + 3 < N) {
      const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
      *reinterpret_cast<float4*>(&stage[(lane & 15) * N + n]) =
          float4{acc[0][0] + b4.x, acc[0][1] + b4.y, acc[0][2] + b4.z, acc[0][3] + b4.w};
      *reinterpret_cast<float4*>(&stage[(16 + (lane & 15)) * N + n]) =
          float4{acc[1][0] + b4.x, acc[1][1] + b4.y, acc[1][2] + b4.z, acc[1][3] + b4.w};
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) {
          stage[(lane & 15) * N + n + r] = acc[0][r] + bias[n + r];
          stage[(16 + (lane & 15)) * N + n + r] = acc[1][r] + bias[n + r];
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) wq[g] = wn[g];
  }
  __syncthreads();
  // copy-out: the tile's rows are consecutive in Y (row-major, ld = N): one contiguous run of rows_valid * N floats
  const int rows_valid = min(R, M - m0);
  const int total4 = rows_valid * N / 4;  // N % 4 == 0 (checked by the launcher)
  float4* dst = reinterpret_cast<float4*>(Y + (size_t)m0 * N);
  const float4* src = reinterpret_cast<const float4*>(stage);
  for (int i = threadIdx.x; i < total4; i += 256) dst[i] = src[i];
}

hipError_t launch_linear(const float* X, const float* Wp, const float* bias, float* Y, int M, int N, int D, int ldy,
                         hipStream_t s) {
  if (M <= 0) return hipSuccess;
  if (ldy == N && N % 4 == 0 && N <= 4 * D && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && cdiv(M, 32) >= 512) {
    // large row-major outputs: LDS-staged whole-row stores
    dim3 grid(cdiv(M, 32)), block(256);
    switch (D) {
#define X(d) \
      case d: hipLaunchKernelGGL(k_linear_rm<d>, grid, block, 0, s, X, Wp, bias, Y, M, N); break;
      FFD_D_LIST(X)
#undef X
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  dim3 grid(cdiv(M, 64)), block(256);
  switch (D) {
#define X(d) \
    case d: hipLaunchKernelGGL(k_linear<d>, grid, block, 0, s, X, Wp, bias, Y, M, N, ldy); break;
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_linear_hm(const float* X, const float* Wp, const float* bias, float* out0, float* out1, float* out2,
                            int M, int nreg, int D, int L, int H, int hd, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  const int N = nreg * D;
  const unsigned hd_inv = (65536u + hd - 1) / hd;  // exact floor(c / hd) for c < 2^16 / hd
  dim3 grid(cdiv(M, 32)), block(256);
  switch (D) {
#define X(d) \
    case d: hipLaunchKernelGGL(k_linear_hm<d>, grid, block, 0, s, X, Wp, bias, out0, out1, out2, M, N, L, H, hd, hd_inv); break;
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_linear_res_ln(const float* X, const float* Wp, const float* bias, const float* R, const float* g,
                                const float* beta, float* Y, int M, int D, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  if (((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(R) | reinterpret_cast<uintptr_t>(Y)) & 15) != 0)
    return hipErrorInvalidValue;
  static int resident = 0;  // persistent grid: two workgroups per CU
  if (resident == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    resident = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                prop.multiProcessorCount > 0) ? 2 * prop.multiProcessorCount : 512;
  }
  const int nwg = cdiv(cdiv(M, 16), 4);  // four 16-row wave tiles per workgroup
  dim3 grid(nwg < resident ? nwg : resident), block(256);
  switch (D) {
#define X(d) \
    case d: hipLaunchKernelGGL(k_linear_res_ln<d>, grid, block, 0, s, X, Wp, bias, R, g, beta, Y, M); break;
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace ffd
