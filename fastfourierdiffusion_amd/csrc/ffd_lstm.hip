// Residual LSTM layer of LSTMScoreModule (score_models.py:472-477, 502-504):
//     x <- x + LSTM(x)[0]      nn.LSTM(d, d, batch_first), zero initial state,
//     gates (i, f, g, o);  c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c').
// The input projection gx = x W_ih^T + (b_ih + b_hh) for all L positions is one
// MFMA GEMM (k_linear); this kernel runs the L-step recurrence.  The recurrence is
// latency-bound (L sequential cell steps), so a workgroup owns BT samples for the
// whole sequence: W_hh lives in VGPRs for all L steps (see the thread layout below),
// h lives in LDS, c in the registers of the cell threads.
#include "ffd_internal.h"

namespace ffd {

// (samples per workgroup are chosen per d_model in launch_lstm_layer)

// sigmoid / tanh from one v_exp_f32 + one v_rcp_f32 each (absolute error ~1e-7, far inside the
// parity tolerance; the library expf / tanhf cost ~10x the instructions on the critical path)
__device__ __forceinline__ float sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
  // 1 - 2/(exp(2x)+1); exp2 overflow -> rcp(inf) = 0 -> 1, underflow -> 1 - 2 = -1
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// Thread layout: tid = 4*e + j -- the lane quad of hidden unit e splits the recurrent dot products over k:
// lane j holds W_hh[gate][e][k-slice j] for all four gates (i, f, g, o) of its unit (4 x D/4 weights in VGPRs) and
// reads only its quarter of h from LDS, so the h broadcast costs a quarter of the LDS bandwidth it would with one
// full row per lane (that broadcast paced the cell step once two workgroups shared a CU).  The four partial sums
// per gate are added across the quad with two DPP xor-adds, lane j then activates gate j, the activated gates are
// exchanged with four DPP quad-broadcasts (no LDS round trip), every lane of the quad updates c/h redundantly,
// and one barrier per cell step (h is double-buffered in LDS) is all the synchronisation the recurrence needs.
template <int P>
__device__ __forceinline__ float quad_xor_add(float v) {
  // v + (value of the lane whose index inside the quad differs by xor P), P = 1 or 2
  constexpr int perm = (P == 1) ? (1 | (0 << 2) | (3 << 4) | (2 << 6)) : (2 | (3 << 2) | (0 << 4) | (1 << 6));
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), perm, 0xF, 0xF, true));
}

template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  // quad_perm:[K,K,K,K] : every lane of a quad reads lane K of that quad
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v),
                                                               K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true));
}

// Global memory is touched once per chunk of CH cell steps, not once per step: the gate
// pre-activations gx and the residual rows of the next chunk are loaded into registers while the
// current chunk runs (their latency is hidden behind CH recurrence steps), parked in LDS at the
// chunk boundary, and the inner loop works on LDS only (a per-step global load would put a full
// s_waitcnt vmcnt(0) memory round trip on the critical path of every cell step).
// A 5-wave workgroup (d = 72: 288 threads) puts two of its waves on SIMD 0; at 3 waves per SIMD a second workgroup
// then never fits next to it and the CU runs one sequence at a time (measured: time exactly doubled from B=256 to
// B=512).  Capping the VGPRs at 128 (4 waves per SIMD; the spills land on the chunk boundaries, not in the cell
// step) lets two workgroups share a CU: 3.75 -> 2.64 ms per diffusion step at B=512.
template <int D, int BT>
__global__ __launch_bounds__(((4 * D + 63) / 64) * 64, (4 * D > 256 ? 4 : 1)) void k_lstm_layer(float* __restrict__ x, const float* __restrict__ gx,
                                                    const float* __restrict__ whh, int B, int L) {
  constexpr int G4 = 4 * D;
  constexpr int CH = 16;
  constexpr int GQ = CH * G4 / 4;  // float4 per sample per chunk (gate pre-activations)
  constexpr int XQ = CH * D / 4;   // float4 per sample per chunk (residual rows)
  constexpr int NT = ((G4 + 63) / 64) * 64;
  constexpr int GPT = (GQ + NT - 1) / NT, XPT = (XQ + NT - 1) / NT;
  constexpr int KQ = ((D / 4 + 3) / 4) * 4;  // k-slice length per lane of the quad, padded to whole float4 (zeros)
  __shared__ __align__(16) float hbuf[2][BT][4 * KQ];  // h, slice-major: unit k lives at (k / (D/4)) * KQ + k % (D/4)
  __shared__ __align__(16) float gxs[BT][CH * G4];
  __shared__ __align__(16) float xsb[BT][CH * D];
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * BT;
  const bool live = tid < G4;
  static_assert(D % 4 == 0, "the quad splits k into four equal slices");
  constexpr int DQ = D / 4;
  const int e = live ? tid >> 2 : 0, gate = tid & 3;  // `gate` doubles as this lane's k-slice index j
  const int rowi = gate * D + e;  // column of the gate pre-activations this lane activates
  const bool writer = live && gate == 0;
  const int hpos = (e / DQ) * KQ + (e % DQ);  // where unit e sits in the slice-major h buffer

  float w[4][KQ];  // W_hh[g*D + e][gate*DQ + k], zero padded
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < KQ; ++k) w[g][k] = (k < DQ) ? whh[(size_t)(g * D + e) * D + gate * DQ + k] : 0.f;
  float c[BT];
#pragma unroll
  for (int bt = 0; bt < BT; ++bt) c[bt] = 0.f;
  for (int i = tid; i < 2 * BT * 4 * KQ; i += blockDim.x) (&hbuf[0][0][0])[i] = 0.f;

  // activation constants: gate 2 (g) is tanh, the others sigmoid; both are rcp(1 + exp2(s*x)) based
  const float sarg = (gate == 2) ? 2.8853900817779268f : -1.4426950408889634f;

  float4 gq[BT][GPT], xq[BT][XPT];
  auto fetch = [&](int s0) {  // chunk starting at step s0 -> registers (zero beyond L)
    const int nst = min(CH, L - s0);
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
      const int b = min(b0 + bt, B - 1);
      const float4* g4 = reinterpret_cast<const float4*>(gx + ((size_t)b * L + s0) * G4);
      const float4* x4 = reinterpret_cast<const float4*>(x + ((size_t)b * L + s0) * D);
#pragma unroll
      for (int i = 0; i < GPT; ++i) {
        const int q = tid + i * NT;
        gq[bt][i] = (q < nst * G4 / 4) ? g4[q] : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int q = tid + i * NT;
        xq[bt][i] = (q < nst * D / 4) ? x4[q] : float4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  fetch(0);
  int step = 0;
  for (int s0 = 0; s0 < L; s0 += CH) {
    const int nst = min(CH, L - s0);
    // park the prefetched chunk in LDS
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
#pragma unroll
      for (int i = 0; i < GPT; ++i) {
        const int q = tid + i * NT;
        if (q < GQ) reinterpret_cast<float4*>(gxs[bt])[q] = gq[bt][i];
      }
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int q = tid + i * NT;
        if (q < XQ) reinterpret_cast<float4*>(xsb[bt])[q] = xq[bt][i];
      }
    }
    __syncthreads();
    if (s0 + CH < L) fetch(s0 + CH);  // next chunk: in flight during the nst steps below
    for (int sl = 0; sl < nst; ++sl, ++step) {
      const int cur = step & 1;
      float p[BT][4];  // partial pre-activations of the unit's four gates over this lane's k-slice
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        const float gxv = gxs[bt][sl * G4 + rowi];  // the input contribution enters through the lane of its gate
#pragma unroll
        for (int g = 0; g < 4; ++g) p[bt][g] = (g == gate) ? gxv : 0.f;
      }
#pragma unroll
      for (int k = 0; k < KQ; k += 4) {
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
          const float4 hv = *reinterpret_cast<const float4*>(&hbuf[cur][bt][gate * KQ + k]);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            p[bt][g] = fmaf(w[g][k], hv.x, p[bt][g]);
            p[bt][g] = fmaf(w[g][k + 1], hv.y, p[bt][g]);
            p[bt][g] = fmaf(w[g][k + 2], hv.z, p[bt][g]);
            p[bt][g] = fmaf(w[g][k + 3], hv.w, p[bt][g]);
          }
        }
      }
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        float pre = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float full = quad_xor_add<2>(quad_xor_add<1>(p[bt][g]));  // sum over the quad's four k-slices
          pre = (g == gate) ? full : pre;
        }
        const float t = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(sarg * pre));
        const float a = (gate == 2) ? 1.0f - 2.0f * t : t;  // own gate, activated
        const float ai = quad_bcast<0>(a), af = quad_bcast<1>(a), ag = quad_bcast<2>(a), ao = quad_bcast<3>(a);
        c[bt] = af * c[bt] + ai * ag;
        const float h = ao * tanh_fast(c[bt]);
        if (writer) {
          hbuf[cur ^ 1][bt][hpos] = h;
          xsb[bt][sl * D + e] += h;  // residual: x <- x + LSTM(x)
        }
      }
      __syncthreads();  // LDS only inside this loop
    }
    // write the chunk's output rows back (coalesced)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
      if (b0 + bt < B) {
        float4* o4 = reinterpret_cast<float4*>(x + ((size_t)(b0 + bt) * L + s0) * D);
        for (int q = tid; q < nst * D / 4; q += blockDim.x) o4[q] = reinterpret_cast<const float4*>(xsb[bt])[q];
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Large batches: batch-tiled recurrence on the exact-fp32 matrix core, input gates fused.
//
// A workgroup (4 waves, one per SIMD) owns 16 S samples for the whole sequence.  Per cell step the gate
// pre-activations of a 16-sample tile are  G^T (4d x 16) = W_ih x_t^T + W_hh h_{t-1}^T + b  on
// v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand, resident in VGPRs for all L steps (each wave keeps the
// fragments of its own 16-row tiles: 2 x 5 x 18 registers at d = 72), so neither W_ih nor W_hh is ever re-read and
// the (B, L, 4d) gate tensor of the small-batch path (185 MB per 512 samples and layer) never exists.
// Row order inside a 16-row tile is (unit, gate) = (i >> 2, i & 3): the accumulator D[i = 4 (l >> 4) + r][j = l & 15]
// then leaves lane l with all FOUR gates (r = i, f, g, o) of unit 4 T + (l >> 4) for sample (l & 15) -- the cell update
// is lane-local, c lives in a register, and the new h of unit 4 T + q sits in the lane that, as the B operand of
// k-step s = T, must supply h[sample j][k = 4 s + q]: the same lane.  Waves own disjoint unit tiles, so h is exchanged
// through a double-buffered LDS image (one barrier per cell step); x_t fragments come straight from global memory
// (prefetched one step ahead; the four waves share the rows through L1) and double as the residual input.
// 18 tiles over 4 waves split 5 / 5 / 4 / 4: the matrix pipe of the 5-tile SIMDs paces the step.
// ---------------------------------------------------------------------------
template <int D, int S, int GRP>
__global__ __launch_bounds__(256 * GRP, GRP) void k_lstm_mfma(float* __restrict__ x, const float* __restrict__ wih,
                                                      const float* __restrict__ whh, const float* __restrict__ bsum,
                                                      int B, int L) {
  constexpr int NT = D / 4;          // unit tiles of 4 units x 4 gates == k-steps of 4
  constexpr int NTW = (NT + 3) / 4;  // tiles per wave (upper bound)
  constexpr int NG = (NT + 3) / 4;   // groups of 4 k-steps (one float4 of A fragments each)
  constexpr int HS = D + 2;          // LDS row stride: HS / 2 odd -> the 16 rows x 2 k of a 32-lane half hit 32 banks
  // W_hh fragments stay in VGPRs (they sit on the serial h chain); the W_ih fragments and the biases are streamed from
  // a wave-private LDS image, one ds_read_b128 per 4 k-steps -- with both matrices in registers (2 x 90 at d = 72)
  // hipcc has no room left to keep the x / h fragments of a step in flight and serialises every load with its use.
  extern __shared__ __align__(16) float lds[];
  float4* wlds = reinterpret_cast<float4*>(lds);            // [wave 4][g NG][tt NTW][lane 64]
  float4* blds = wlds + 4 * NG * NTW * 64;                  // [wave 4][tt NTW][lane 64]  (accumulator layout)
  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: tile ownership tests are s_cbranch, not exec masks
  // GRP = 2: waves w and w + 4 own the SAME unit tiles for two different groups of 16 S samples.  They share a SIMD
  // (a workgroup's waves are dealt to the SIMDs cyclically), so one's cell update / LDS waits run under the other's
  // MFMAs, and they share the W_ih / bias image in LDS.
  const int wave = wave8 & 3, grp = wave8 >> 2;
  float* hbuf = reinterpret_cast<float*>(blds + 4 * NTW * 64) + grp * (2 * S * 16 * HS);  // per group: [2][S][16][HS]
  const int j = lane & 15, q = lane >> 4;
  const int t0 = wave * (NT / 4) + min(wave, NT % 4);
  const int ntw = NT / 4 + (wave < NT % 4 ? 1 : 0);
  // A wave with fewer than NTW tiles runs the MFMAs of its phantom tile on zero weights: the step is paced by the
  // NTW-tile waves anyway (barrier), and the main loop stays straight-line code.

  // weight fragments (A operand: lane holds W[row(T, i = lane & 15)][k = 4 s + q]) and biases (accumulator layout)
  float wh[NTW][NT];
#pragma unroll
  for (int tt = 0; tt < NTW; ++tt) {
    const int T = min(t0 + tt, NT - 1);
    const bool on = tt < ntw;
    const size_t row = (size_t)((j & 3) * D + 4 * T + (j >> 2)) * D;
#pragma unroll
    for (int s = 0; s < NT; ++s) wh[tt][s] = on ? whh[row + 4 * s + q] : 0.f;
    if (grp == 0) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        float4 v;
        v.x = (on && 4 * g + 0 < NT) ? wih[row + 4 * (4 * g + 0) + q] : 0.f;
        v.y = (on && 4 * g + 1 < NT) ? wih[row + 4 * (4 * g + 1) + q] : 0.f;
        v.z = (on && 4 * g + 2 < NT) ? wih[row + 4 * (4 * g + 2) + q] : 0.f;
        v.w = (on && 4 * g + 3 < NT) ? wih[row + 4 * (4 * g + 3) + q] : 0.f;
        wlds[((wave * NG + g) * NTW + tt) * 64 + lane] = v;
      }
      float4 bv;
      bv.x = on ? bsum[0 * D + 4 * T + q] : 0.f;
      bv.y = on ? bsum[1 * D + 4 * T + q] : 0.f;
      bv.z = on ? bsum[2 * D + 4 * T + q] : 0.f;
      bv.w = on ? bsum[3 * D + 4 * T + q] : 0.f;
      blds[(wave * NTW + tt) * 64 + lane] = bv;
    }
  }
  for (int i = threadIdx.x & 255; i < 2 * S * 16 * HS; i += 256) hbuf[i] = 0.f;
  const float4* wl = wlds + (size_t)wave * NG * NTW * 64 + lane;
  const float4* bl = blds + (size_t)wave * NTW * 64 + lane;

  const int b0 = (blockIdx.x * GRP + grp) * (16 * S);
  float* xrow[S];
  bool live[S];
#pragma unroll
  for (int ss = 0; ss < S; ++ss) {
    const int b = b0 + 16 * ss + j;
    live[ss] = b < B;
    xrow[ss] = x + (size_t)min(b, B - 1) * L * D + q;
  }
  float c[NTW][S];
#pragma unroll
  for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
    for (int ss = 0; ss < S; ++ss) c[tt][ss] = 0.f;

  // One cell step.  xc holds x_t as B fragments (k-step s <-> units 4 s + q); the fragments of x_{t+1} are requested
  // into xn right after the input-part MFMAs were issued, a full step ahead of their use.  The time loop below is
  // unrolled by two with the roles of the two fragment sets swapped, so no register copy ties the loads to their use.
  // xr: x_t[u] of the lane's own units (residual input); a second, 5-value view of the same rows (L1 hits) -- selecting
  // them out of the fragment registers by the runtime tile index costs an 18-way v_cndmask chain per value
  auto step = [&](int t, int cur, float (&xc)[S][NT], float (&xn)[S][NT], float (&xrc)[S][NTW], float (&xrn)[S][NTW]) {
    // input part (independent of h): acc = b + W_ih x_t
    f32x4 acc[NTW][S];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
      const float4 bv = bl[tt * 64];
#pragma unroll
      for (int ss = 0; ss < S; ++ss) acc[tt][ss] = f32x4{bv.x, bv.y, bv.z, bv.w};
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float4 wv[NTW];
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) wv[tt] = wl[(g * NTW + tt) * 64];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int s = 4 * g + i;
        if (s < NT) {
#pragma unroll
          for (int tt = 0; tt < NTW; ++tt) {
            const float a = i == 0 ? wv[tt].x : i == 1 ? wv[tt].y : i == 2 ? wv[tt].z : wv[tt].w;
#pragma unroll
            for (int ss = 0; ss < S; ++ss) acc[tt][ss] = mfma16(a, xc[ss][s], acc[tt][ss]);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < L) {
#pragma unroll
      for (int ss = 0; ss < S; ++ss)
#pragma unroll
        for (int s = 0; s < NT; ++s) xn[ss][s] = xrow[ss][(size_t)(t + 1) * D + 4 * s];
#pragma unroll
      for (int ss = 0; ss < S; ++ss)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) xrn[ss][tt] = xrow[ss][(size_t)(t + 1) * D + 4 * min(t0 + tt, NT - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // h_{t-1} of every wave is in hbuf[cur]
    float hb[S][NT];
#pragma unroll
    for (int ss = 0; ss < S; ++ss)
#pragma unroll
      for (int s = 0; s < NT; ++s) hb[ss][s] = hbuf[((cur * S + ss) * 16 + j) * HS + 4 * s + q];
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
        for (int ss = 0; ss < S; ++ss) acc[tt][ss] = mfma16(wh[tt][s], hb[ss][s], acc[tt][ss]);
    // cell update, lane-local: (i, f, g, o) = acc[0..3] of unit 4 T + q, sample j
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
      if (tt < ntw) {
        const int u = 4 * (t0 + tt) + q;
#pragma unroll
        for (int ss = 0; ss < S; ++ss) {
          const f32x4 a = acc[tt][ss];
          const float gi = sigmoid_fast(a[0]), gf = sigmoid_fast(a[1]), gg = tanh_fast(a[2]), go = sigmoid_fast(a[3]);
          c[tt][ss] = gf * c[tt][ss] + gi * gg;
          const float h = go * tanh_fast(c[tt][ss]);
          hbuf[(((cur ^ 1) * S + ss) * 16 + j) * HS + u] = h;
          if (live[ss]) xrow[ss][(size_t)t * D + 4 * (t0 + tt)] = xrc[ss][tt] + h;  // x <- x + LSTM(x)
        }
      }
  };

  float xa[S][NT], xb[S][NT], xra[S][NTW], xrb[S][NTW];
#pragma unroll
  for (int ss = 0; ss < S; ++ss) {
#pragma unroll
    for (int s = 0; s < NT; ++s) xa[ss][s] = xrow[ss][4 * s];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) xra[ss][tt] = xrow[ss][4 * min(t0 + tt, NT - 1)];
  }
  __syncthreads();  // LDS images written
  for (int t = 0; t < L; t += 2) {
    step(t, 0, xa, xb, xra, xrb);
    if (t + 1 < L) step(t + 1, 1, xb, xa, xrb, xra);
  }
}

constexpr size_t lstm_mfma_lds(int D, int S, int GRP) {
  const int NT = D / 4, NTW = (NT + 3) / 4, NG = (NT + 3) / 4;
  return (size_t)(4 * NG * NTW * 64 + 4 * NTW * 64) * 16 + (size_t)GRP * 2 * S * 16 * (D + 2) * 4;
}

// batch from which the batch-tiled kernel is faster than one sample per workgroup (measured crossover)
int g_lstm_mfma_min_batch = 1536;
int g_lstm_mfma_s = 0;  // 16-sample tiles per workgroup: 0 = by batch, 1, 2

bool lstm_mfma_selected(int B, int D) { return B >= g_lstm_mfma_min_batch && D % 4 == 0 && D >= 16; }

template <int D, int S, int GRP>
static hipError_t launch_lstm_mfma_t(float* x, const float* wih, const float* whh, const float* bsum, int B, int L,
                                     hipStream_t s) {
  constexpr size_t lds = lstm_mfma_lds(D, S, GRP);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lstm_mfma<D, S, GRP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((k_lstm_mfma<D, S, GRP>), dim3(cdiv(B, 16 * S * GRP)), dim3(256 * GRP), lds, s, x, wih, whh, bsum, B, L);
  return hipGetLastError();
}

hipError_t launch_lstm_mfma(float* x, const float* wih, const float* whh, const float* bsum, int B, int L, int D,
                            hipStream_t s) {
  if (B <= 0) return hipSuccess;
  // 16 samples per workgroup (4 waves) while the 16-sample tiles do not exceed the CU count; beyond that 32 samples
  // per workgroup as two 4-wave groups (ffd_tune "lstm_mfma_s": 1 = 16, 2 = 2 x 16 in eight waves)
  const bool two = g_lstm_mfma_s ? g_lstm_mfma_s == 2 : B > 16 * 256;
  switch (D) {
#define X(d) \
  case d: return two ? launch_lstm_mfma_t<d, 1, 2>(x, wih, whh, bsum, B, L, s) : launch_lstm_mfma_t<d, 1, 1>(x, wih, whh, bsum, B, L, s);
    X(16) X(24) X(32) X(48) X(60) X(64) X(72)
#undef X
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_lstm_layer(float* x, const float* gx, const float* whh, int B, int L, int D, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  // two samples per workgroup while the W_hh row + chunk registers fit, one for d_model >= 64
  switch (D) {
#define X(d) \
    case d: hipLaunchKernelGGL((k_lstm_layer<d, (d >= 60 ? 1 : 2)>), dim3(cdiv(B, (d >= 60 ? 1 : 2))), dim3(((4 * d + 63) / 64) * 64), 0, s, \
                               x, gx, whh, B, L); break;
    FFD_D_LIST(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace ffd
