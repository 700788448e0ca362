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
// The cell step on the exact-fp32 matrix core (the building block of k_lstm_wave below; until round 4 also a kernel of
// its own, k_lstm_mfma, one layer per launch -- retired once the wavefront was pinned by the 1000-step golden: it was the
// slower or equal form at every batch size).
//
// A 16-sample tile is advanced one cell step by
//     G^T (4d x 16) = W_ih x_t^T + W_hh h_{t-1}^T + b          on v_mfma_f32_16x16x4_f32
// with the WEIGHTS as the A operand: a wave keeps the fragments of its own 16-row tiles in VGPRs for all the steps it
// runs (5 x 18 registers at d = 72; biases too) -- neither matrix is re-read from memory, and a (B, L, 4d) gate tensor
// (185 MB per 512 samples and layer) never exists.
// Row order inside a 16-row tile is (unit, gate) = (i >> 2, i & 3): the accumulator D[i = 4 (l >> 4) + r][j = l & 15]
// then leaves lane l with all FOUR gates (r = i, f, g, o) of unit 4 T + (l >> 4) for sample (l & 15) -- the cell update
// is lane-local and c lives in a register.  Waves own disjoint unit tiles (18 tiles over 4 waves: 5 / 5 / 4 / 4), so h
// is exchanged through a double-buffered LDS image with one rendezvous per step.
// x_t never travels as fragments: 256 threads move the tile's 16 rows (16 x d floats) per step as whole float4 --
// HBM -> registers three steps ahead, registers -> a 3-slot ring of LDS row images two steps ahead, B fragments from the
// image like h -- and write x_t + h_t back the same way one step later (x_t from its ring slot, h_t rows from the h
// image).  Fragment-shaped global accesses (16 rows x 16 B per instruction, every wave re-reading the rows) cost 27 %
// of a step.
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// Mid-size batches (16-sample tiles fit the chip once per layer group): the layers as a WAVEFRONT in one launch.
//
// Layer l at cell step t needs layer l-1 at step t and its own step t-1: the critical path of NL layers is
// L + (NL - 1) * lag cell steps, not NL * L.  One workgroup per (16-sample tile, layer) runs the cell step above
// over the whole sequence, in place on the same (B, L, d) buffer: a row passes through the layers in
// order, each adding its h_t.  Layer l's workgroup publishes its progress every CHP steps -- rows leave as
// write-through (sc1) stores, every wave waits for its own stores, the step's workgroup barrier, then ONE lane stores
// the step count (sc1) -- and layer l+1's workgroup of the same tile polls that word from one wave, joins its own
// step barrier and reads the rows with sc1 loads (the flag form of MI355X_MICROARCH.md, Workgroup dispatch ...: no
// fences, nothing depends on which CU or XCD a workgroup landed on).  Lower layers never wait on higher ones and the
// grid (tiles x layers <= CUs, one workgroup per CU by its LDS) is co-resident; every spin is bounded all the same.
// ---------------------------------------------------------------------------
struct LstmWaveArgs {
  const float* wih[16];
  const float* whh[16];
  const float* bsum[16];
};

// Work units.  A unit is (chunk k of T cell steps, layer, tile); unit index u = (k V + layer n_tiles + tile) with
// V = n_layers n_tiles, and workgroup p of the (resident) grid of P workgroups runs units p, p + P, p + 2 P, ... in that
// order.  P is every CU (the V pairs if there are fewer): the grid need not be a multiple of the tile count.
//   * T >= L (one chunk): the units are the (tile, layer) pairs; V <= CUs: all of them resident, a pure wavefront.
//   * V > CUs: the pairs go through the CUs in ceil(K V / P) rounds of T = L / K steps.  With K > 1 every unit hands
//     the recurrent state (h, c of its 16 samples) to the unit that continues the (tile, layer) through a state block
//     in global memory, published with the same write-through + flag protocol as the rows, and all pairs advance at
//     P / V of full speed instead of the last pass running on a fraction of the chip; the launcher picks the K with
//     the least estimated time (a unit costs ~ 15 us of hand-over).
// A unit depends on units of smaller index only ((k - 1, layer, tile) for the state, (k, layer - 1, tile) for the
// rows, polled as they are produced), and a workgroup runs its units in index order: the unfinished unit of smallest
// index is always running with its dependencies done, so the grid cannot deadlock (every spin is bounded anyway).
// Every wait on a progress word is bounded in TIME (spin_ticks of the 100 MHz real-time counter).  A wait that runs out
// is an ERROR, not a fall-through: the waiting wave writes a code to the host-visible word `err` (the library returns
// FFD_ERR_STATE for it at its next entry point / ffd_async_status) and raises the launch's abort word, after which every
// wait of every workgroup passes at once: the grid drains (on data that no longer means anything) instead of hanging.
// The protocol needs the grid's P workgroups co-resident, i.e. the device's CUs to itself (include/ffd.h).
// `fault` (tests, ffd_tune "lstm_wave_fault"): unit fault - 1 never publishes its progress.
template <int D>
__global__ __launch_bounds__(512, 2) void k_lstm_wave(float* __restrict__ x, LstmWaveArgs wa, int n_layers, int n_tiles,
                                                      int T, int B, int L, int* __restrict__ prog,
                                                      float* __restrict__ state, int* __restrict__ err,
                                                      int* __restrict__ abort_word, unsigned spin_ticks, int fault,
                                                      unsigned long long* __restrict__ trace) {
  // Eight waves, two per SIMD, in two roles (the cell step of k_lstm_mfma split in two):
  //   waves 0-3 (recurrent): acc = gx_t image; acc += W_hh h_{t-1}^T (W_hh fragments in VGPRs); lane-local cell
  //                          update; h_t -> LDS.  Only this is on the recurrence's critical path.
  //   waves 4-7 (input):     gx_{t+1} = b + W_ih x_{t+1}^T (W_ih fragments in VGPRs) -> the other gx image; the tile's
  //                          rows: global -> registers -> LDS ring three / two steps ahead, x_{t-1} + h_{t-1} back;
  //                          the wavefront's polling and publishing.
  // One workgroup barrier per cell step orders h, gx and the x ring for both roles.
  constexpr int NT = D / 4;          // unit tiles of 4 units x 4 gates == k-steps of 4
  constexpr int NTW = (NT + 3) / 4;  // tiles per wave (upper bound)
  constexpr int HS = D + 2;          // LDS row stride: HS / 2 odd -> the 16 rows x 2 k of a 32-lane half hit 32 banks
  constexpr int NF4 = 16 * NT;       // float4 slots of a 16-row tile
  constexpr int NSL = (NF4 + 255) / 256;  // slots per thread (of the 256 input-role threads)
#ifndef FFD_LSTM_CHP
#define FFD_LSTM_CHP 2
#endif
  constexpr int CHP = FFD_LSTM_CHP;  // cell steps between two publications of a layer's progress
  constexpr int SST = 16 * D + 4 * NTW * 64;  // floats of a (tile, layer) state block: h rows, then c per recurrent lane
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __align__(16) float lds[];
  float4* gxi = reinterpret_cast<float4*>(lds);       // [2][NT tiles][64 lanes] gate pre-activations, accumulator layout
  float* hbuf = lds + 2 * NT * 64 * 4;                 // [2][16][HS]  h_{t-1} / h_t
  float* xbuf = hbuf + 2 * 16 * HS;                    // [3][16][HS]  x_t rows in slot t % 3
  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool recur = wave8 < 4;
  // tile ownership: 18 unit tiles over 4 waves are 5 / 5 / 4 / 4; the input-role wave that shares a SIMD with
  // recurrent wave w (wave w + 4) takes the mirrored share, so that every SIMD carries 9 tiles per step
  const int wave = recur ? wave8 : 7 - wave8;
  const int tg = threadIdx.x & 255;
  const int j = lane & 15, q = lane >> 4;
  const int t0 = wave * (NT / 4) + min(wave, NT % 4);
  const int ntw = NT / 4 + (wave < NT % 4 ? 1 : 0);
  const int V = n_layers * n_tiles;
  const int K = (L + T - 1) / T;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(x, 0, (int)((size_t)B * L * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rst =
      __builtin_amdgcn_make_buffer_rsrc(state, 0, state ? (int)((size_t)V * SST * 4) : 0, 0x00020000);

  // this role's weight fragments (A operand: lane holds W[row(T, i = lane & 15)][k = 4 s + q]); row order inside a
  // 16-row tile is (unit, gate) = (i >> 2, i & 3), see k_lstm_mfma.  Loaded at the start of every unit (as loop-carried
  // registers, prefetched at the end of the previous unit, they cost the kernel its register budget: 1 KB of scratch).
  float wf[NTW][NT];
  f32x4 bias[NTW];
  // (round 4: from the fragment-ordered packs of k_pack_lstm_wave -- whole float4, every wave instruction one contiguous
  //  KiB: 25 + 5 loads per lane at d_model 72 where the (out, in) matrices took 90 + 20 scalar ones, each touching 16 rows)
  constexpr int S4W = (NT + 3) / 4;  // float4 groups of a tile's NT k-steps
  auto load_weights = [&](int layer) {
    const float4* wp = reinterpret_cast<const float4*>(recur ? wa.whh[layer] : wa.wih[layer]) + (size_t)wave * NTW * S4W * 64 + lane;
    const float4* bp = reinterpret_cast<const float4*>(wa.bsum[layer]) + (size_t)wave * NTW * 64 + lane;
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
#pragma unroll
      for (int g = 0; g < S4W; ++g) {
        const float4 v = wp[(tt * S4W + g) * 64];
        if (4 * g + 0 < NT) wf[tt][4 * g + 0] = v.x;
        if (4 * g + 1 < NT) wf[tt][4 * g + 1] = v.y;
        if (4 * g + 2 < NT) wf[tt][4 * g + 2] = v.z;
        if (4 * g + 3 < NT) wf[tt][4 * g + 3] = v.w;
      }
      const float4 b4 = bp[tt * 64];
      bias[tt] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
  };

  for (int u = blockIdx.x; u < K * V; u += gridDim.x) {
  // trace (diagnostics, ffd_lstm_trace; nullptr otherwise): per unit {real time at its start, after its start-up (barrier
  // C), at its end, ticks spent waiting on progress words | workgroup << 48}, 100 MHz ticks, written by one lane
  const unsigned long long tr0 = trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  unsigned long long tr1 = 0ull, tr_wait = 0ull;
  const int kc = u / V, rem = u - kc * V;
  const int layer = rem / n_tiles, tile = rem - layer * n_tiles;  // (uniform)
  const int tb = kc * T, te = min(L, tb + T);                     // this unit's cell steps [tb, te); tb is even
  int* my_prog = prog + layer * n_tiles + tile;
  const bool mute = fault > 0 && u == fault - 1;  // (tests) this unit withholds its progress
  const int* up_prog = layer > 0 ? prog + (layer - 1) * n_tiles + tile : nullptr;
  const unsigned st_base = (unsigned)((layer * n_tiles + tile) * SST * 4);  // byte offset of the state block
  load_weights(layer);
  if (kc == 0)
    for (int i = threadIdx.x; i < 2 * 16 * HS; i += 512) hbuf[i] = 0.f;

  if (recur) {
    // ------------------------------------------------------------------ recurrent role
    float c[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) c[tt] = 0.f;
    __syncthreads();  // A (the input role has seen this (tile, layer)'s previous chunk published)
    if (kc > 0) {
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt)
        c[tt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rst, st_base + (unsigned)((16 * D + (wave * NTW + tt) * 64 + lane) * 4), 0, 16));  // sc1
    }
    __syncthreads();  // B
    __syncthreads();  // C: gx_tb is in image 0, h_{tb-1} in h image 0
    for (int t = tb; t < te; ++t) {
      const int cur = t & 1;
      if (t > tb) __syncthreads();  // h_{t-1} (h image cur) and gx_t (image cur) are complete
      f32x4 acc[NTW];
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) {
        const float4 g4 = gxi[(cur * NT + min(t0 + tt, NT - 1)) * 64 + lane];
        acc[tt] = f32x4{g4.x, g4.y, g4.z, g4.w};
      }
      float hb[NT];
#pragma unroll
      for (int s = 0; s < NT; ++s) hb[s] = hbuf[(cur * 16 + j) * HS + 4 * s + q];
#pragma unroll
      for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) acc[tt] = mfma16(wf[tt][s], hb[s], acc[tt]);
      // cell update, lane-local: (i, f, g, o) = acc[0..3] of unit 4 T + q, sample j
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt)
        if (tt < ntw) {
          const f32x4 a = acc[tt];
          const float gi = sigmoid_fast(a[0]), gf = sigmoid_fast(a[1]), gg = tanh_fast(a[2]), go2 = sigmoid_fast(a[3]);
          c[tt] = gf * c[tt] + gi * gg;
          hbuf[((cur ^ 1) * 16 + j) * HS + 4 * (t0 + tt) + q] = go2 * tanh_fast(c[tt]);
        }
    }
    if (te < L) {  // the cell state goes to the unit that continues this (tile, layer)
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c[tt]), rst,
                                              st_base + (unsigned)((16 * D + (wave * NTW + tt) * 64 + lane) * 4), 0, 16);  // sc1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();  // h_{te-1} complete (and this role's state stores have left)
    __syncthreads();  // (the input role's last stores have left)
    continue;         // next unit (both roles pass the same te - tb + 4 barriers per unit)
  }

  // -------------------------------------------------------------------- input role
  // this thread's float4 slots of the tile's rows: slot f -> row f / NT, columns 4 (f % NT) .. +3; byte offsets
  // into the (B, L, d) buffer for the raw-buffer (sc1) loads and stores
  const int b0 = tile * 16;
  unsigned go[NSL];  // byte offset of the slot at t = 0 (sample clamped; stores masked by `ok`)
  unsigned so[NSL];  // byte offset of the slot's h values in the state block
  int lo[NSL];
  bool has[NSL], ok[NSL];
#pragma unroll
  for (int k = 0; k < NSL; ++k) {
    const int f = tg + 256 * k;
    has[k] = f < NF4;
    const int r = min(f, NF4 - 1) / NT, c4 = min(f, NF4 - 1) - r * NT;
    ok[k] = has[k] && b0 + r < B;
    go[k] = (unsigned)(((size_t)min(b0 + r, B - 1) * L * D + 4 * c4) * 4);
    so[k] = st_base + (unsigned)((r * D + 4 * c4) * 4);
    lo[k] = r * HS + 4 * c4;
  }
  auto gload = [&](int t, float4 (&dst)[NSL]) {
#pragma unroll
    for (int k = 0; k < NSL; ++k)
      if (has[k]) {
        // (whole-vector bit cast: with a per-component __builtin_bit_cast(float, v.x) hipcc / ROCm 7.2 shrinks the load
        //  to buffer_load_dword and still reads four registers -- tools/probes/probe_buffer_sc1.hip)
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, go[k] + (unsigned)t * (D * 4), 0, 16));  // sc1
        dst[k] = float4{v[0], v[1], v[2], v[3]};
      }
  };
  auto xput = [&](int par, const float4 (&src)[NSL]) {
#pragma unroll
    for (int k = 0; k < NSL; ++k)
      if (has[k]) {
        float2* d2 = reinterpret_cast<float2*>(xbuf + par * 16 * HS + lo[k]);
        d2[0] = float2{src[k].x, src[k].y};
        d2[1] = float2{src[k].z, src[k].w};
      }
  };
  auto out_store = [&](int t, int par, int xs) {  // x_t + h_t -> the buffer (write-through)
#pragma unroll
    for (int k = 0; k < NSL; ++k)
      if (ok[k]) {
        const float2* h2 = reinterpret_cast<const float2*>(hbuf + par * 16 * HS + lo[k]);
        const float2* x2 = reinterpret_cast<const float2*>(xbuf + xs * 16 * HS + lo[k]);
        const float2 a = h2[0], b = h2[1], u2 = x2[0], v = x2[1];
        const f32x4 o = f32x4{u2.x + a.x, u2.y + a.y, v.x + b.x, v.y + b.y};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs, go[k] + (unsigned)t * (D * 4), 0, 16);  // sc1
      }
  };
  // wave 4 waits until `word` has reached `need` (the other waves meet it at the next workgroup barrier)
  auto await_word = [&](const int* word, int& known, int need) {
    if (wave8 == 4 && known < need) {
      const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
      for (unsigned spin = 0;; ++spin) {
        known = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (known >= need) break;
        bool out = false;
        if ((spin & 15) == 15) {  // the abort word (another wait has timed out) and the clock, every 16th poll
          out = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
          if (!out && __builtin_amdgcn_s_memrealtime() - t_start > (unsigned long long)spin_ticks) {
            out = true;
            if (lane == 0) {
              __hip_atomic_store(err, 1 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // which unit gave up
              __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        if (out) {
          known = 1 << 30;  // this unit's later waits pass: the launch drains, the host reports the error
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      if (trace) tr_wait += __builtin_amdgcn_s_memrealtime() - t_start;
    }
  };
  int known = layer > 0 ? 0 : L;  // cell steps the layer below is known to have published
  auto await_rows = [&](int need) {
    if (layer > 0) await_word(up_prog, known, need);
  };
  // gx image `img` <- b + W_ih x^T for this wave's tiles, x fragments from ring slot `par`
  auto input_part = [&](int par, int img) {
    float xf[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) xf[s] = xbuf[(par * 16 + j) * HS + 4 * s + q];
    f32x4 acc[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) acc[tt] = bias[tt];
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) acc[tt] = mfma16(wf[tt][s], xf[s], acc[tt]);
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
      if (tt < ntw) gxi[(img * NT + t0 + tt) * 64 + lane] = float4{acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]};
  };

  int nld = 0;  // row-load instructions this wave issues per step (a slot index past the tile issues none)
#pragma unroll
  for (int k = 0; k < NSL; ++k) nld += (256 * k + 64 * (wave8 & 3) < NF4) ? 1 : 0;
  float4 xn[NSL];
  if (kc > 0) {  // the unit that ran steps < tb of this (tile, layer) has published them, its state included
    int mine = 0;
    await_word(my_prog, mine, tb);
  }
  // rows tb .. tb+3 are requested before the loop's first barrier-protected wait: all four are awaited HERE, in front of
  // barrier A (a wait inside the t == tb iteration would let waves 5-7 request row tb+3 before wave 4 has seen it
  // published; with publications every CHP = 2 steps from an even tb this wait costs nothing extra)
  await_rows(min(tb + 4, te));
  __syncthreads();  // A
  if (kc > 0) {  // h_{tb-1} -> h image 0 (tb is even)
#pragma unroll
    for (int k = 0; k < NSL; ++k)
      if (has[k]) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rst, so[k], 0, 16));  // sc1
        float2* d2 = reinterpret_cast<float2*>(hbuf + lo[k]);
        d2[0] = float2{v[0], v[1]};
        d2[1] = float2{v[2], v[3]};
      }
  }
  {  // the first three rows' loads in flight together (one memory latency, not three)
    float4 xa[NSL], xb[NSL];
    gload(tb, xa);
    if (tb + 1 < te) gload(tb + 1, xb);
    if (tb + 2 < te) gload(tb + 2, xn);
    xput(0, xa);
    if (tb + 1 < te) xput(1, xb);
  }
  __syncthreads();  // B: ring slots 0, 1 written
  input_part(0, 0);
  __syncthreads();  // C
  if (trace) tr1 = __builtin_amdgcn_s_memrealtime();
  int s0 = 0, s1 = 1, s2 = 2;  // ring slots of x_t, x_{t+1}, x_{t+2} (= the slot x_{t-1} occupied)
  for (int t = tb; t < te; ++t) {
    const int cur = t & 1;
    // publication due: rows tb .. t-2 were stored during earlier iterations; each wave retires its own stores (its
    // `nld` row loads of step t+2 are younger and may stay in flight), the barrier collects the waves, one lane signs
    const bool publish = t >= tb + 2 && (t - 1) % CHP == 0;
    if (t > tb) {
      if (publish) {
        if (t + 2 >= te) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nld == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (nld == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      await_rows(min(t + 4, te));
      __syncthreads();  // h_{t-1}, gx_t, x_{t+1} (ring slot s1) complete
      if (publish && tg == 0 && !mute) __hip_atomic_store(my_prog, t - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      out_store(t - 1, cur, s2);  // x_{t-1} + h_{t-1}
    }
    if (t + 1 < te) input_part(s1, cur ^ 1);  // gx_{t+1}
    // slot s2 held x_{t-1}: its fragments were read two steps ago and this thread just wrote its rows back
    if (t + 2 < te) xput(s2, xn);
    if (t + 3 < te) gload(t + 3, xn);
    const int r = s0;
    s0 = s1, s1 = s2, s2 = r;
  }
  __syncthreads();  // h_{te-1} complete
  out_store(te - 1, te & 1, s2);
  if (te < L) {  // h_{te-1} (h image te & 1) -> the state block
#pragma unroll
    for (int k = 0; k < NSL; ++k)
      if (has[k]) {
        const float2* h2 = reinterpret_cast<const float2*>(hbuf + (te & 1) * 16 * HS + lo[k]);
        const float2 a = h2[0], b = h2[1];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{a.x, a.y, b.x, b.y}), rst, so[k], 0, 16);  // sc1
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tg == 0 && !mute) __hip_atomic_store(my_prog, te, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (trace && tg == 0) {
    unsigned long long* o = trace + 4 * (size_t)u;
    o[0] = tr0, o[1] = tr1, o[2] = __builtin_amdgcn_s_memrealtime(), o[3] = tr_wait | ((unsigned long long)blockIdx.x << 48);
  }
  }  // units
}

// Fragment-ordered weights of k_lstm_wave: per role (W_hh, W_ih) [wave 4][tile slot NTW][k group S4W][lane 64][4] =
// W[row(lane & 15, tile)][4 (4 g + j) + (lane >> 4)] with row = (gate = i & 3) D + 4 tile + (i >> 2) -- the A operand of the
// cell step, tile = the wave's tt-th unit tile (5 / 5 / 4 / 4 of 18 at d_model 72; slots past a wave's share are zero) -- and
// the summed bias [wave][slot][lane][gate r] = b[r D + 4 tile + (lane >> 4)].
template <int D>
__global__ void k_pack_lstm_wave(const float* __restrict__ wih, const float* __restrict__ whh, const float* __restrict__ bsum,
                                 float* __restrict__ ih_out, float* __restrict__ hh_out, float* __restrict__ b_out) {
  constexpr int NT = D / 4, NTW = (NT + 3) / 4, S4W = (NT + 3) / 4;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of each role's pack
  if (idx >= 4 * NTW * S4W * 64) return;
  const int lane = idx & 63, g = (idx >> 6) % S4W, tt = ((idx >> 6) / S4W) % NTW, wave = (idx >> 6) / (S4W * NTW);
  const int t0 = wave * (NT / 4) + min(wave, NT % 4), ntw = NT / 4 + (wave < NT % 4 ? 1 : 0);
  const int Tt = min(t0 + tt, NT - 1), j = lane & 15, q = lane >> 4;
  const bool on = tt < ntw;
  const size_t row = (size_t)((j & 3) * D + 4 * Tt + (j >> 2)) * D;
  float a[4], b[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int s = 4 * g + c;
    const bool in = on && s < NT;
    a[c] = in ? wih[row + 4 * s + q] : 0.f;
    b[c] = in ? whh[row + 4 * s + q] : 0.f;
  }
  reinterpret_cast<float4*>(ih_out)[idx] = float4{a[0], a[1], a[2], a[3]};
  reinterpret_cast<float4*>(hh_out)[idx] = float4{b[0], b[1], b[2], b[3]};
  if (g == 0) {
    float4 bb{0.f, 0.f, 0.f, 0.f};
    if (on) bb = float4{bsum[0 * D + 4 * Tt + q], bsum[1 * D + 4 * Tt + q], bsum[2 * D + 4 * Tt + q], bsum[3 * D + 4 * Tt + q]};
    reinterpret_cast<float4*>(b_out)[(wave * NTW + tt) * 64 + lane] = bb;
  }
}

// floats of one role's weight pack / of the bias pack
size_t lstm_wave_wpack_floats(int D) { return (size_t)4 * ((D / 4 + 3) / 4) * ((D / 4 + 3) / 4) * 64 * 4; }
size_t lstm_wave_bpack_floats(int D) { return (size_t)4 * ((D / 4 + 3) / 4) * 64 * 4; }
hipError_t launch_pack_lstm_wave(const float* wih, const float* whh, const float* bsum, float* ih_out, float* hh_out,
                                 float* b_out, int D, hipStream_t s) {
  const int n4 = (int)(lstm_wave_wpack_floats(D) / 4);
  switch (D) {
#define X(d) \
  case d: hipLaunchKernelGGL(k_pack_lstm_wave<d>, dim3(cdiv(n4, 256)), dim3(256), 0, s, wih, whh, bsum, ih_out, hh_out, b_out); break;
    X(16) X(24) X(32) X(48) X(60) X(64) X(72)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

constexpr size_t lstm_wave_lds(int D) { return (size_t)(2 * (D / 4) * 64 * 4 + 5 * 16 * (D + 2)) * 4; }

thread_local int g_lstm_wave_persist = 1;  // 1: one launch, workgroups run their units in order (ffd_tune "lstm_wave_persist"); 0: a launch per layer group
thread_local int g_lstm_wave_per = 0;      // > 0: at most this many layers in flight (tests)
thread_local int g_lstm_wave_chunk = 0;    // cell steps per unit where the (tile, layer) pairs outnumber the CUs: 0 by the pass count, 1 never, even n forced
thread_local int g_lstm_wave_fault = 0;    // tests: unit fault - 1 never publishes its progress (the waits on it must time out as an ERROR)
thread_local int g_lstm_wave_spin_ms = 2000;  // time limit of one wait on a progress word (ffd_tune "lstm_wave_spin_ms")
thread_local int g_lstm_wave = 1;  // 1 (or 2): the layer-wavefront kernel, the LSTM path at every batch; 0: the per-layer kernels k_linear_rm + k_lstm_layer (tests' cross-check)

// (batches past a 16-sample tile per CU go through the launcher in sub-batches of 16 CUs samples)
// ... and so that a sub-batch's rows stay inside one raw-buffer resource (< 2^31 bytes; long sequences)
int lstm_wave_max_batch(int L, int D) {
  const long long by_bytes = ((1ll << 31) - 1) / ((long long)L * D * 4) / 16 * 16;
  const long long by_cus = 16ll * num_cus();
  return (int)(by_bytes < by_cus ? by_bytes : by_cus);
}
bool lstm_wave_selected(int B, int D) { return g_lstm_wave != 0 && D % 4 == 0 && D >= 16; }

// floats of the state blocks (h rows + cell values per recurrent lane) of the time-chunked form, <= 16 layers per launch
size_t lstm_wave_state_floats(int B, int D, int NL) {
  const int nl = NL < 16 ? NL : 16;
  return (size_t)nl * cdiv(B, 16) * (16 * D + 4 * ((D / 4 + 3) / 4) * 64);
}

template <int D>
static hipError_t launch_lstm_wave_t(float* x, const float* const* wih, const float* const* whh, const float* const* bsum,
                                     int NL, int B, int L, int* prog_all, float* state, int* err, hipStream_t s,
                                     unsigned long long* trace) {
  int* abort_word = prog_all;  // word 0 of the buffer; the progress words start one 64-byte line further
  int* prog = prog_all + 16;
  const unsigned spin_ticks = (unsigned)g_lstm_wave_spin_ms * 100000u;  // 100 MHz ticks
  constexpr size_t lds = lstm_wave_lds(D);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lstm_wave<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const int n_tiles = cdiv(B, 16);
  if (n_tiles > num_cus()) return hipErrorInvalidValue;  // (the caller hands over sub-batches of a tile per CU)
  int per = num_cus() / n_tiles;  // whole layers the resident workgroups hold
  if (g_lstm_wave_per > 0 && g_lstm_wave_per < per) per = g_lstm_wave_per;
  if (per > 16) per = 16;
  const int Lfull = L + (L & 1);  // one chunk (an even step count >= L)
  // up to 16 layers per launch (the argument block)
  for (int l0 = 0; l0 < NL; l0 += 16) {
    const int nl = NL - l0 < 16 ? NL - l0 : 16;
    if (!g_lstm_wave_persist) {  // a launch per group of `per` layers, every workgroup one (tile, layer)
      for (int l1 = 0; l1 < nl; l1 += per) {
        const int n1 = nl - l1 < per ? nl - l1 : per;
        LstmWaveArgs wa{};
        for (int i = 0; i < n1; ++i) wa.wih[i] = wih[l0 + l1 + i], wa.whh[i] = whh[l0 + l1 + i], wa.bsum[i] = bsum[l0 + l1 + i];
        hipError_t e = hipMemsetAsync(prog_all, 0, sizeof(int) * (16 + (size_t)n1 * n_tiles), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_lstm_wave<D>), dim3(n1 * n_tiles), dim3(512), lds, s, x, wa, n1, n_tiles, Lfull, B, L, prog,
                           (float*)nullptr, err, abort_word, spin_ticks, g_lstm_wave_fault, trace);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
      }
      continue;
    }
    // One launch: P resident workgroups run the K V units (chunk, layer, tile), V = nl n_tiles, in index order.  P is
    // every CU (or the V pairs if fewer; tests: `per` layers' worth); the chunk count K minimises the estimate
    // rounds x (steps of a chunk x ~3 us + ~15 us of hand-over for K > 1), rounds = ceil(K V / P): B = 512 (V = 320):
    // K = 4 -> 5 rounds of 64 steps instead of 2 passes of 251; B = 1536 (V = 960): K = 1 -> 4 passes instead of the
    // 5 that whole layers per workgroup (2 x 96 of 256 CUs) took.
    const int V = nl * n_tiles;
    const int cap = g_lstm_wave_per > 0 ? per * n_tiles : num_cus();
    const int P = V < cap ? V : cap;
    int K = 1;
    if (g_lstm_wave_chunk >= 2) {
      K = cdiv(L, g_lstm_wave_chunk & ~1);
    } else if (g_lstm_wave_chunk == 0 && state != nullptr && V > P) {
      double best = 0.0;
      for (int k = 1; k <= 8; ++k) {
        const int Tk = (cdiv(L, k) + 1) & ~1;
        if (k > 1 && Tk >= L) break;
        const double est = (double)cdiv(k * V, P) * (Tk * 3.0 + (k > 1 ? 15.0 : 0.0));
        if (best == 0.0 || est < best * 0.98) best = est, K = k;
      }
    }
    const int Tc = g_lstm_wave_chunk >= 2 ? (g_lstm_wave_chunk & ~1) : (cdiv(L, K) + 1) & ~1;
    const bool chunked = K > 1 && state != nullptr && Tc < L;
    LstmWaveArgs wa{};
    for (int i = 0; i < nl; ++i) wa.wih[i] = wih[l0 + i], wa.whh[i] = whh[l0 + i], wa.bsum[i] = bsum[l0 + i];
    hipError_t e = hipMemsetAsync(prog_all, 0, sizeof(int) * (16 + (size_t)nl * n_tiles), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_lstm_wave<D>), dim3(P), dim3(512), lds, s, x, wa, nl, n_tiles, chunked ? Tc : Lfull, B, L, prog,
                       chunked ? state : (float*)nullptr, err, abort_word, spin_ticks, g_lstm_wave_fault, trace);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_lstm_wave(float* x, const float* const* wih, const float* const* whh, const float* const* bsum, int NL,
                            int B, int L, int D, int* prog, float* state, int* err, hipStream_t s, unsigned long long* trace) {
  if (B <= 0 || NL <= 0) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (size_t)B * L * D * 4 >= (1ull << 31) || err == nullptr) return hipErrorInvalidValue;
  switch (D) {
#define X(d) \
  case d: return launch_lstm_wave_t<d>(x, wih, whh, bsum, NL, B, L, prog, state, err, s, trace);
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
