// Packed ortho real FFT / inverse along dim 1 of (B, L, C)  (fourier.py:8-94).
//
// One workgroup owns one sample's (L x CG) slab: a single coalesced HBM read into
// LDS, every Stockham autosort pass ping-pongs between two LDS buffers, a single
// coalesced HBM write of the packed spectrum -- 8 B of HBM traffic per element.
// Any length: L is factored into radices (8/4/2 for powers of two, odd primes as
// they come, e.g. 187 = 11 * 17, 251 = 251) and each pass evaluates its radix-R
// butterflies output-by-output straight from one length-L twiddle table
//     w_R^{jk} = W_L[(jk mod R) L/R]      w_n^{pk} = W_L[p k s]      (n s = L)
// computed on the host in double precision.
//
// Packed layout (fourier.py:24-47): out[0 .. L/2] = Re X_k, out[L/2+1 ..] = Im X_k for
// k = 1 .. ceil(L/2)-1; scale 1/sqrt(L) both ways.
//
// Power-of-two lengths (the config-5 shard, L = 512) take the k_rfft_pow2 path further down: a REAL transform
// as a half-length complex FFT plus a split pass, whole radix-8/4/2 butterflies in registers, float4 slab I/O.
#include <math.h>

#include <algorithm>

#include <map>
#include <mutex>
#include <vector>

#include "ffd_internal.h"

namespace ffd {

struct FftPlan {
  int npass;
  int radix[16];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// All Stockham passes of a length-L complex FFT over cg channels (channel-innermost LDS image
// of stride CG); returns the buffer holding the result.  W holds the (possibly conjugated) twiddles.
__device__ __forceinline__ float2* stockham(float2* x, float2* y, const float2* W, const FftPlan& plan, int L, int CG,
                                            int cg) {
  int n = L, s = 1;
  for (int ps = 0; ps < plan.npass; ++ps) {
    const int R = plan.radix[ps];
    const int m = n / R;
    const int LR = L / R;
    for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
      const int i = idx / cg, cc = idx - i * cg;
      const int q = i % s;
      const int t = i / s;
      const int k = t % R;
      const int p = t / R;
      const float2* xp = x + (size_t)(q + s * p) * CG + cc;
      const int stride = s * m * CG;
      float2 acc = xp[0];
      int tw = 0;  // (j*k mod R) * L/R
      for (int j = 1; j < R; ++j) {
        tw += k * LR;
        if (tw >= L) tw -= L;
        float2 a = xp[(size_t)j * stride];
        float2 w = W[tw];
        acc.x = fmaf(a.x, w.x, fmaf(-a.y, w.y, acc.x));
        acc.y = fmaf(a.x, w.y, fmaf(a.y, w.x, acc.y));
      }
      acc = cmul(acc, W[p * k * s]);
      y[(size_t)i * CG + cc] = acc;
    }
    __syncthreads();
    float2* tmp = x;
    x = y;
    y = tmp;
    n = m;
    s *= R;
  }
  return x;
}

template <bool INVERSE>
__global__ __launch_bounds__(256) void k_fft(const float* __restrict__ in, float* __restrict__ out,
                                             const float2* __restrict__ Wg, FftPlan plan, int L, int C, int CG,
                                             float scale, const float* __restrict__ a0,
                                             const float* __restrict__ a1) {
  // a0 / a1: the optional affine wrappers of k_rfft_pow2 (forward: (X - a0) / a1; inverse: x * a0 + a1 on the input)
  extern __shared__ __align__(16) float2 sm[];
  float2* W = sm;               // L twiddles
  float2* bufA = sm + L;        // L * CG
  float2* bufB = bufA + L * CG;
  const int b = blockIdx.x;
  const int c0 = blockIdx.y * CG;
  const int cg = min(CG, C - c0);
  const float* src = in + (size_t)b * L * C;
  float* dst = out + (size_t)b * L * C;
  const int n_re = L / 2 + 1;

  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    float2 w = Wg[i];
    if (INVERSE) w.y = -w.y;
    W[i] = w;
  }
  if (!INVERSE) {
    for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
      int l = idx / cg, cc = idx - l * cg;
      bufA[l * CG + cc] = make_float2(src[(size_t)l * C + c0 + cc], 0.f);
    }
  } else {
    // rebuild the Hermitian spectrum X[k], k = 0..L-1 from the packed layout
    for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
      int k = idx / cg, cc = idx - k * cg;
      int kk = (k < n_re) ? k : L - k;  // source harmonic
      const size_t gr = (size_t)kk * C + c0 + cc;
      float re = src[gr];
      if (a0) re = __fadd_rn(__fmul_rn(re, a0[gr]), a1[gr]);
      float im = 0.f;
      if (kk >= 1 && kk <= L - n_re) {
        const size_t gi = (size_t)(n_re + kk - 1) * C + c0 + cc;
        im = src[gi];
        if (a0) im = __fadd_rn(__fmul_rn(im, a0[gi]), a1[gi]);
      }
      if (k >= n_re) im = -im;
      bufA[k * CG + cc] = make_float2(re, im);
    }
  }
  __syncthreads();

  float2* x = stockham(bufA, bufB, W, plan, L, CG, cg);

  if (!INVERSE) {
    for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
      int o = idx / cg, cc = idx - o * cg;
      float v;
      if (o < n_re) v = x[o * CG + cc].x;
      else v = x[(o - n_re + 1) * CG + cc].y;
      const size_t go = (size_t)o * C + c0 + cc;
      v *= scale;
      if (a0) v = __fdiv_rn(__fsub_rn(v, a0[go]), a1[go]);
      dst[go] = v;
    }
  } else {
    for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
      int o = idx / cg, cc = idx - o * cg;
      dst[(size_t)o * C + c0 + cc] = x[o * CG + cc].x * scale;
    }
  }
}

// ---------------------------------------------------------------------------
// FreSca (fresca.py:111-268): score <- irfft( (l*low + h*high) (.) rfft(score) ), low = [k <= Rc].
//   k_fresca_spectrum : per sample, partial[b][k] = sum_c |X[b,k,c]|   (ortho rfft, k = 0..L/2)
//   k_fresca_cutoff   : spec[k] = mean_{b,c}; Rc = first k with cumsum >= r0 * sum  (fresca.py:46-58)
//   k_fresca_apply    : FFT -> scale each bin by l or h -> inverse FFT, all in LDS, Rc read on device
// The energy cutoff is a batch-wide statistic, reduced in a fixed order (deterministic).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fresca_spectrum(const float* __restrict__ in, float* __restrict__ partial,
                                                         const float2* __restrict__ Wg, FftPlan plan, int L, int C,
                                                         int CG, float scale) {
  extern __shared__ __align__(16) float2 sm[];
  float2* W = sm;
  float2* bufA = sm + L;
  float2* bufB = bufA + L * CG;
  const int b = blockIdx.x;
  const int c0 = blockIdx.y * CG;
  const int cg = min(CG, C - c0);
  const float* src = in + (size_t)b * L * C;
  for (int i = threadIdx.x; i < L; i += blockDim.x) W[i] = Wg[i];
  for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
    const int l = idx / cg, cc = idx - l * cg;
    bufA[l * CG + cc] = make_float2(src[(size_t)l * C + c0 + cc], 0.f);
  }
  __syncthreads();
  float2* x = stockham(bufA, bufB, W, plan, L, CG, cg);
  const int nf = L / 2 + 1;
  for (int k = threadIdx.x; k < nf; k += blockDim.x) {
    float acc = 0.f;
    for (int c = 0; c < cg; ++c) {
      const float2 v = x[k * CG + c];
      acc += sqrtf(fmaf(v.x * scale, v.x * scale, (v.y * scale) * (v.y * scale)));
    }
    partial[((size_t)b * gridDim.y + blockIdx.y) * nf + k] = acc;
  }
}

__global__ void k_fresca_cutoff(const float* __restrict__ partial, int* __restrict__ rc_out, int B, int nrows, int nf,
                                int C, double cutoff_ratio) {
  extern __shared__ float spec[];
  for (int k = threadIdx.x; k < nf; k += blockDim.x) {
    float acc = 0.f;
    for (int b = 0; b < nrows; ++b) acc += partial[(size_t)b * nf + k];
    spec[k] = acc / (float)(B * C);  // mean over batch and channels (fresca.py:150)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float etot = 0.f;
    for (int k = 0; k < nf; ++k) etot += spec[k];  // torch.abs(freq_spectrum).sum() in fp32
    const double thr = cutoff_ratio * (double)etot;
    double cum = 0.0;  // python float accumulation of .item() values
    int rc = 0;
    for (int k = 0; k < nf; ++k) {
      cum += (double)spec[k];
      if (cum >= thr) {
        rc = k;
        break;
      }
    }
    *rc_out = rc;
  }
}

__global__ __launch_bounds__(256) void k_fresca_apply(const float* __restrict__ in, float* __restrict__ out,
                                                      const float2* __restrict__ Wg, FftPlan plan, int L, int C,
                                                      int CG, const int* __restrict__ rc_dev, float rc_host,
                                                      float low, float high, float scale2) {
  extern __shared__ __align__(16) float2 sm[];
  float2* W = sm;
  float2* bufA = sm + L;
  float2* bufB = bufA + L * CG;
  const int b = blockIdx.x;
  const int c0 = blockIdx.y * CG;
  const int cg = min(CG, C - c0);
  const float* src = in + (size_t)b * L * C;
  float* dst = out + (size_t)b * L * C;
  for (int i = threadIdx.x; i < L; i += blockDim.x) W[i] = Wg[i];
  for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
    const int l = idx / cg, cc = idx - l * cg;
    bufA[l * CG + cc] = make_float2(src[(size_t)l * C + c0 + cc], 0.f);
  }
  __syncthreads();
  float2* x = stockham(bufA, bufB, W, plan, L, CG, cg);
  // low-pass set: k <= Rc (energy: integer index from the cutoff kernel; spatial: Rc = r0 * n_freq)
  const float rc = rc_dev ? (float)(*rc_dev) : rc_host;
  for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
    const int k = idx / cg, cc = idx - k * cg;
    const int kk = (k <= L / 2) ? k : L - k;  // Hermitian partner shares its bin's factor
    const float f = ((float)kk <= rc) ? low : high;
    float2 v = x[k * CG + cc];
    x[k * CG + cc] = make_float2(v.x * f, v.y * f);
  }
  for (int i = threadIdx.x; i < L; i += blockDim.x) W[i].y = -W[i].y;  // conjugate twiddles: inverse transform
  __syncthreads();
  float2* other = (x == bufA) ? bufB : bufA;
  float2* y = stockham(x, other, W, plan, L, CG, cg);
  for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
    const int l = idx / cg, cc = idx - l * cg;
    dst[(size_t)l * C + c0 + cc] = y[l * CG + cc].x * scale2;
  }
}

// ---------------------------------------------------------------------------
// Power-of-two lengths: real transform = complex FFT of half the length + split pass.
//
//   z[n] = x[2n] + i x[2n+1]  (n < N = L/2);  Z = FFT_N(z);  with W = exp(-2 pi i / L):
//   forward  X[k]   = E + W^k D,  X[N-k] = conj(E - W^k D),  E = (Z[k] + conj Z[N-k]) / 2,  D = (Z[k] - conj Z[N-k]) / 2i
//   inverse  Z[k]   = S + i Q,    Z[N-k] = conj(S - i Q),    S = X[k] + conj X[N-k],  Q = conj(W^k) (X[k] - conj X[N-k])
// One workgroup owns one sample's (L x CG) slab.  LDS holds the length-L twiddle table and two N x CG complex
// buffers; the second doubles as the (L x CG) real staging image of the packed spectrum, so that every HBM access
// is a flat float4 copy of the slab (VEC: CG == C, C % 4 == 0) and half the LDS / flops of the full complex
// transform are spent.  The Stockham passes keep one whole radix-8 / 4 / 2 butterfly per thread in registers
// (R reads, R-1 twiddles, R writes per butterfly); the channel index is the fastest thread index, rounded up to a
// power of two so that every index split is a shift.
// ---------------------------------------------------------------------------
struct Pow2Plan {
  int npass;
  int radix[12];
};

// radices of the N-point complex transform, N = 2^e; radix 8 last (the last pass has unit twiddles)
constexpr __host__ __device__ Pow2Plan pow2_plan(int N) {
  Pow2Plan p{};
  int e = 0;
  while ((1 << e) < N) ++e;
  const int r = e % 3;
  if (r == 1 && e >= 4) {
    p.radix[p.npass++] = 4;
    p.radix[p.npass++] = 4;
    e -= 4;
  } else if (r == 1) {
    p.radix[p.npass++] = 2;
    e -= 1;
  } else if (r == 2) {
    p.radix[p.npass++] = 4;
    e -= 2;
  }
  for (; e > 0; e -= 3) p.radix[p.npass++] = 8;
  return p;
}
constexpr __host__ __device__ int ilog2c(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

template <bool INV>
__device__ __forceinline__ float2 mul_mi(float2 a) {  // a * (-i) forward, a * (+i) inverse
  return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

template <bool INV>
__device__ __forceinline__ void dft4(float2& v0, float2& v1, float2& v2, float2& v3) {
  const float2 s0 = cadd(v0, v2), d0 = csub(v0, v2), s1 = cadd(v1, v3), d1 = mul_mi<INV>(csub(v1, v3));
  v0 = cadd(s0, s1), v2 = csub(s0, s1), v1 = cadd(d0, d1), v3 = csub(d0, d1);
}

template <int R, bool INV>
__device__ __forceinline__ void dft_r(float2 (&v)[R]) {
  if constexpr (R == 2) {
    const float2 t = csub(v[0], v[1]);
    v[0] = cadd(v[0], v[1]), v[1] = t;
  } else if constexpr (R == 4) {
    dft4<INV>(v[0], v[1], v[2], v[3]);
  } else {
    static_assert(R == 8, "radix 2, 4 or 8");
    dft4<INV>(v[0], v[2], v[4], v[6]);  // even inputs -> E[0..3] in v[0], v[2], v[4], v[6]
    dft4<INV>(v[1], v[3], v[5], v[7]);  // odd inputs  -> O[0..3] in v[1], v[3], v[5], v[7]
    constexpr float h = 0.70710678118654752440f;
    // w8^k O[k]:  w8 = exp(-+ i pi / 4)
    const float2 o0 = v[1];
    const float2 o1 = INV ? make_float2(h * (v[3].x - v[3].y), h * (v[3].x + v[3].y))
                          : make_float2(h * (v[3].x + v[3].y), h * (v[3].y - v[3].x));
    const float2 o2 = mul_mi<INV>(v[5]);
    const float2 o3 = INV ? make_float2(-h * (v[7].x + v[7].y), h * (v[7].x - v[7].y))
                          : make_float2(h * (v[7].y - v[7].x), -h * (v[7].x + v[7].y));
    const float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    v[0] = cadd(e0, o0), v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1), v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2), v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3), v[7] = csub(e3, o3);
  }
}

// One Stockham pass of radix R over the N-point transforms of cg channels: sub-length n = R m, stride s (n s = N).
//   y[q + s (R p + k)] = w_n^{p k} sum_j x[q + s (p + j m)] w_R^{j k},   w_n^{pk} = T[2 p k s]   (T: length L = 2N)
template <int R, bool INV>
__device__ __forceinline__ void pow2_pass(const float2* __restrict__ x, float2* __restrict__ y,
                                          const float2* __restrict__ T, int N, int ls, int lm, int CG, int cg, int lc) {
  const int nb = (N / R) << lc;
  const int s = 1 << ls, m = 1 << lm;
  for (int t = threadIdx.x; t < nb; t += blockDim.x) {
    const int c = t & ((1 << lc) - 1);
    if (c >= cg) continue;
    const int jb = t >> lc;
    const int q = jb & (s - 1), p = jb >> ls;
    float2 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = x[(q + s * (p + j * m)) * CG + c];
    dft_r<R, INV>(v);
    if (m > 1) {  // the last pass (m == 1, p == 0) has unit twiddles
      const int e = p << (ls + 1);
#pragma unroll
      for (int k = 1; k < R; ++k) {
        float2 w = T[e * k];
        if (INV) w.y = -w.y;
        v[k] = cmul(v[k], w);
      }
    }
    float2* yo = y + (q + s * R * p) * CG + c;
#pragma unroll
    for (int k = 0; k < R; ++k) yo[s * k * CG] = v[k];
  }
  __syncthreads();
}

template <bool INV>
__device__ __forceinline__ float2* pow2_fft(float2* x, float2* y, const float2* T, const Pow2Plan& plan, int N, int CG,
                                            int cg, int lc) {
  int ls = 0, lm = 31 - __builtin_clz(N);
  for (int ps = 0; ps < plan.npass; ++ps) {
    const int R = plan.radix[ps];
    const int lr = R == 8 ? 3 : (R == 4 ? 2 : 1);
    lm -= lr;
    if (R == 8) pow2_pass<8, INV>(x, y, T, N, ls, lm, CG, cg, lc);
    else if (R == 4) pow2_pass<4, INV>(x, y, T, N, ls, lm, CG, cg, lc);
    else pow2_pass<2, INV>(x, y, T, N, ls, lm, CG, cg, lc);
    ls += lr;
    float2* tmp = x;
    x = y;
    y = tmp;
  }
  return x;
}

// The same transform with N (and so the radix plan and every stride) a compile-time constant: the passes unroll.
template <bool INV, int N, int LS, int LM>
__device__ __forceinline__ float2* pow2_fft_static(float2* x, float2* y, const float2* T, int CG, int cg, int lc) {
  if constexpr (LM == 0) {
    return x;
  } else {
    constexpr int LR = (LM % 3 == 0) ? 3 : (LM % 3 == 2) ? 2 : (LM >= 4 ? 2 : 1);  // pow2_plan's order
    pow2_pass<(1 << LR), INV>(x, y, T, N, LS, LM - LR, CG, cg, lc);
    return pow2_fft_static<INV, N, LS + LR, LM - LR>(y, x, T, CG, cg, lc);
  }
}

template <bool INV, int LT>
__device__ __forceinline__ float2* pow2_fft_any(float2* x, float2* y, const float2* T, const Pow2Plan& plan, int N, int CG,
                                                int cg, int lc) {
  if constexpr (LT != 0) return pow2_fft_static<INV, LT / 2, 0, ilog2c(LT / 2)>(x, y, T, CG, cg, lc);
  else return pow2_fft<INV>(x, y, T, plan, N, CG, cg, lc);
}

// Z (N x CG complex) -> packed ortho spectrum rows (L x CG floats) in `st`, scaled by sc.
// Work items per channel: item 0 = the two self-paired bins k = 0 and k = N/2, item k (0 < k < N/2) = the pair (k, N-k):
// max(1, N/2) items, a whole number of passes over the workgroup for the power-of-two shapes.
__device__ __forceinline__ void pow2_split_fwd(const float2* __restrict__ Z, float* __restrict__ st,
                                               const float2* __restrict__ T, int N, int CG, int cg, int lc, float sc) {
  const int h = N >> 1;
  const int nk = (h > 0 ? h : 1) << lc;
  for (int t = threadIdx.x; t < nk; t += blockDim.x) {
    const int c = t & ((1 << lc) - 1);
    if (c >= cg) continue;
    const int k = t >> lc;
    if (k == 0) {
      const float2 a = Z[c];
      st[c] = (a.x + a.y) * sc;
      st[N * CG + c] = (a.x - a.y) * sc;
      if (h > 0) {  // X[N/2] = conj(Z[N/2])
        const float2 m = Z[h * CG + c];
        st[h * CG + c] = m.x * sc;
        st[(N + h) * CG + c] = -m.y * sc;
      }
    } else {
      const float2 A = Z[k * CG + c], B = Z[(N - k) * CG + c];
      const float2 E = make_float2(0.5f * (A.x + B.x), 0.5f * (A.y - B.y));
      const float2 D = make_float2(0.5f * (A.y + B.y), -0.5f * (A.x - B.x));  // (A - conj B) / 2i
      const float2 P = cmul(T[k], D);
      st[k * CG + c] = (E.x + P.x) * sc;
      st[(N + k) * CG + c] = (E.y + P.y) * sc;
      st[(N - k) * CG + c] = (E.x - P.x) * sc;
      st[(2 * N - k) * CG + c] = -(E.y - P.y) * sc;
    }
  }
  __syncthreads();
}

// packed spectrum rows (L x CG floats) in `st` -> Z (N x CG complex), unnormalised
__device__ __forceinline__ void pow2_split_inv(const float* __restrict__ st, float2* __restrict__ Z,
                                               const float2* __restrict__ T, int N, int CG, int cg, int lc) {
  const int h = N >> 1;
  const int nk = (h > 0 ? h : 1) << lc;
  for (int t = threadIdx.x; t < nk; t += blockDim.x) {
    const int c = t & ((1 << lc) - 1);
    if (c >= cg) continue;
    const int k = t >> lc;
    if (k == 0) {
      const float r0 = st[c], rn = st[N * CG + c];
      Z[c] = make_float2(r0 + rn, r0 - rn);
      if (h > 0) Z[h * CG + c] = make_float2(2.f * st[h * CG + c], -2.f * st[(N + h) * CG + c]);  // 2 conj(X[N/2])
    } else {
      const float2 X = make_float2(st[k * CG + c], st[(N + k) * CG + c]);
      const float2 Y = make_float2(st[(N - k) * CG + c], st[(2 * N - k) * CG + c]);  // X[N-k]
      const float2 S = make_float2(X.x + Y.x, X.y - Y.y);    // X[k] + conj X[N-k]
      const float2 Dd = make_float2(X.x - Y.x, X.y + Y.y);   // X[k] - conj X[N-k]
      float2 w = T[k];
      w.y = -w.y;
      const float2 Q = cmul(w, Dd);
      Z[k * CG + c] = make_float2(S.x - Q.y, S.y + Q.x);           // S + iQ
      Z[(N - k) * CG + c] = make_float2(S.x + Q.y, -(S.y - Q.x));  // conj(S - iQ)
    }
  }
  __syncthreads();
}

// Slab I/O of the VEC path (CG == C, C % 4 == 0: the (L x C) slab is a flat array of n4 = L C / 4 float4).  The
// kernels are persistent over samples and keep the NEXT sample's slab in registers (PF float4 per thread) while the
// current one is transformed in LDS, so a workgroup always has a slab of HBM loads in flight.
constexpr int PF = 8;  // float4 prefetch registers per thread: slabs up to 8 * 256 float4 = 32 KiB are prefetched

__device__ __forceinline__ void slab_prefetch(const float* __restrict__ src, int n4, float4 (&pre)[PF]) {
  const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int f = threadIdx.x + i * 256;
    if (f < n4) pre[i] = s4[f];
  }
}

// time-domain slab -> z image (even rows real parts, odd rows imaginary parts): z[n][c] = (x[2n][c], x[2n+1][c])
// row of float4 slot f in a slab with C4 float4 per row (C4 a power of two in every shipped shape: no division)
__device__ __forceinline__ int slab_row(int f, int C4) { return (C4 & (C4 - 1)) == 0 ? f >> (31 - __builtin_clz(C4)) : f / C4; }

__device__ __forceinline__ void z_put(float* __restrict__ zf, int f, float4 v, int C4, int CG) {
  const int l = slab_row(f, C4), cq = (f - l * C4) << 2;
  float* d = zf + (((l >> 1) * CG + cq) << 1) + (l & 1);
  d[0] = v.x, d[2] = v.y, d[4] = v.z, d[6] = v.w;
}

template <bool VEC>
__device__ __forceinline__ void pow2_load_time(const float* __restrict__ src, float2* __restrict__ z, int L, int C,
                                               int c0, int CG, int cg, bool prefetched, const float4 (&pre)[PF]) {
  float* zf = reinterpret_cast<float*>(z);
  if (VEC) {
    const int C4 = C >> 2, n4 = L * C4;
    if (prefetched) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < n4) z_put(zf, f, pre[i], C4, CG);
      }
    } else {
      const float4* s4 = reinterpret_cast<const float4*>(src);
      for (int f = threadIdx.x; f < n4; f += blockDim.x) z_put(zf, f, s4[f], C4, CG);
    }
  } else {
    for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
      const int l = idx / cg, cc = idx - l * cg;
      zf[(((l >> 1) * CG + cc) << 1) + (l & 1)] = src[(size_t)l * C + c0 + cc];
    }
  }
  __syncthreads();
}

// z image -> slab (L x C), scaled
template <bool VEC>
__device__ __forceinline__ void pow2_store_time(const float2* __restrict__ z, float* __restrict__ dst, int L, int C,
                                                int c0, int CG, int cg, float sc) {
  const float* zf = reinterpret_cast<const float*>(z);
  if (VEC) {
    const int C4 = C >> 2, n4 = L * C4;
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int f = threadIdx.x; f < n4; f += blockDim.x) {
      const int l = slab_row(f, C4), cq = (f - l * C4) << 2;
      const float* p = zf + (((l >> 1) * CG + cq) << 1) + (l & 1);
      d4[f] = make_float4(p[0] * sc, p[2] * sc, p[4] * sc, p[6] * sc);
    }
  } else {
    for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
      const int l = idx / cg, cc = idx - l * cg;
      dst[(size_t)l * C + c0 + cc] = zf[(((l >> 1) * CG + cc) << 1) + (l & 1)] * sc;
    }
  }
}

// dft (INVERSE = false) / idft (true), persistent over the samples b = blockIdx.x, += gridDim.x, with the optional
// affine wrappers:
//   forward : out = (dft(x) - a0) / a1                    (datamodules.py:42-43,61-62; a0 = mean, a1 = std, (L, C))
//   inverse : out = idft(x * a0 + a1)                     (cmd/sample.py:107-113;     a0 = std,  a1 = mean)
// each product / sum rounded separately like the reference's tensor ops.
// LT / CT != 0: the slab shape (L = LT, C = CG = CT) is a compile-time constant -- every index split is a shift by a
// literal, the radix plan unrolls, the trip counts are known.  The generic instance (0, 0) spends most of its issue
// slots on integer address arithmetic (64-bit multiplies by runtime strides): 65 us vs the config-5 shard's 45 us
// of HBM time.  (512, 8) is the BASELINE configs[4] shape.
#define FFD_POW2_SHAPE                                            \
  const int L = LT ? LT : L_;                                     \
  const int C = CT ? CT : C_;                                     \
  const int CG = CT ? CT : CG_;                                   \
  const int lc = CT ? ilog2c(CT) : lc_;                           \
  const Pow2Plan plan = LT ? pow2_plan(LT / 2) : plan_;

template <bool INVERSE, bool VEC, int LT, int CT>
__global__ __launch_bounds__(256) void k_rfft_pow2(const float* __restrict__ in, float* __restrict__ out,
                                                   const float2* __restrict__ Wg, Pow2Plan plan_, int B, int L_, int C_,
                                                   int CG_, int lc_, float scale, const float* __restrict__ a0,
                                                   const float* __restrict__ a1) {
  FFD_POW2_SHAPE
  extern __shared__ __align__(16) float2 sm[];
  const int N = L >> 1;
  float2* T = sm;                 // L twiddles
  float2* bufA = sm + L;          // N * CG complex
  float2* bufB = bufA + N * CG;   // N * CG complex == L * CG floats
  const int c0 = blockIdx.y * CG;
  const int cg = min(CG, C - c0);
  const int n4 = (L * C) >> 2;
  const bool pf = VEC && n4 <= PF * 256;
  for (int i = threadIdx.x; i < L; i += blockDim.x) T[i] = Wg[i];
  float4 pre[PF];
  if (pf && (int)blockIdx.x < B) slab_prefetch(in + (size_t)blockIdx.x * L * C, n4, pre);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float* src = in + (size_t)b * L * C;
    float* dst = out + (size_t)b * L * C;
    const int bn = b + gridDim.x;
    if (!INVERSE) {
      pow2_load_time<VEC>(src, bufA, L, C, c0, CG, cg, pf, pre);  // (barrier inside; also covers the twiddle table)
      if (pf && bn < B) slab_prefetch(in + (size_t)bn * L * C, n4, pre);
      float2* Z = pow2_fft_any<false, LT>(bufA, bufB, T, plan, N, CG, cg, lc);
      float* st = reinterpret_cast<float*>(Z == bufA ? bufB : bufA);
      pow2_split_fwd(Z, st, T, N, CG, cg, lc, scale);
      if (VEC) {
        const float4* s4 = reinterpret_cast<const float4*>(st);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int f = threadIdx.x; f < n4; f += blockDim.x) {
          float4 v = s4[f];
          if (a0) {
            const float4 mu = reinterpret_cast<const float4*>(a0)[f], sd = reinterpret_cast<const float4*>(a1)[f];
            v = make_float4(__fdiv_rn(__fsub_rn(v.x, mu.x), sd.x), __fdiv_rn(__fsub_rn(v.y, mu.y), sd.y),
                            __fdiv_rn(__fsub_rn(v.z, mu.z), sd.z), __fdiv_rn(__fsub_rn(v.w, mu.w), sd.w));
          }
          d4[f] = v;
        }
      } else {
        for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
          const int o = idx / cg, cc = idx - o * cg;
          float v = st[o * CG + cc];
          const size_t gi = (size_t)o * C + c0 + cc;
          if (a0) v = __fdiv_rn(__fsub_rn(v, a0[gi]), a1[gi]);
          dst[gi] = v;
        }
      }
    } else {
      float* st = reinterpret_cast<float*>(bufB);
      if (VEC) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        auto put = [&](int f, float4 v) {
          if (a0) {
            const float4 sd = reinterpret_cast<const float4*>(a0)[f], mu = reinterpret_cast<const float4*>(a1)[f];
            v = make_float4(__fadd_rn(__fmul_rn(v.x, sd.x), mu.x), __fadd_rn(__fmul_rn(v.y, sd.y), mu.y),
                            __fadd_rn(__fmul_rn(v.z, sd.z), mu.z), __fadd_rn(__fmul_rn(v.w, sd.w), mu.w));
          }
          reinterpret_cast<float4*>(st)[f] = v;
        };
        if (pf) {
#pragma unroll
          for (int i = 0; i < PF; ++i) {
            const int f = threadIdx.x + i * 256;
            if (f < n4) put(f, pre[i]);
          }
        } else {
          for (int f = threadIdx.x; f < n4; f += blockDim.x) put(f, s4[f]);
        }
      } else {
        for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
          const int o = idx / cg, cc = idx - o * cg;
          const size_t gi = (size_t)o * C + c0 + cc;
          float v = src[gi];
          if (a0) v = __fadd_rn(__fmul_rn(v, a0[gi]), a1[gi]);
          st[o * CG + cc] = v;
        }
      }
      __syncthreads();
      if (pf && bn < B) slab_prefetch(in + (size_t)bn * L * C, n4, pre);
      pow2_split_inv(st, bufA, T, N, CG, cg, lc);
      float2* z = pow2_fft_any<true, LT>(bufA, bufB, T, plan, N, CG, cg, lc);
      pow2_store_time<VEC>(z, dst, L, C, c0, CG, cg, scale);
    }
    __syncthreads();  // the LDS images are rewritten by the next sample
  }
}

// FreSca on the power-of-two path: per-sample |X_k| partial sums, and FFT -> per-bin scale -> inverse FFT.
template <bool VEC, int LT, int CT>
__global__ __launch_bounds__(256) void k_fresca_spectrum_pow2(const float* __restrict__ in, float* __restrict__ partial,
                                                              const float2* __restrict__ Wg, Pow2Plan plan_, int B, int L_,
                                                              int C_, int CG_, int lc_, float scale) {
  FFD_POW2_SHAPE
  extern __shared__ __align__(16) float2 sm[];
  const int N = L >> 1;
  float2* T = sm;
  float2* bufA = sm + L;
  float2* bufB = bufA + N * CG;
  const int c0 = blockIdx.y * CG, cg = min(CG, C - c0);
  const int n4 = (L * C) >> 2;
  const bool pf = VEC && n4 <= PF * 256;
  for (int i = threadIdx.x; i < L; i += blockDim.x) T[i] = Wg[i];
  float4 pre[PF];
  if (pf && (int)blockIdx.x < B) slab_prefetch(in + (size_t)blockIdx.x * L * C, n4, pre);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    pow2_load_time<VEC>(in + (size_t)b * L * C, bufA, L, C, c0, CG, cg, pf, pre);
    if (pf && b + (int)gridDim.x < B) slab_prefetch(in + (size_t)(b + gridDim.x) * L * C, n4, pre);
    float2* Z = pow2_fft_any<false, LT>(bufA, bufB, T, plan, N, CG, cg, lc);
    float* st = reinterpret_cast<float*>(Z == bufA ? bufB : bufA);
    pow2_split_fwd(Z, st, T, N, CG, cg, lc, scale);
    const int nf = N + 1;
    for (int k = threadIdx.x; k < nf; k += blockDim.x) {
      const bool has_im = k >= 1 && k < N;
      float acc = 0.f;
      for (int c = 0; c < cg; ++c) {
        const float re = st[k * CG + c], im = has_im ? st[(N + k) * CG + c] : 0.f;
        acc += sqrtf(fmaf(re, re, im * im));
      }
      partial[((size_t)b * gridDim.y + blockIdx.y) * nf + k] = acc;
    }
    __syncthreads();
  }
}

template <bool VEC, int LT, int CT>
__global__ __launch_bounds__(256) void k_fresca_apply_pow2(const float* __restrict__ in, float* __restrict__ out,
                                                           const float2* __restrict__ Wg, Pow2Plan plan_, int B, int L_,
                                                           int C_, int CG_, int lc_, const int* __restrict__ rc_dev,
                                                           float rc_host, float low, float high, float scale) {
  FFD_POW2_SHAPE
  extern __shared__ __align__(16) float2 sm[];
  const int N = L >> 1;
  float2* T = sm;
  float2* bufA = sm + L;
  float2* bufB = bufA + N * CG;
  const int c0 = blockIdx.y * CG, cg = min(CG, C - c0);
  const int n4 = (L * C) >> 2;
  const bool pf = VEC && n4 <= PF * 256;
  for (int i = threadIdx.x; i < L; i += blockDim.x) T[i] = Wg[i];
  // low-pass set: k <= Rc (energy: integer index from the cutoff kernel; spatial: Rc = r0 * n_freq)
  const float rc = rc_dev ? (float)(*rc_dev) : rc_host;
  float4 pre[PF];
  if (pf && (int)blockIdx.x < B) slab_prefetch(in + (size_t)blockIdx.x * L * C, n4, pre);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    pow2_load_time<VEC>(in + (size_t)b * L * C, bufA, L, C, c0, CG, cg, pf, pre);
    if (pf && b + (int)gridDim.x < B) slab_prefetch(in + (size_t)(b + gridDim.x) * L * C, n4, pre);
    float2* Z = pow2_fft_any<false, LT>(bufA, bufB, T, plan, N, CG, cg, lc);
    float2* other = (Z == bufA) ? bufB : bufA;
    float* st = reinterpret_cast<float*>(other);
    pow2_split_fwd(Z, st, T, N, CG, cg, lc, scale);
    if (VEC) {  // rows are whole float4 groups of one bin
      const int C4 = C >> 2;
      for (int f = threadIdx.x; f < n4; f += blockDim.x) {
        const int r = slab_row(f, C4);
        const int k = r <= N ? r : r - N;  // rows N+1.. hold Im X_k, k = r - N
        const float fac = ((float)k <= rc) ? low : high;
        float4 v = reinterpret_cast<float4*>(st)[f];
        reinterpret_cast<float4*>(st)[f] = make_float4(v.x * fac, v.y * fac, v.z * fac, v.w * fac);
      }
    } else {
      for (int idx = threadIdx.x; idx < L * cg; idx += blockDim.x) {
        const int r = idx / cg, cc = idx - r * cg;
        const int k = r <= N ? r : r - N;
        st[r * CG + cc] *= ((float)k <= rc) ? low : high;
      }
    }
    __syncthreads();
    pow2_split_inv(st, Z, T, N, CG, cg, lc);
    float2* z = pow2_fft_any<true, LT>(Z, other, T, plan, N, CG, cg, lc);
    pow2_store_time<VEC>(z, out + (size_t)b * L * C, L, C, c0, CG, cg, scale);
    __syncthreads();
  }
}

static bool is_pow2(int L) { return L >= 2 && (L & (L - 1)) == 0; }

static Pow2Plan make_pow2_plan(int N) { return pow2_plan(N); }

// channels per workgroup and LDS bytes of the power-of-two path: the whole slab when it fits
// (<= 64 KiB keeps several workgroups on a CU; up to the CU's 160 KiB before the channels are split)
static size_t pow2_lds(int L, int CG) { return (size_t)(L + (size_t)L * CG) * sizeof(float2); }
static int pow2_cg(int L, int C) {
  const size_t cap = 160 * 1024;
  if (pow2_lds(L, C) <= cap) return C;
  int CG = C;
  while (CG > 1 && pow2_lds(L, CG) > 64 * 1024) CG = (CG + 1) / 2;
  return CG;
}
static int ceil_log2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
// persistent grid: as many workgroups as the LDS image lets a CU hold (at most 8), times the CU count
static int persistent_blocks(int B, size_t lds) {
  int per_cu = (int)((160 * 1024) / lds);
  per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
  const int cap = 256 * per_cu;
  return B < cap ? B : cap;
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename K>
static hipError_t allow_lds(K kernel, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

static FftPlan make_plan(int L);

static FftPlan make_plan(int L) {
  FftPlan p{};
  int n = L;
  auto push = [&](int r) { p.radix[p.npass++] = r; };
  while (n % 8 == 0) push(8), n /= 8;
  while (n % 4 == 0) push(4), n /= 4;
  while (n % 2 == 0) push(2), n /= 2;
  for (int f = 3; (long)f * f <= n; f += 2)
    while (n % f == 0) push(f), n /= f;
  if (n > 1) push(n);
  if (L == 1) push(1);
  return p;
}

namespace {
std::mutex g_tw_mutex;
std::map<std::pair<int, int>, float2*> g_twiddles;  // (device, L) -> device table
}  // namespace

static hipError_t get_twiddles(int L, const float2** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(g_tw_mutex);
  auto key = std::make_pair(dev, L);
  auto it = g_twiddles.find(key);
  if (it != g_twiddles.end()) {
    *out = it->second;
    return hipSuccess;
  }
  std::vector<float2> h(L);
  for (int i = 0; i < L; ++i) {
    double a = -2.0 * M_PI * (double)i / (double)L;
    h[i] = make_float2((float)cos(a), (float)sin(a));
  }
  float2* d = nullptr;
  e = hipMalloc(&d, sizeof(float2) * L);
  if (e != hipSuccess) return e;
  e = hipMemcpy(d, h.data(), sizeof(float2) * L, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return e;
  }
  g_twiddles[key] = d;
  *out = d;
  return hipSuccess;
}

hipError_t launch_dft(const float* in, float* out, int B, int L, int C, int inverse, const float* a0, const float* a1,
                      hipStream_t s) {
  if (B <= 0) return hipSuccess;
  if (L < 1 || C < 1 || L > 8192) return hipErrorInvalidValue;
  if ((a0 == nullptr) != (a1 == nullptr)) return hipErrorInvalidValue;
  const float2* W = nullptr;
  hipError_t e = get_twiddles(L, &W);
  if (e != hipSuccess) return e;
  const float scale = (float)(1.0 / sqrt((double)L));
  if (is_pow2(L)) {
    const Pow2Plan plan = make_pow2_plan(L / 2);
    const int CG = pow2_cg(L, C);
    const size_t lds = pow2_lds(L, CG);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int lc = ceil_log2(CG);
    const bool vec = CG == C && C % 4 == 0 && aligned16(in) && aligned16(out) && aligned16(a0) && aligned16(a1);
    dim3 grid(persistent_blocks(B, lds), cdiv(C, CG)), block(256);
#define FFD_RFFT(INV, VEC, LT, CT)                                                                           \
  do {                                                                                                       \
    if ((e = allow_lds(k_rfft_pow2<INV, VEC, LT, CT>, lds)) != hipSuccess) return e;                         \
    hipLaunchKernelGGL((k_rfft_pow2<INV, VEC, LT, CT>), grid, block, lds, s, in, out, W, plan, B, L, C, CG, lc, scale, a0, \
                       a1);                                                                                  \
  } while (0)
    const bool shard = vec && L == 512 && C == 8;  // the config-5 shape has its own instance
    if (inverse) {
      if (shard) FFD_RFFT(true, true, 512, 8); else if (vec) FFD_RFFT(true, true, 0, 0); else FFD_RFFT(true, false, 0, 0);
    } else {
      if (shard) FFD_RFFT(false, true, 512, 8); else if (vec) FFD_RFFT(false, true, 0, 0); else FFD_RFFT(false, false, 0, 0);
    }
#undef FFD_RFFT
    return hipGetLastError();
  }
  FftPlan plan = make_plan(L);
  if (plan.npass > 16) return hipErrorInvalidValue;
  // channels per workgroup so that twiddles + two slabs fit in 64 KiB of LDS
  int CG = C;
  while (CG > 1 && (size_t)(L + 2 * (size_t)L * CG) * sizeof(float2) > 64 * 1024) CG = (CG + 1) / 2;
  size_t lds = (size_t)(L + 2 * (size_t)L * CG) * sizeof(float2);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  dim3 grid(B, cdiv(C, CG)), block(256);
  if (inverse) {
    if ((e = allow_lds(k_fft<true>, lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_fft<true>, grid, block, lds, s, in, out, W, plan, L, C, CG, scale, a0, a1);
  } else {
    if ((e = allow_lds(k_fft<false>, lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_fft<false>, grid, block, lds, s, in, out, W, plan, L, C, CG, scale, a0, a1);
  }
  return hipGetLastError();
}

// geometry shared by the FreSca / decomposition launchers
struct SlabGeom {
  bool pow2;
  int CG, lc;
  size_t lds;
  FftPlan plan;
  Pow2Plan plan2;
};
static hipError_t slab_geom(int L, int C, SlabGeom* g) {
  g->pow2 = is_pow2(L);
  if (g->pow2) {
    g->plan2 = make_pow2_plan(L / 2);
    g->CG = pow2_cg(L, C);
    g->lds = pow2_lds(L, g->CG);
    g->lc = ceil_log2(g->CG);
  } else {
    g->plan = make_plan(L);
    int CG = C;
    while (CG > 1 && (size_t)(L + 2 * (size_t)L * CG) * sizeof(float2) > 64 * 1024) CG = (CG + 1) / 2;
    g->CG = CG;
    g->lds = (size_t)(L + 2 * (size_t)L * CG) * sizeof(float2);
    g->lc = 0;
  }
  return g->lds > 160 * 1024 ? hipErrorInvalidValue : hipSuccess;
}

static hipError_t launch_fresca_apply(const float* in, float* out, const float2* W, const SlabGeom& g, int B, int L, int C,
                                      const int* rc_dev, float rc_host, float low, float high, float sc, hipStream_t s) {
  dim3 grid(B, cdiv(C, g.CG)), block(256);
  hipError_t e;
  if (g.pow2) {
    const bool vec = g.CG == C && C % 4 == 0 && aligned16(in) && aligned16(out);
    grid.x = persistent_blocks(B, g.lds);
    if (vec && L == 512 && C == 8) {
      if ((e = allow_lds(k_fresca_apply_pow2<true, 512, 8>, g.lds)) != hipSuccess) return e;
      hipLaunchKernelGGL((k_fresca_apply_pow2<true, 512, 8>), grid, block, g.lds, s, in, out, W, g.plan2, B, L, C, g.CG,
                         g.lc, rc_dev, rc_host, low, high, sc);
    } else if (vec) {
      if ((e = allow_lds(k_fresca_apply_pow2<true, 0, 0>, g.lds)) != hipSuccess) return e;
      hipLaunchKernelGGL((k_fresca_apply_pow2<true, 0, 0>), grid, block, g.lds, s, in, out, W, g.plan2, B, L, C, g.CG, g.lc,
                         rc_dev, rc_host, low, high, sc);
    } else {
      if ((e = allow_lds(k_fresca_apply_pow2<false, 0, 0>, g.lds)) != hipSuccess) return e;
      hipLaunchKernelGGL((k_fresca_apply_pow2<false, 0, 0>), grid, block, g.lds, s, in, out, W, g.plan2, B, L, C, g.CG, g.lc,
                         rc_dev, rc_host, low, high, sc);
    }
  } else {
    if ((e = allow_lds(k_fresca_apply, g.lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_fresca_apply, grid, block, g.lds, s, in, out, W, g.plan, L, C, g.CG, rc_dev, rc_host, low, high,
                       sc * sc);
  }
  return hipGetLastError();
}

hipError_t launch_fresca(const float* in, float* out, float* work, int B, int L, int C, float low, float high,
                         double cutoff_ratio, int strategy, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  if (L < 2 || C < 1 || L > 4096) return hipErrorInvalidValue;
  const float2* W = nullptr;
  hipError_t e = get_twiddles(L, &W);
  if (e != hipSuccess) return e;
  SlabGeom g;
  if ((e = slab_geom(L, C, &g)) != hipSuccess) return e;
  const int NGc = cdiv(C, g.CG);
  const int nf = L / 2 + 1;
  const float sc = (float)(1.0 / sqrt((double)L));
  const int* rc_dev = nullptr;
  float rc_host = 0.f;
  if (strategy == 1) {  // energy (fresca.py:46-58)
    float* partial = work;
    int* rc = reinterpret_cast<int*>(work + (size_t)B * NGc * nf);
    dim3 grid(B, NGc), block(256);
    if (g.pow2) {
      grid.x = persistent_blocks(B, g.lds);
      const bool vec = g.CG == C && C % 4 == 0 && aligned16(in);
      if (vec && L == 512 && C == 8) {
        if ((e = allow_lds(k_fresca_spectrum_pow2<true, 512, 8>, g.lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_fresca_spectrum_pow2<true, 512, 8>), grid, block, g.lds, s, in, partial, W, g.plan2, B, L, C,
                           g.CG, g.lc, sc);
      } else if (vec) {
        if ((e = allow_lds(k_fresca_spectrum_pow2<true, 0, 0>, g.lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_fresca_spectrum_pow2<true, 0, 0>), grid, block, g.lds, s, in, partial, W, g.plan2, B, L, C, g.CG,
                           g.lc, sc);
      } else {
        if ((e = allow_lds(k_fresca_spectrum_pow2<false, 0, 0>, g.lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_fresca_spectrum_pow2<false, 0, 0>), grid, block, g.lds, s, in, partial, W, g.plan2, B, L, C, g.CG,
                           g.lc, sc);
      }
    } else {
      if ((e = allow_lds(k_fresca_spectrum, g.lds)) != hipSuccess) return e;
      hipLaunchKernelGGL(k_fresca_spectrum, grid, block, g.lds, s, in, partial, W, g.plan, L, C, g.CG, sc);
    }
    hipLaunchKernelGGL(k_fresca_cutoff, dim3(1), dim3(256), nf * sizeof(float), s, partial, rc, B, B * NGc, nf, C,
                       cutoff_ratio);
    rc_dev = rc;
  } else {  // spatial (fresca.py:40-43): Rc = r0 * n_freq, compared in fp32 like the reference's k tensor
    rc_host = (float)(cutoff_ratio * (double)nf);
  }
  return launch_fresca_apply(in, out, W, g, B, L, C, rc_dev, rc_host, low, high, sc, s);
}

// frequency_decompose_fft (fourier.py:219-286): rfft along L, keep bins k < n_low for the low part and
// k >= n_low for the high part, irfft each.  Same LDS-resident FFT -> mask -> inverse FFT kernel as FreSca
// with factors (1,0) and (0,1); the tensor is (B, L, D) with D innermost, exactly the (B, L, C) layout.
hipError_t launch_freq_decompose(const float* in, float* low, float* high, int B, int L, int D, double low_freq_ratio,
                                 hipStream_t s) {
  if (B <= 0) return hipSuccess;
  if (L < 2 || D < 1 || L > 4096) return hipErrorInvalidValue;
  const float2* W = nullptr;
  hipError_t e = get_twiddles(L, &W);
  if (e != hipSuccess) return e;
  SlabGeom g;
  if ((e = slab_geom(L, D, &g)) != hipSuccess) return e;
  const int nf = L / 2 + 1;
  int n_low = (int)((double)nf * low_freq_ratio);  // fourier.py:249  max(1, int(n_freq * ratio))
  if (n_low < 1) n_low = 1;
  const float rc = (float)n_low - 0.5f;             // bins k <= rc  <=>  k < n_low
  const float sc = (float)(1.0 / sqrt((double)L));
  if ((e = launch_fresca_apply(in, low, W, g, B, L, D, nullptr, rc, 1.f, 0.f, sc, s)) != hipSuccess) return e;
  return launch_fresca_apply(in, high, W, g, B, L, D, nullptr, rc, 0.f, 1.f, sc, s);
}

// spectral_density on a packed spectrum (fourier.py:111-130): |X_k|^2 for k = 0..L/2; the imaginary part of
// bin 0 (and of the Nyquist bin when L is even) is the implicit zero of the packed layout.
__global__ void k_spectral_density(const float* __restrict__ xf, float* __restrict__ out, int B, int L, int C) {
  const int nr = L / 2 + 1;
  const size_t n = (size_t)B * nr * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t bk = i / C;
    const int k = (int)(bk % nr);
    const size_t b = bk / nr;
    const float re = xf[(b * L + k) * C + c];
    const bool has_im = k >= 1 && (nr + k - 1) < L;
    const float im = has_im ? xf[(b * L + nr + k - 1) * C + c] : 0.f;
    out[i] = re * re + im * im;
  }
}

hipError_t launch_spectral_density(const float* xf, float* out, int B, int L, int C, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  const size_t n = (size_t)B * (L / 2 + 1) * C;
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(k_spectral_density, dim3(blocks), dim3(256), 0, s, xf, out, B, L, C);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// FreSca, 4-D branch (fresca.py:184-213): x (B, H, W, C) -> irfft2( (l*low + h*high) (.) rfft2(x) ) over (H, W), ortho,
// low = [sqrt(kh^2 + kw^2) <= Rc] on the raw bin indices (fresca.py:73-81).  Not on the sampling path (scores are 3-D):
// a cold path, built for the completeness of fdiff.utils.fresca -- separable direct DFT sums in LDS, one (sample,
// channel) image per workgroup, fp64 accumulation with fp32 intermediates, H, W <= 256 and H W <= 4096.
//   k_fresca2d_fwd : X = rfft2(x) -> spec (complex), |X| -> mag            (per image)
//   k_fresca2d_cut : energy strategy: spec_mean = mean_{b,c} |X|; Rc = first R in 0..int(min(H, nW) / 2) with
//                    sum(spec_mean [dist <= R]) >= r0 * sum(spec_mean) (fresca.py:89-101), else 0; fixed summation order
//   k_fresca2d_inv : scale, complex inverse DFT along H, c2r along W (imaginary parts of the kw = 0 and Nyquist columns
//                    are ignored exactly as a c2r transform ignores them)
// ---------------------------------------------------------------------------
constexpr int F2D_MAX_HW = 4096, F2D_MAX_DIM = 256, F2D_MAX_SPEC = F2D_MAX_HW / 2 + F2D_MAX_DIM;

__device__ __forceinline__ void f2d_tables(double2* twH, double2* twW, int H, int W) {
  for (int r = threadIdx.x; r < H; r += blockDim.x) {
    double sn, cs;
    sincospi(2.0 * (double)r / (double)H, &sn, &cs);
    twH[r] = make_double2(cs, sn);
  }
  for (int r = threadIdx.x; r < W; r += blockDim.x) {
    double sn, cs;
    sincospi(2.0 * (double)r / (double)W, &sn, &cs);
    twW[r] = make_double2(cs, sn);
  }
}

__global__ __launch_bounds__(256) void k_fresca2d_fwd(const float* __restrict__ in, float2* __restrict__ spec,
                                                      float* __restrict__ mag, int H, int W, int C) {
  __shared__ float img[F2D_MAX_HW];
  __shared__ float2 T[F2D_MAX_SPEC];
  __shared__ double2 twH[F2D_MAX_DIM], twW[F2D_MAX_DIM];
  const int nW = W / 2 + 1;
  const int b = blockIdx.x / C, c = blockIdx.x - b * C;
  f2d_tables(twH, twW, H, W);
  for (int i = threadIdx.x; i < H * W; i += blockDim.x) img[i] = in[((size_t)b * H * W + i) * C + c];
  __syncthreads();
  const double sw = 1.0 / sqrt((double)W), sh = 1.0 / sqrt((double)H);
  for (int i = threadIdx.x; i < H * nW; i += blockDim.x) {  // rfft along W
    const int h = i / nW, kw = i - h * nW;
    double re = 0.0, im = 0.0;
    int r = 0;
    for (int w = 0; w < W; ++w) {
      const double v = (double)img[h * W + w];
      re += v * twW[r].x, im -= v * twW[r].y;
      r += kw;
      r -= r >= W ? W : 0;
    }
    T[i] = make_float2((float)(re * sw), (float)(im * sw));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H * nW; i += blockDim.x) {  // complex DFT along H
    const int kh = i / nW, kw = i - kh * nW;
    double re = 0.0, im = 0.0;
    int r = 0;
    for (int h = 0; h < H; ++h) {
      const double a = (double)T[h * nW + kw].x, bb = (double)T[h * nW + kw].y;
      re += a * twH[r].x + bb * twH[r].y, im += bb * twH[r].x - a * twH[r].y;
      r += kh;
      r -= r >= H ? H : 0;
    }
    const float2 X = make_float2((float)(re * sh), (float)(im * sh));
    spec[(size_t)blockIdx.x * H * nW + i] = X;
    mag[(size_t)blockIdx.x * H * nW + i] = sqrtf(fmaf(X.x, X.x, X.y * X.y));
  }
}

__global__ __launch_bounds__(256) void k_fresca2d_cut(const float* __restrict__ mag, float* __restrict__ rc_out, int nimg,
                                                      int H, int nW, float cutoff_ratio) {
  __shared__ float sm[F2D_MAX_SPEC];
  __shared__ float energy[F2D_MAX_DIM];
  __shared__ float etot_s;
  const int P = H * nW;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    float acc = 0.f;
    for (int i = 0; i < nimg; ++i) acc += mag[(size_t)i * P + p];
    sm[p] = acc / (float)nimg;  // mean over batch and channels (fresca.py:191)
  }
  __syncthreads();
  const int rmax = min(H, nW) / 2;  // range(int(min(H, W) / 2) + 1)
  if ((int)threadIdx.x <= rmax) {
    double e = 0.0;
    const float R = (float)threadIdx.x;
    for (int p = 0; p < P; ++p) {
      const int kh = p / nW, kw = p - kh * nW;
      if (sqrtf((float)(kh * kh + kw * kw)) <= R) e += (double)sm[p];
    }
    energy[threadIdx.x] = (float)e;
  }
  if (threadIdx.x == 255) {
    double e = 0.0;
    for (int p = 0; p < P; ++p) e += (double)sm[p];
    etot_s = (float)e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float thr = cutoff_ratio * etot_s;
    int rc = 0;
    for (int R = 0; R <= rmax; ++R)
      if (energy[R] >= thr) {
        rc = R;
        break;
      }
    *rc_out = (float)rc;
  }
}

__global__ __launch_bounds__(256) void k_fresca2d_inv(const float2* __restrict__ spec, float* __restrict__ out,
                                                      const float* __restrict__ rc_dev, float rc_host, float low,
                                                      float high, int H, int W, int C) {
  __shared__ float2 Xs[F2D_MAX_SPEC];
  __shared__ float2 Z[F2D_MAX_SPEC];
  __shared__ double2 twH[F2D_MAX_DIM], twW[F2D_MAX_DIM];
  const int nW = W / 2 + 1;
  const int b = blockIdx.x / C, c = blockIdx.x - b * C;
  const float rc = rc_dev ? *rc_dev : rc_host;
  f2d_tables(twH, twW, H, W);
  for (int i = threadIdx.x; i < H * nW; i += blockDim.x) {
    const int kh = i / nW, kw = i - kh * nW;
    const float f = sqrtf((float)(kh * kh + kw * kw)) <= rc ? low : high;  // low*low_mask*X + high*high_mask*X
    const float2 v = spec[(size_t)blockIdx.x * H * nW + i];
    Xs[i] = make_float2(f * v.x, f * v.y);
  }
  __syncthreads();
  const double sw = 1.0 / sqrt((double)W), sh = 1.0 / sqrt((double)H);
  for (int i = threadIdx.x; i < H * nW; i += blockDim.x) {  // inverse complex DFT along H
    const int h = i / nW, kw = i - h * nW;
    double re = 0.0, im = 0.0;
    int r = 0;
    for (int kh = 0; kh < H; ++kh) {
      const double a = (double)Xs[kh * nW + kw].x, bb = (double)Xs[kh * nW + kw].y;
      re += a * twH[r].x - bb * twH[r].y, im += bb * twH[r].x + a * twH[r].y;
      r += h;
      r -= r >= H ? H : 0;
    }
    Z[i] = make_float2((float)(re * sh), (float)(im * sh));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H * W; i += blockDim.x) {  // c2r along W
    const int h = i / W, w = i - h * W;
    double acc = 0.0;
    int r = 0;
    for (int kw = 0; kw < nW; ++kw) {
      const double a = (double)Z[h * nW + kw].x, bb = (double)Z[h * nW + kw].y;
      const double wgt = (kw == 0 || 2 * kw == W) ? 1.0 : 2.0;
      acc += wgt * (a * twW[r].x - bb * twW[r].y);
      r += w;
      r -= r >= W ? W : 0;
    }
    out[((size_t)b * H * W + i) * C + c] = (float)(acc * sw);
  }
}

bool fresca2d_supported(int H, int W) {
  return H >= 1 && W >= 1 && H <= F2D_MAX_DIM && W <= F2D_MAX_DIM && H * W <= F2D_MAX_HW;
}
size_t fresca2d_work_floats(int B, int H, int W, int C) { return (size_t)3 * B * C * H * (W / 2 + 1) + 4; }

// work: fresca2d_work_floats(B, H, W, C) floats, 8-byte aligned
hipError_t launch_fresca2d(const float* in, float* out, float* work, int B, int H, int W, int C, float low, float high,
                           double cutoff_ratio, int strategy, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  if (C < 1 || !fresca2d_supported(H, W)) return hipErrorInvalidValue;
  const int nW = W / 2 + 1, nimg = B * C;
  float* rc = work;
  float2* spec = reinterpret_cast<float2*>(work + 4);
  float* mag = work + 4 + (size_t)2 * nimg * H * nW;
  hipLaunchKernelGGL(k_fresca2d_fwd, dim3(nimg), dim3(256), 0, s, in, spec, mag, H, W, C);
  const float* rc_dev = nullptr;
  float rc_host = 0.f;
  if (strategy == 1) {
    hipLaunchKernelGGL(k_fresca2d_cut, dim3(1), dim3(256), 0, s, mag, rc, nimg, H, nW, (float)cutoff_ratio);
    rc_dev = rc;
  } else {  // spatial (fresca.py:84-86): Rc = r0 * min(H / 2, nW / 2), compared with the fp32 distances
    rc_host = (float)(cutoff_ratio * fmin((double)H / 2.0, (double)nW / 2.0));
  }
  hipLaunchKernelGGL(k_fresca2d_inv, dim3(nimg), dim3(256), 0, s, spec, out, rc_dev, rc_host, low, high, H, W, C);
  return hipGetLastError();
}


}  // namespace ffd
