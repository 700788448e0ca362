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
                                             float scale) {
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
      float re = src[(size_t)kk * C + c0 + cc];
      float im = 0.f;
      if (kk >= 1 && kk <= L - n_re) im = src[(size_t)(n_re + kk - 1) * C + c0 + cc];
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
      dst[(size_t)o * C + c0 + cc] = v * scale;
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

hipError_t launch_dft(const float* in, float* out, int B, int L, int C, int inverse, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  if (L < 1 || C < 1 || L > 8192) return hipErrorInvalidValue;
  const float2* W = nullptr;
  hipError_t e = get_twiddles(L, &W);
  if (e != hipSuccess) return e;
  FftPlan plan = make_plan(L);
  if (plan.npass > 16) return hipErrorInvalidValue;
  // channels per workgroup so that twiddles + two slabs fit in 64 KiB of LDS
  int CG = C;
  while (CG > 1 && (size_t)(L + 2 * (size_t)L * CG) * sizeof(float2) > 64 * 1024) CG = (CG + 1) / 2;
  size_t lds = (size_t)(L + 2 * (size_t)L * CG) * sizeof(float2);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  dim3 grid(B, cdiv(C, CG)), block(256);
  float scale = (float)(1.0 / sqrt((double)L));
  if (inverse)
    hipLaunchKernelGGL(k_fft<true>, grid, block, lds, s, in, out, W, plan, L, C, CG, scale);
  else
    hipLaunchKernelGGL(k_fft<false>, grid, block, lds, s, in, out, W, plan, L, C, CG, scale);
  return hipGetLastError();
}

hipError_t launch_fresca(const float* in, float* out, float* work, int B, int L, int C, float low, float high,
                         double cutoff_ratio, int strategy, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  if (L < 2 || C < 1 || L > 4096) return hipErrorInvalidValue;
  const float2* W = nullptr;
  hipError_t e = get_twiddles(L, &W);
  if (e != hipSuccess) return e;
  FftPlan plan = make_plan(L);
  int CG = C;
  while (CG > 1 && (size_t)(L + 2 * (size_t)L * CG) * sizeof(float2) > 64 * 1024) CG = (CG + 1) / 2;
  const size_t lds = (size_t)(L + 2 * (size_t)L * CG) * sizeof(float2);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  const int NGc = cdiv(C, CG);
  const int nf = L / 2 + 1;
  const float sc = (float)(1.0 / sqrt((double)L));
  const int* rc_dev = nullptr;
  float rc_host = 0.f;
  if (strategy == 1) {  // energy (fresca.py:46-58)
    float* partial = work;
    int* rc = reinterpret_cast<int*>(work + (size_t)B * NGc * nf);
    hipLaunchKernelGGL(k_fresca_spectrum, dim3(B, NGc), dim3(256), lds, s, in, partial, W, plan, L, C, CG, sc);
    hipLaunchKernelGGL(k_fresca_cutoff, dim3(1), dim3(256), nf * sizeof(float), s, partial, rc, B, B * NGc, nf, C,
                       cutoff_ratio);
    rc_dev = rc;
  } else {  // spatial (fresca.py:40-43): Rc = r0 * n_freq, compared in fp32 like the reference's k tensor
    rc_host = (float)(cutoff_ratio * (double)nf);
  }
  hipLaunchKernelGGL(k_fresca_apply, dim3(B, NGc), dim3(256), lds, s, in, out, W, plan, L, C, CG, rc_dev, rc_host, low,
                     high, sc * sc);
  return hipGetLastError();
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
  FftPlan plan = make_plan(L);
  int CG = D;
  while (CG > 1 && (size_t)(L + 2 * (size_t)L * CG) * sizeof(float2) > 64 * 1024) CG = (CG + 1) / 2;
  const size_t lds = (size_t)(L + 2 * (size_t)L * CG) * sizeof(float2);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  const int nf = L / 2 + 1;
  int n_low = (int)((double)nf * low_freq_ratio);  // fourier.py:249  max(1, int(n_freq * ratio))
  if (n_low < 1) n_low = 1;
  const float rc = (float)n_low - 0.5f;             // bins k <= rc  <=>  k < n_low
  const float sc = (float)(1.0 / sqrt((double)L));
  dim3 grid(B, cdiv(D, CG)), block(256);
  hipLaunchKernelGGL(k_fresca_apply, grid, block, lds, s, in, low, W, plan, L, D, CG, (const int*)nullptr, rc, 1.f, 0.f,
                     sc * sc);
  hipLaunchKernelGGL(k_fresca_apply, grid, block, lds, s, in, high, W, plan, L, D, CG, (const int*)nullptr, rc, 0.f, 1.f,
                     sc * sc);
  return hipGetLastError();
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

}  // namespace ffd
