// What keeps ONE wave per SIMD from issuing v_mfma_f32_32x32x2_f32 back to back (64 cycles each)?  One workgroup of NWV
// waves, every wave runs the same pattern; cycles per MFMA from s_memtime.  Patterns:
//   0 one dependent accumulation chain, nothing else
//   1 two accumulators alternating
//   2 chain + one ds_read_b128 (result used 3 groups later) per 4 MFMAs
//   3 two accumulators alternating, B operand = v_max of a register (4 v_max per 4 MFMAs), + the ds_read
//   4 as 3 with the A operands taken from the ds_read results (the FFN's GEMM2 shape)
//   5 as 4 + one global_load_lds piece per 4 MFMAs
// build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o tools/probes/bench_mfma32_patterns tools/probes/bench_mfma32_patterns.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mf(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float e(const float4& q, int j) { return j == 0 ? q.x : j == 1 ? q.y : j == 2 ? q.z : q.w; }

template <int P>
__global__ __launch_bounds__(1024) void k(float* out, const float* gsrc, long long* cyc, int iters) {
  __shared__ __align__(16) float lds[32 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32 * 1024; i += blockDim.x) lds[i] = i * 1e-6f;
  __syncthreads();
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x16 acc0 = {0}, acc1 = {0};
  float hv[16];
  for (int i = 0; i < 16; ++i) hv[i] = i - 7.5f + lane;
  float4 f0 = {a, b, a, b}, f1 = f0, f2 = f0, f3 = f0;
  const float* base = lds + lane * 4;
  const unsigned ldsb = (unsigned)(unsigned long)((const __attribute__((address_space(3))) float*)lds) + wave * 1024;
  const unsigned ldsu = __builtin_amdgcn_readfirstlane(ldsb + 65536);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {  // 8 groups of 4 MFMAs
      float4 w = (P >= 4) ? f0 : float4{a, b, a, b};
      if (P >= 2) {
        f3 = *reinterpret_cast<const float4*>(base + ((it * 8 + g) & 31) * 256);
      }
      float bb[2] = {b, b};
      if (P >= 3) {
        bb[0] = __builtin_fmaxf(hv[2 * g], 0.f);
        bb[1] = __builtin_fmaxf(hv[2 * g + 1], 0.f);
      }
      if (P >= 5) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(ldsu), "v"(gsrc + lane * 4) : "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (P == 0 || P == 2) acc0 = mf(e(w, j), b, acc0);
        else if (j & 1) acc1 = mf(e(w, j), bb[j >> 1], acc1);
        else acc0 = mf(e(w, j), bb[j >> 1], acc0);
      }
      __builtin_amdgcn_sched_barrier(0);
      f0 = f1, f1 = f2, f2 = f3;
    }
    if (P >= 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc0[0] + acc1[0] + f0.x + f1.x + f2.x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}
int main() {
  float *o, *g; long long* c; (void)hipMalloc(&o, 1 << 22); (void)hipMalloc(&g, 1 << 22); (void)hipMalloc(&c, 64 * 16);
  const int iters = 500; long long h[16];
#define RUN(P, NWV)                                                                                   \
  hipLaunchKernelGGL((k<P>), dim3(1), dim3(64 * NWV), 0, 0, o, g, c, iters);                          \
  (void)hipMemcpy(h, c, 8 * NWV, hipMemcpyDeviceToHost);                                              \
  printf("pattern %d, %2d waves/CU: %.1f cycles per MFMA per wave (64 x waves per SIMD is the pipe rate)\n", P, NWV, (double)h[0] / (iters * 32.0));
  RUN(0, 4) RUN(1, 4) RUN(2, 4) RUN(3, 4) RUN(4, 4) RUN(5, 4)
  RUN(0, 8) RUN(3, 8) RUN(4, 8) RUN(5, 8) RUN(4, 12) RUN(5, 12)
  return 0;
}
