// Do fp32 MFMAs and fp32 VALU FMAs overlap on one SIMD?  One workgroup of 8 waves (2 per SIMD): waves 0-3 run an
// MFMA loop (v_mfma_f32_16x16x4_f32 or v_mfma_f32_32x32x2_f32), waves 4-7 a dependent-free v_fma_f32 loop.
// Reports cycles of each alone and of both together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>  // bit0: MFMA waves active, bit1: VALU waves active, bit2: use 32x32x2, bit3: the VALU waves run integer
                     // v_max_i32 (the ReLU as an integer max on the float bits) instead of v_fma_f32
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x4 acc[4];
  f32x16 big[2];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 2; ++i) big[i] = f32x16{0};
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  int vi[8] = {1, -2, 3, -4, 5, -6, 7, -8};
  const int ia = threadIdx.x;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (MODE & 1) {
      for (int it = 0; it < iters; ++it) {
        if (MODE & 4) {
#pragma unroll
          for (int r = 0; r < 8; ++r) big[r & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[r & 1], 0, 0, 0);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[r & 3], 0, 0, 0);
        }
      }
    }
  } else if (MODE & 2) {
    for (int it = 0; it < iters; ++it) {
      if (MODE & 8) {
#pragma unroll
        for (int r = 0; r < 128; ++r) asm volatile("v_max_i32 %0, %0, %1" : "+v"(vi[r & 7]) : "v"(ia));
      } else {
#pragma unroll
        for (int r = 0; r < 128; ++r) v[r & 7] = __builtin_fmaf(v[r & 7], a, b);
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  s += big[0][0] + big[1][0];
  for (int i = 0; i < 8; ++i) s += v[i] + vi[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
int main() {
  float* o; long long* c; (void)hipMalloc(&o, 1 << 22); (void)hipMalloc(&c, 64 * 8);
  const int iters = 2000; long long h[8];
#define RUN(mode, name)                                                                                   \
  hipLaunchKernelGGL((k<mode>), dim3(1), dim3(512), 0, 0, o, c, iters);                                     \
  (void)hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);                                                         \
  printf("%-44s mfma wave %8lld cycles   valu wave %8lld cycles\n", name, h[0], h[4]);
  RUN(1, "16x16x4 MFMA alone (512 cyc/iter ideal)") RUN(2, "v_fma_f32 alone (512 cyc/iter ideal)") RUN(3, "16x16x4 MFMA + v_fma_f32 together")
  RUN(5, "32x32x2 MFMA alone") RUN(7, "32x32x2 MFMA + v_fma_f32 together")
  RUN(10, "v_max_i32 alone") RUN(11, "16x16x4 MFMA + v_max_i32 together")
  return 0;
}
