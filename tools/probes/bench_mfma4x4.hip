// Issue-rate / latency microbenchmark of v_mfma_f32_4x4x1_16b_f32 vs 16x16x4 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH, int CBSZ>
__global__ void k4(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x4 acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = f32x4{0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], CBSZ, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
__global__ void k16(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x4 acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = f32x4{0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* o; long long* c; (void)hipMalloc(&o, 1 << 20); (void)hipMalloc(&c, 64);
  const int iters = 2000; long long h;
#define RUN(name, kern, ch, waves)                                                         \
  hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, o, c, iters);                  \
  (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);                                        \
  printf("%-34s chains=%d waves/CU=%d : %.1f cycles per MFMA per wave\n", name, ch, waves, (double)h / (iters * ch));
  RUN("4x4x1 cbsz=0", (k4<1, 0>), 1, 1) RUN("4x4x1 cbsz=0", (k4<2, 0>), 2, 1) RUN("4x4x1 cbsz=0", (k4<4, 0>), 4, 1)
  RUN("4x4x1 cbsz=0", (k4<8, 0>), 8, 1) RUN("4x4x1 cbsz=4", (k4<8, 4>), 8, 1) RUN("4x4x1 cbsz=4", (k4<1, 4>), 1, 1)
  RUN("4x4x1 cbsz=0 (4 waves = 1/SIMD)", (k4<8, 0>), 8, 4) RUN("4x4x1 cbsz=0 (8 waves = 2/SIMD)", (k4<8, 0>), 8, 8)
  RUN("16x16x4", (k16<1>), 1, 1) RUN("16x16x4", (k16<4>), 4, 1) RUN("16x16x4 (8 waves)", (k16<4>), 4, 8)
  return 0;
}
