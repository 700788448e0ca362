// How does the v_mfma_f32_16x16x4_f32 pipe share between 1..4 waves per SIMD, with 2 or 4 accumulator chains?
// Reports cycles per MFMA per SIMD (32 = pipe saturated).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(1024) void k16(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x4 acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = f32x4{0, 0, 0, 0};
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* o; long long* c; (void)hipMalloc(&o, 1 << 22); (void)hipMalloc(&c, 64);
  const int iters = 500; long long h;
#define RUN(kern, ch, waves)                                                                         \
  hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, o, c, iters);                              \
  (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);                                                    \
  printf("chains=%d waves/SIMD=%d : %.1f cycles per MFMA per SIMD\n", ch, waves / 4,               \
         (double)h / ((double)iters * 8 * ch * (waves / 4)));
  RUN((k16<2>), 2, 4) RUN((k16<2>), 2, 8) RUN((k16<2>), 2, 12) RUN((k16<2>), 2, 16)
  RUN((k16<4>), 4, 4) RUN((k16<4>), 4, 8) RUN((k16<4>), 4, 12) RUN((k16<1>), 1, 4) RUN((k16<1>), 1, 8) RUN((k16<1>), 1, 12)
  return 0;
}
