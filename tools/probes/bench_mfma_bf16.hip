// Issue rate of the bf16 MFMA shapes on gfx950: 16x16x32 (gfx950), 16x16x16 (the CDNA3 "_1k" form), 32x32x16, 32x32x8.
// Question behind it: is the 16-wide-K form half the cycles of the 32-wide one (then a K remainder of <= 16 is cheaper on it)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int CH, int KIND>
__global__ void kb(float* out, long long* cyc, int iters) {
  bf16x8 a8, b8;
  s16x4 a4, b4;
  for (int i = 0; i < 8; ++i) a8[i] = (__bf16)(threadIdx.x * 0.01f + i), b8[i] = (__bf16)(1.0f + i);
  for (int i = 0; i < 4; ++i) a4[i] = (short)(0x3f80 + threadIdx.x + i), b4[i] = (short)(0x3f80 + i);
  f32x4 acc[CH];
  f32x16 acc32[CH > 4 ? 4 : CH];
  for (int i = 0; i < CH; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < (CH > 4 ? 4 : CH); ++i)
    for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
      if (KIND == 2 && i < 4) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc32[i], 0, 0, 0);
      if (KIND == 3 && i < 4) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc32[i], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < (CH > 4 ? 4 : CH); ++i) s += acc32[i][0] + acc32[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* o; long long* c; (void)hipMalloc(&o, 1 << 20); (void)hipMalloc(&c, 64);
  const int iters = 2000; long long h;
#define RUN(name, kern, ch, waves)                                                         \
  hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), 0, 0, o, c, iters);                  \
  (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);                                        \
  printf("%-28s chains=%d waves/CU=%d : %.1f cycles per MFMA per wave\n", name, ch, waves, (double)h / (iters * ch));
  RUN("16x16x32 bf16", (kb<1, 0>), 1, 1) RUN("16x16x32 bf16", (kb<4, 0>), 4, 1) RUN("16x16x32 bf16", (kb<8, 0>), 8, 4)
  RUN("16x16x16 bf16 (_1k)", (kb<1, 1>), 1, 1) RUN("16x16x16 bf16 (_1k)", (kb<4, 1>), 4, 1) RUN("16x16x16 bf16 (_1k)", (kb<8, 1>), 8, 4)
  RUN("32x32x16 bf16", (kb<1, 2>), 1, 1) RUN("32x32x16 bf16", (kb<4, 2>), 4, 1)
  RUN("32x32x8 bf16 (_1k)", (kb<1, 3>), 1, 1) RUN("32x32x8 bf16 (_1k)", (kb<4, 3>), 4, 1)
  return 0;
}
