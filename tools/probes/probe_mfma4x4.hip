// Probe the operand / result lane maps of v_mfma_f32_4x4x1_16b_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  int l = threadIdx.x;
  float a = 1.0f + l;       // A operand value identifies the source lane
  float b = 1000.0f + l;    // B operand
  f32x4 c = {0, 0, 0, 0};
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    float v = h[l * 4 + r]; int fa = -1, fb = -1;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if ((1.0f + la) * (1000.0f + lb) == v) { fa = la; fb = lb; }
    printf("lane %2d reg %d = A[lane %2d] * B[lane %2d]\n", l, r, fa, fb);
  }
  return 0;
}
